#!/usr/bin/env python3
"""Header-chain data commitment by MapReduce (data_commitment_mr.HeaderChainMapReduce): python3 profiles/chain_mr_probe.py [headers=256] [leaf=4] [fan=8]"""
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

n, leaf, fan = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 256), (2, 4), (3, 8)))
pkg = graft.load_package()
dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
bs = importlib.import_module(graft.PKG_NAME + ".blobstream")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
provers = [pkg.Prover(0) for _ in range(3)]
for p in provers:
    p.set_poseidon_constants(*consts)
rng = np.random.default_rng(21)
lens = (4, 12, 5, 13, 72, 34, 34, 34, 34, 34, 34, 34, 34, 22)


def chain(start, first, count):
    prev, out = start, []
    for k in range(count):
        f = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for L in lens]
        f[2] = b"\x08" + bs.encode_varint(first + k)
        f[4] = b"\x0a\x20" + prev + f[4][34:]
        f[6] = b"\x0a\x20" + f[6][2:]
        out.append(f)
        prev = dm.HeaderChainMapReduce.header_hash(f)
    return out, prev


mr = dm.HeaderChainMapReduce(provers[0], consts, leaf_headers=leaf, fan_in=fan, map_provers=provers[1:])
res = {"headers": n, "leaf_headers": leaf, "fan_in": fan}
for run in ("first_run_records_circuits", "steady_state"):
    start, first = hashlib.sha256(run.encode()).digest(), 4_000_000
    hdrs, end = chain(start, first, n)
    t0 = time.perf_counter()
    out = mr.prove_chain(start, first, hdrs)
    dt = time.perf_counter() - t0
    lvl = [hashlib.sha256(b"\x00" + int(first + k).to_bytes(32, "big") + hdrs[k][6][2:]).digest() for k in range(n)]
    while len(lvl) > 1:
        lvl = [hashlib.sha256(b"\x01" + lvl[i] + lvl[i + 1]).digest() for i in range(0, len(lvl), 2)]
    ok = out["end_hash"] == end and out["commitment"] == lvl[0] and mr.verify_chain(out["root_proof"], out["key"], start, end, lvl[0], first)
    res[run] = {"seconds": round(dt, 3), "map_seconds": out["map_seconds"], "reduce_seconds": out["reduce_seconds"], "levels": out["levels"],
                "end_hash_and_commitment_match_hashlib_and_verify": bool(ok), "headers_per_second": round(n / dt, 1)}
res["record_seconds"] = out["record_seconds"]
res["leaf"] = {k: v for k, v in mr.leaf_program.stats.items() if k in ("rows", "rows_used", "sha_rows")}
print(json.dumps(res))

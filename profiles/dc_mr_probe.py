#!/usr/bin/env python3
"""DataCommitment of a large range by MapReduce (data_commitment_mr.py): python3 profiles/dc_mr_probe.py [blocks=4096] [leaf_blocks=64] [fan_in=8]
-> one JSON line (first run records the circuits; the second run is the steady state)."""
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

blocks, leaf_blocks, fan_in = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 4096), (2, 64), (3, 8)))
pkg = graft.load_package()
dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0)
pr.set_poseidon_constants(*consts)
rng = np.random.default_rng(12)


def rnd_range():
    hs = [3_000_000 + i for i in range(blocks)]
    return hs, [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in hs]


def root(hs, rs):
    lvl = [hashlib.sha256(b"\x00" + int(h).to_bytes(32, "big") + r).digest() for h, r in zip(hs, rs)]
    while len(lvl) > 1:
        lvl = [hashlib.sha256(b"\x01" + lvl[i] + lvl[i + 1]).digest() for i in range(0, len(lvl), 2)]
    return lvl[0]


n_extra = int(os.environ.get("DC_MR_EXTRA_PROVERS", "2"))
extra = [pkg.Prover(0) for _ in range(n_extra)]
for p in extra:
    p.set_poseidon_constants(*consts)
mr = dm.DataCommitmentMapReduce(pr, consts, leaf_blocks=leaf_blocks, fan_in=fan_in, map_provers=extra)
res = {"blocks": blocks, "leaf_blocks": leaf_blocks, "fan_in": fan_in, "map_provers": 1 + n_extra}
for run in ("first_run_records_circuits", "steady_state"):
    hs, rs = rnd_range()
    t0 = time.perf_counter()
    out = mr.prove_range(hs, rs)
    t1 = time.perf_counter()
    ok = out["commitment"] == root(hs, rs) and mr.verify(out["root_proof"], out["key"], hs, rs, out["commitment"])
    t2 = time.perf_counter()
    res[run] = {"seconds": round(t1 - t0, 3), "map_seconds": out["map_seconds"], "reduce_seconds": out["reduce_seconds"], "levels": out["levels"],
                "commitment_matches_hashlib_and_verifies": bool(ok), "verify_seconds": round(t2 - t1, 4), "root_proof_bytes": len(out["root_proof"])}
res["record_seconds"] = out["record_seconds"]
res["leaf_rows"] = mr.leaf_program.stats["rows"]
res["node_stats"] = {f"level{k[0]}_fan{k[1]}": {s: v for s, v in rp.stats.items() if s in ("rows", "rows_used", "poseidon_rows", "sha_rows", "arith_gates")}
                     for k, rp in mr.nodes.items()}
print(json.dumps(res))

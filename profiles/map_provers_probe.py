#!/usr/bin/env python3
"""profiles/map_provers_probe.py — Map throughput of the round-3 leaves as a function of the number of concurrent provers (ctxs) per GPU:
64 signature-slot leaves and 64 eight-header chain leaves (2^17 rows x 144 wires each), recordings made once, then timed per prover count."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
sm = importlib.import_module(graft.PKG_NAME + ".signature_mr")
dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
ec = importlib.import_module(graft.PKG_NAME + ".ed25519_circuit")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
N = 64
for k in (1, 2, 3, 4, 5):
    provers = [pkg.Prover(0) for _ in range(k)]
    for p in provers:
        p.set_poseidon_constants(*consts)
    sig = sm.SignatureSetMapReduce(provers[0], consts, map_provers=provers[1:])
    sig._record_leaf()
    msgs = [sig.vote_bytes(bytes(32), i) for i in range(N)]
    keys = [ec.keypair_and_sign(bytes([i]) * 32, msgs[i]) for i in range(N)]
    slots = ([kk[0] for kk in keys], [kk[1] for kk in keys], msgs, [True] * N)
    sig._map(slots, 0, 8)
    t0 = time.perf_counter()
    sig._map(slots, 0, N)
    t_sig = time.perf_counter() - t0
    ch = dm.HeaderChainMapReduce(provers[0], consts, leaf_headers=8, fan_in=8, map_provers=provers[1:])
    ch._record_leaf()
    hdrs, _ = ch.synthetic_chain(8 * N, 4_000_000)
    hashes = [bytes(32)] + [ch.header_hash(h) for h in hdrs]
    ch._map_chain(hashes, 4_000_000, hdrs, 0, 64)
    t0 = time.perf_counter()
    ch._map_chain(hashes, 4_000_000, hdrs, 0, 8 * N)
    t_ch = time.perf_counter() - t0
    print(json.dumps({"provers_per_gpu": k, "signature_leaves_ms_each": round(1e3 * t_sig / N, 2), "chain_leaves_ms_each": round(1e3 * t_ch / N, 2)}), flush=True)
    sig.free()
    ch.free()
    for p in provers:
        p.close()

#!/usr/bin/env python3
"""Workload for rocprofv3: ONE leaf circuit of the round-3 MapReduces, recorded once and proved 6 times on one prover.
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/leaf_kstats -- python3 profiles/leaf_workload.py sig     (Ed25519 slot leaf, 2^17 x 144)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/leaf_kstats -- python3 profiles/leaf_workload.py chain   (8-header chain leaf, 2^17 x 144)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0)
pr.set_poseidon_constants(*consts)
which = sys.argv[1] if len(sys.argv) > 1 else "sig"
if which == "sig":
    sm = importlib.import_module(graft.PKG_NAME + ".signature_mr")
    ec = importlib.import_module(graft.PKG_NAME + ".ed25519_circuit")
    mr = sm.SignatureSetMapReduce(pr, consts)
    msg = mr.vote_bytes(bytes(32), 1)
    pub, sig = ec.keypair_and_sign(bytes(range(32)), msg)
    job = lambda: mr.prove_leaf(pub, sig, msg, True)
else:
    dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
    mr = dm.HeaderChainMapReduce(pr, consts, leaf_headers=8, fan_in=8)
    hdrs, _ = mr.synthetic_chain(8, 4_000_000)
    job = lambda: mr.prove_leaf(bytes(32), 4_000_000, hdrs)
job()
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    job()
    ts.append(time.perf_counter() - t0)
print({"leaf": which, "rows": mr.leaf_program.stats["rows"], "rows_used": mr.leaf_program.stats["rows_used"],
       "witness_plus_prove_ms": [round(1e3 * t, 2) for t in ts], "record_seconds": mr.record_seconds})

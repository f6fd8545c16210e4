#!/usr/bin/env python3
"""Workload for rocprofv3: the 64-block DataCommitment circuit on the SHA row gates (2^16 rows x 144 wires), built once and proved 6 times.
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sha_kstats -- python3 profiles/sha_rows_workload.py"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
gd = importlib.import_module(graft.PKG_NAME + ".gadgets")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
rc, circ, diag = (np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0)
pr.set_poseidon_constants(rc, circ, diag)
rng = np.random.default_rng(11)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
heights = [2_000_000 + i for i in range(nb)]
roots = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(nb)]
ck, dw, public, root = gd.data_commitment_rows_circuit(pr, heights, roots)
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    proof = ck.prove_(dw, 28, 16, public=public)
    ts.append(time.perf_counter() - t0)
print({"blocks": nb, "rows": 1 << ck.log_n, "prove_ms": [round(1e3 * t, 2) for t in ts], "verified": bool(ck.verify(proof, 28, 16, public=public))})

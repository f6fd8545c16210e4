#!/usr/bin/env python3
"""profiles/record_profile.py — where the seconds of a FIRST run go: cProfile over the recording of one recursion node (8 DataCommitment leaf proofs of
64 blocks each, 28 queries) — Python builder vs WitnessProgram construction vs circuit setup on the GPU."""
import cProfile
import importlib
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0)
pr.set_poseidon_constants(*consts)
mr = dm.DataCommitmentMapReduce(pr, consts, leaf_blocks=64, fan_in=8)
n = 8 * 64
heights = list(range(1000, 1000 + n))
roots = [bytes([k % 251]) * 32 for k in range(n)]
t0 = time.perf_counter()
leaves = mr.prove_leaves(heights, roots)
t1 = time.perf_counter()
prof = cProfile.Profile()
prof.enable()
out = mr.reduce(leaves, [])
prof.disable()
t2 = time.perf_counter()
print(f"leaf recording + 8 leaf proofs {t1 - t0:.2f} s; node recording + proof {t2 - t1:.2f} s; recorded: {mr.record_seconds}")
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:160] for l in s.getvalue().splitlines()[:60]))
t3 = time.perf_counter()
mr.reduce(leaves, [])
print(f"second reduce {time.perf_counter() - t3:.3f} s")
mr.free()
pr.close()

#!/usr/bin/env python3
"""Fold 16 leaf proofs into ONE proof by verifying them in-circuit, save the recorded circuit, load it again (needs an MI355X).
    python examples/recursion_reduce.py"""
import importlib
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402  (only for its synthetic leaf circuit)

pkg = graft.load_package()
vc = importlib.import_module(graft.PKG_NAME + ".verifier_circuit")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0)
pr.set_poseidon_constants(*consts)

c, s, w = bench.synthetic_circuit(pr, 14, 80)                                   # a 2^14-row leaf circuit
leaf = pkg.PlonkCircuit(pr, c, s)
dw = pr.to_device(w)
proofs = [leaf.prove_(dw, 28, 16) for _ in range(16)]

t0 = time.perf_counter()
rp = vc.RecursionProgram(pr, proofs, leaf.cap(), 28, 16, 80, consts)            # recorded once per (leaf circuit, fan-in)
print(f"recorded the recursion circuit in {time.perf_counter() - t0:.1f} s: {rp.stats['rows']} rows")
t0 = time.perf_counter()
root, public = rp.prove(proofs)
print(f"root proof in {1e3 * (time.perf_counter() - t0):.0f} ms, {len(root)} bytes, {len(public)} public inputs")
assert pr.plonk_verify(root, rp.key(), 28, 16, public=public)                    # needs no leaf proof

with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "recursion16.npz")
    rp.save(path)
    t0 = time.perf_counter()
    again = vc.RecursionProgram.load(pr, path, consts)
    print(f"loaded and committed the recording in {time.perf_counter() - t0:.2f} s; same key: {np.array_equal(again.key(), rp.key())}")
    root2, public2 = again.prove(proofs)
    assert pr.plonk_verify(root2, rp.key(), 28, 16, public=public2)
    again.free()
rp.free()
pr.close()
print("ok")

#!/usr/bin/env python3
"""Where the per-Reduce host time of the recursion goes: RecursionProgram.witness() split into its stages, for FAN leaf proofs of the bench's
leaf circuit (2^16 x 80).  Usage: python3 profiles/recursion_witness_breakdown.py [fan]  -> one JSON line."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

fan = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pkg = graft.load_package()
vcm = importlib.import_module(graft.PKG_NAME + ".verifier_circuit")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
rc, circ, diag = (np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0)
pr.set_poseidon_constants(rc, circ, diag)
consts, sigmas, wires = bench.synthetic_circuit(pr, 16, 80)
ck = pkg.PlonkCircuit(pr, consts, sigmas)
dw = pr.to_device(wires)
proofs = [ck.prove_(dw, 28, 16) for _ in range(fan)]
t0 = time.perf_counter()
rp = vcm.RecursionProgram(pr, proofs, ck.cap(), 28, 16, 80, (rc, circ, diag), ext_gate=bool(int(os.environ.get("GLP_REC_EXT", "0"))))
t_rec = time.perf_counter() - t0
out = {"fan": fan, "record_seconds": round(t_rec, 3), "stats": rp.stats, "prog_words": int(rp.program.prog.size)}
for rep in range(2):
    t = [time.perf_counter()]
    inputs, ws = rp.program.inputs_from_words(proofs); t.append(time.perf_counter())
    vals = rp.program.evaluate(rp.consts, inputs); t.append(time.perf_counter())
    rp.program.check_words(vals, ws); t.append(time.perf_counter())
    dwv, pub = rp.program.device_witness(pr, vals); pr.sync(); t.append(time.perf_counter())
    proof = rp.circuit.prove_(dwv, 28, 16, public=pub); t.append(time.perf_counter())
    dwv.free()
    names = ["inputs_from_words", "evaluate", "check_words", "device_witness", "prove"]
    out[f"rep{rep}"] = {n: round(b - a, 4) for n, a, b in zip(names, t, t[1:])}
t0 = time.perf_counter()
rp.save("/tmp/rp_rec.npz")
t1 = time.perf_counter()
rp2 = vcm.RecursionProgram.load(pr, "/tmp/rp_rec.npz", (rc, circ, diag))
t2 = time.perf_counter()
out["save_seconds"], out["load_and_commit_seconds"], out["file_mb"] = round(t1 - t0, 3), round(t2 - t1, 3), round(os.path.getsize("/tmp/rp_rec.npz") / 2**20, 1)
out["loaded_key_matches"] = bool(np.array_equal(rp2.key(), rp.key()))
out["verified"] = bool(rp.circuit.verify(proof, 28, 16, public=pub))
print(json.dumps(out))

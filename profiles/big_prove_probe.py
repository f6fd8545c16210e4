"""profiles/big_prove_probe.py — measurement aid: one proof of the build-defined circuit ABOVE the bench size
(default 2^22 rows x 80 wires: 2.7e9 LDE elements per batch, i.e. element counts past 2^31) checked by the native
verifier — a scale test of the index arithmetic, not a benchmark."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
graft = bench.graft
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
W = int(sys.argv[2]) if len(sys.argv) > 2 else 80
pkg = graft.load_package()
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
pr = pkg.Prover(0)
rc, circ, diag = pc.default_constants()
pr.set_poseidon_constants(np.array(rc, dtype=np.uint64), np.array(circ, dtype=np.uint64), np.array(diag, dtype=np.uint64))
t0 = time.perf_counter()
consts, sigmas, wires = bench.synthetic_circuit(pr, log_n, W)
print(f"circuit 2^{log_n} x {W} generated in {time.perf_counter() - t0:.1f} s", flush=True)
ck = pkg.PlonkCircuit(pr, consts, sigmas)
dw = pr.to_device(wires)
del consts, sigmas, wires
for rep in range(2):
    pr.sync()
    t0 = time.perf_counter()
    proof = ck.prove_(dw, 28, 16)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    ok = ck.verify(proof, 28, 16)
    print(f"rep {rep}: prove {dt:.4f} s, {len(proof)} bytes, native verify {time.perf_counter() - t1:.4f} s -> {ok} {pr.last_reject or ''}", flush=True)
    assert ok
pr.set_profiling(True)
ck.prove_(dw, 28, 16)
print("stage ms:", {k: round(v, 1) for k, v in pr.last_stage_ms()}, flush=True)
pr.set_profiling(False)

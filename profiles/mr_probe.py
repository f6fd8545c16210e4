import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
graft = bench.graft
use_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
if use_torch:
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(4, device="cuda")
pkg = graft.load_package()
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
pr = pkg.Prover(0)
rc, circ, diag = pc.default_constants()
pr.set_poseidon_constants(np.array(rc, dtype=np.uint64), np.array(circ, dtype=np.uint64), np.array(diag, dtype=np.uint64))
consts, sigmas, wires = bench.synthetic_circuit(pr, 16, 80)
ck = pkg.PlonkCircuit(pr, consts, sigmas)
dw = pr.to_device(wires)
ck.prove_(dw, 28, 16)
ts = []
for i in range(16):
    t0 = time.perf_counter(); p = ck.prove_(dw, 28, 16); ts.append(time.perf_counter() - t0)
print("torch" if use_torch else "plain", "per-leaf ms:", [round(t * 1e3, 1) for t in ts], "total", round(sum(ts), 4))

import cProfile, pstats, importlib, sys, os, io, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as graft, bench
pkg = graft.load_package()
vcm = importlib.import_module(graft.PKG_NAME + ".verifier_circuit")
rec = importlib.import_module(graft.PKG_NAME + ".recursion")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
rc, circ, diag = (np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0); pr.set_poseidon_constants(rc, circ, diag)
consts, sigmas, wires = bench.synthetic_circuit(pr, 16, 80)
ck = pkg.PlonkCircuit(pr, consts, sigmas); dw = pr.to_device(wires)
proofs = [ck.prove_(dw, 28, 16) for _ in range(4)]
b = rec.CircuitBuilder(pr)
prof = cProfile.Profile(); prof.enable()
t0=time.perf_counter()
for k, p in enumerate(proofs):
    b.begin_segment(); vcm.verify_in_circuit(b, p, ck.cap(), 28, 16, 80, None, 0, 4, None, proof_id=k); b.end_segment()
t1=time.perf_counter()
prog = b.program()
t2=time.perf_counter()
c = prog.setup(pr)
t3=time.perf_counter()
prof.disable()
print("builder", t1-t0, "program()", t2-t1, "setup", t3-t2)
s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:6000])

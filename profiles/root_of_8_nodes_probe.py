import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as graft, bench
pkg = graft.load_package()
mr = importlib.import_module(graft.PKG_NAME + ".mapreduce")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0); pr.set_poseidon_constants(*consts)
c, s, w = bench.synthetic_circuit(pr, 16, 80)
ck = pkg.PlonkCircuit(pr, c, s); dw = pr.to_device(w)
leaves = [ck.prove_(dw, 28, 16) for _ in range(16)]
f = mr.RecursionFolders(pr, {"key": ck.cap(), "num_queries": 28, "pow_bits": 16, "n_wires": 80}, consts)
t0 = time.perf_counter()
nodes = [f.fold_local(leaves) for _ in range(8)]            # what 8 ranks would each produce (same leaves here)
t1 = time.perf_counter()
root = f.fold_root(nodes)
t2 = time.perf_counter()
root = f.fold_root(nodes)
t3 = time.perf_counter()
ok = pr.plonk_verify(root, f.key, 28, 16, public=f.public)
print({"nodes_s": round(t1 - t0, 2), "root_first_s_incl_record": round(t2 - t1, 2), "root_steady_s": round(t3 - t2, 4), "verified": bool(ok),
       "root_stats": f.programs[(2, 8)].stats, "record": f.record_seconds, "root_bytes": len(root), "public": len(f.public)})

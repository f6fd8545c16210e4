import cProfile, pstats, io, importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as graft
pkg = graft.load_package()
dm = importlib.import_module(graft.PKG_NAME + ".data_commitment_mr")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
consts = tuple(np.array(a, dtype=np.uint64) for a in pc.default_constants())
pr = pkg.Prover(0); pr.set_poseidon_constants(*consts)
mr = dm.DataCommitmentMapReduce(pr, consts, leaf_blocks=64, fan_in=8)
rng = np.random.default_rng(1)
hs = [100 + i for i in range(64 * 16)]; rs = [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in hs]
mr.prove_leaves(hs[:128], rs[:128])
prof = cProfile.Profile(); prof.enable()
t0 = time.perf_counter()
mr.prove_leaves(hs, rs)
t1 = time.perf_counter()
prof.disable()
print("16 leaves, one prover:", round(t1 - t0, 4))
s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])

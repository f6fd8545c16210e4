"""profiles/mr_probe.py — measurement aid: where the time of a one-rank MapReduce goes
(per-leaf prove, pack, gather/unpack, native verification)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
graft = bench.graft
torch.cuda.set_device(0)
pkg = graft.load_package()
mr = importlib.import_module(graft.PKG_NAME + ".mapreduce")
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
pr = pkg.Prover(0)
rc, circ, diag = pc.default_constants()
pr.set_poseidon_constants(np.array(rc, dtype=np.uint64), np.array(circ, dtype=np.uint64), np.array(diag, dtype=np.uint64))
consts, sigmas, wires = bench.synthetic_circuit(pr, 16, 80)
ck = pkg.PlonkCircuit(pr, consts, sigmas)
dw = pr.to_device(wires)
for rep in range(3):
    t0 = time.perf_counter()
    blobs = [(i, ck.prove_(dw, 28, 16)) for i in range(16)]
    t1 = time.perf_counter()
    packed = mr.pack_leaves(blobs, 1 << 18)
    t2 = time.perf_counter()
    proofs = mr.allgather_leaf_proofs(blobs, 16, 1 << 18)
    t3 = time.perf_counter()
    ok = all(ck.verify(p, 28, 16) for p in proofs)
    t4 = time.perf_counter()
    print(f"rep {rep}: prove {t1 - t0:.4f}  pack {t2 - t1:.4f}  gather {t3 - t2:.4f}  verify {t4 - t3:.4f} ok={ok}", flush=True)

# concurrent provers on one GPU: K ctxs, K host threads
for K in (1, 2, 3, 4):
    provers = [pr] + [pkg.Prover(0) for _ in range(K - 1)]
    for q in provers[1:]:
        q.set_poseidon_constants(np.array(rc, dtype=np.uint64), np.array(circ, dtype=np.uint64), np.array(diag, dtype=np.uint64))
    cks = [ck] + [pkg.PlonkCircuit(q, consts, sigmas) for q in provers[1:]]
    dws = [dw] + [q.to_device(wires) for q in provers[1:]]
    workers = [(lambda i, c=c, d=d: c.prove_(d, 28, 16)) for c, d in zip(cks, dws)]
    mr.map_prove_gather(workers, K, padded_len=1 << 18)
    for rep in range(2):
        t0 = time.perf_counter()
        proofs = mr.map_prove_gather(workers, 16, padded_len=1 << 18)
        dt = time.perf_counter() - t0
        print(f"K={K} rep {rep}: 16 leaves in {dt:.4f} s  ({16 / dt:.1f} leaves/s)  same={proofs[0] == proofs[5]}", flush=True)
    for c, d, q in list(zip(cks, dws, provers))[1:]:
        d.free(); c.free(); q.close()

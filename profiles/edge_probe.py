"""profiles/edge_probe.py — scale/edge probe (not a benchmark): a batch larger than the NTT scratch cap (chunked path) at
1024 x 2^20, the widest (128 wires) and narrowest (8 wires) circuits at 2^18 rows, and a 2^27-leaf leaf-major Merkle tree."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
graft = bench.graft
pkg = graft.load_package()
P = pkg.P
pc = importlib.import_module(graft.PKG_NAME + ".poseidon_constants")
pr = pkg.Prover(0)
rc, circ, diag = pc.default_constants()
pr.set_poseidon_constants(np.array(rc, dtype=np.uint64), np.array(circ, dtype=np.uint64), np.array(diag, dtype=np.uint64))

# 1. chunked batch: 1024 polynomials of 2^20 (8 GiB) against a 4 GiB scratch: delta rows -> w^(k*j) spot values, round trip
log_n, batch = 20, 1024
n = 1 << log_n
d = pr.alloc(batch * n * 8)
z = np.zeros(n, dtype=np.uint64)
rng = np.random.default_rng(3)
pos = {}
for b in range(batch):
    k = int(rng.integers(0, n)) if b % 97 == 0 or b in (0, 1, 511, 512, 1023) else 1
    z[:] = 0
    z[k] = 1
    pos[b] = k
    pr._chk(pr.lib.glp_h2d(pr.ctx, d.ptr + b * n * 8, z.ctypes.data, z.nbytes), "h2d")
t0 = time.perf_counter()
pr.ntt_(d, log_n, batch)
pr.sync()
dt = time.perf_counter() - t0
w = pow(7, (P - 1) >> log_n, P)
for b in (0, 1, 97, 511, 512, 970, 1023):
    for j in (0, 1, n // 2 + 3, n - 1, int(rng.integers(0, n))):
        got = int(d.download((1,), offset_bytes=(b * n + j) * 8)[0])
        assert got == pow(w, pos[b] * j, P), (b, j)
pr.ntt_(d, log_n, batch, inverse=True)
for b in (0, 97, 512, 1023):
    row = d.download((n,), offset_bytes=b * n * 8)
    assert row[pos[b]] == 1 and int(row.sum()) == 1, b
print(f"chunked batch {batch} x 2^{log_n}: forward {dt * 1e3:.1f} ms ({16.0 * n * batch / dt / 1e9:.0f} GB/s), spot values and round trip ok", flush=True)
d.free()

# 2. widest and narrowest circuits
for W in (128, 8):
    consts, sigmas, wires = bench.synthetic_circuit(pr, 18, W)
    ck = pkg.PlonkCircuit(pr, consts, sigmas)
    proof = ck.prove(wires, 28, 16)
    ok = ck.verify(proof, 28, 16)
    print(f"2^18 x {W}: proof {len(proof)} bytes, native verify -> {ok} {pr.last_reject or ''}", flush=True)
    assert ok
    ck.free()

# 3. leaf-major Merkle tree with 2^27 leaves of 3 elements (noop-hash leaves) and 2^24 leaves of 9 elements
for log_leaves, leaf_len in ((27, 3), (24, 9)):
    nl = 1 << log_leaves
    leaves = pr.alloc(nl * leaf_len * 8)
    blk = rng.integers(0, P, size=(1 << 20) * leaf_len, dtype=np.uint64)
    for off in range(0, nl * leaf_len, blk.size):
        pr._chk(pr.lib.glp_h2d(pr.ctx, leaves.ptr + off * 8, blk.ctypes.data, min(blk.nbytes, (nl * leaf_len - off) * 8)), "h2d")
    dig = pr.alloc(8 * pkg.Prover.merkle_digest_len(log_leaves, 4))
    t0 = time.perf_counter()
    cap = pr.merkle_(leaves, leaf_len, log_leaves, 4, dig)
    dt = time.perf_counter() - t0
    # every leaf block is the same 2^20-leaf pattern: the 2^(log_leaves-20) subtrees of height 20 are identical, so all cap entries are equal
    cap = np.asarray(cap).reshape(-1, 4)
    assert (cap == cap[0]).all() and cap[0].any()
    print(f"merkle 2^{log_leaves} leaves x {leaf_len}: {dt * 1e3:.1f} ms, cap entries identical as constructed", flush=True)
    leaves.free()
    dig.free()
pr.close()

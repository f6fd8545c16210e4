"""profiles/big_ntt_probe.py — scale probe (not a benchmark): one transform of 2^30 points (8 GiB, byte offsets past 2^32
inside one polynomial): delta -> w^k spot values, constant -> n*delta_0, round trip; and a 2^24 -> 2^27 coset LDE of a
constant and of the monomial X.  Only sampled words are downloaded."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft

pkg = graft.load_package()
P = pkg.P
pr = pkg.Prover(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << log_n
d = pr.alloc(n * 8)
chunk = np.zeros(1 << 24, dtype=np.uint64)
for off in range(0, n, 1 << 24):                       # zero fill, then delta at index 1
    pr._chk(pr.lib.glp_h2d(pr.ctx, d.ptr + off * 8, chunk.ctypes.data, chunk.nbytes), "h2d")
one = np.array([0, 1], dtype=np.uint64)
pr._chk(pr.lib.glp_h2d(pr.ctx, d.ptr, one.ctypes.data, 16), "h2d")
t0 = time.perf_counter()
pr.ntt_(d, log_n, 1)
pr.sync()
dt = time.perf_counter() - t0
w = pow(7, (P - 1) >> log_n, P)
rng = np.random.default_rng(1)
idx = [0, 1, 2, n // 2, n // 2 + 1, n - 1, (1 << 29) + 12345 if log_n > 29 else 77] + [int(rng.integers(0, n)) for _ in range(24)]
for k in idx:
    got = int(d.download((1,), offset_bytes=k * 8)[0])
    assert got == pow(w, k, P), k
print(f"2^{log_n} forward: {dt * 1e3:.1f} ms ({16.0 * n / dt / 1e9:.0f} GB/s algorithmic), plan {pr.describe_plan(log_n, 1)}; delta spot values ok", flush=True)
pr.ntt_(d, log_n, 1, inverse=True)
for k in idx:
    got = int(d.download((1,), offset_bytes=k * 8)[0])
    assert got == (1 if k == 1 else 0), k
print("round trip spot values ok", flush=True)
for rep in range(3):                                  # steady state (tables, plan and scratch exist now)
    t0 = time.perf_counter()
    pr.ntt_(d, log_n, 1)
    pr.sync()
    dt = time.perf_counter() - t0
print(f"2^{log_n} forward, warm: {dt * 1e3:.1f} ms ({16.0 * n / dt / 1e9:.0f} GB/s algorithmic)", flush=True)
d.free()
# coset LDE 2^24 -> 2^27 (bit-reversed): constant 5 and the monomial X
ln, rb = 24, 3
m, N = 1 << ln, 1 << (ln + rb)
c = np.zeros((2, m), dtype=np.uint64)
c[0, 0] = 5
c[1, 1] = 1
di = pr.to_device(c)
do = pr.alloc(2 * N * 8)
pr.lde_coset_(di, do, ln, rb, 2, 7, pkg.NTT_BITREV)
wN = pow(7, (P - 1) >> (ln + rb), P)
for i in [0, 1, N - 1, N // 2, m, m - 1] + [int(rng.integers(0, N)) for _ in range(20)]:
    assert int(do.download((1,), offset_bytes=i * 8)[0]) == 5
    x = 7 * pow(wN, int(format(i, f"0{ln + rb}b")[::-1], 2), P) % P
    assert int(do.download((1,), offset_bytes=(N + i) * 8)[0]) == x, i
print("coset LDE 2^24 -> 2^27 spot values ok", flush=True)
pr.close()

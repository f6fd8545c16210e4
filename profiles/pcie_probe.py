"""profiles/pcie_probe.py — measurement aid: the PCIe-inclusive rate of the bench workload when the boundary is
handed HOST buffers (h2d of 128 x 2^20 u64 = 1 GiB, one forward NTT, d2h), pageable numpy memory through glp_h2d/glp_d2h."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
from bench import splitmix_fill  # noqa: E402

pkg = graft.load_package()
pr = pkg.Prover(0)
log_n, batch = 20, 128
x = splitmix_fill(batch << log_n, 1).reshape(batch, 1 << log_n)
d = pr.alloc(x.nbytes)
for rep in range(3):
    t0 = time.perf_counter()
    d.upload(x)
    t1 = time.perf_counter()
    pr.ntt_(d, log_n, batch)
    pr.sync()
    t2 = time.perf_counter()
    y = d.download(x.shape)
    t3 = time.perf_counter()
    gb = x.nbytes / 1e9
    print(f"rep {rep}: h2d {t1 - t0:.4f} s ({gb / (t1 - t0):.1f} GB/s)  ntt {t2 - t1:.4f} s  d2h {t3 - t2:.4f} s ({gb / (t3 - t2):.1f} GB/s)  "
          f"end-to-end {16.0 * (batch << log_n) / (t3 - t0) / 1e9:.1f} GB/s algorithmic", flush=True)
d.free()
pr.close()

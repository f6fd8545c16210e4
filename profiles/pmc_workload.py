#!/usr/bin/env python3
"""profiles/pmc_workload.py — the fixed workload profiled under rocprofv3 --pmc / --kernel-trace
(measurement aid, not product code).  It runs, on cuda:0:
  1. a calibration kernel with a KNOWN byte count and the same 8-byte-per-lane access width as
     the NTT kernels: glp_field_op(add) over 2^27 elements = 2 GiB read + 1 GiB written;
  2. 3 forward NTTs, n = 2^20, batch = 128 (the bench workload; 16*n*batch = 2 GiB algorithmic);
  3. 3 forward NTTs, n = 2^24, batch = 8.
Usage on the GPU box (from the repo root):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 profiles/pmc_workload.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 profiles/pmc_workload.py
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktrace -- python3 profiles/pmc_workload.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
from bench import splitmix_fill  # noqa: E402

pkg = graft.load_package()
pr = pkg.Prover(0)
n = 1 << 27
a = pr.to_device(splitmix_fill(n, 1))
b = pr.to_device(splitmix_fill(n, 2))
o = pr.alloc(n * 8)
for _ in range(2):
    pr._chk(pr.lib.glp_field_op(pr.ctx, 0, a.ptr, b.ptr, o.ptr, n), "field_op")
pr.sync()
for x in (a, b, o):
    x.free()
for log_n, batch in ((20, 128), (24, 8)):
    d = pr.to_device(splitmix_fill(batch << log_n, 3))
    for _ in range(3):
        pr.ntt_(d, log_n, batch)
    pr.sync()
    d.free()
pr.close()
print("pmc workload done")

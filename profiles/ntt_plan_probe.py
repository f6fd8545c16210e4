#!/usr/bin/env python3
"""profiles/ntt_plan_probe.py [log_n batch plan ...] — time alternative pass plans of one size (HIP events, per pass) and check each against the
default plan's output bit for bit.  Measurement aid for the radix-32 / split-exchange work of round 3."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

pkg = graft.load_package()
pr = pkg.Prover(0)
log_n, batch = int(sys.argv[1]), int(sys.argv[2])
plans = [None] + sys.argv[3:]
n = 1 << log_n
host = bench.splitmix_fill(n * batch, 3).reshape(batch, n)
ref = None
for rnd in range(2):
    for plan in plans:
        pr.set_plan(log_n, plan)
        d = pr.to_device(host)
        pr.ntt_(d, log_n, batch)
        out = d.download((batch, n))
        if ref is None:
            ref = out
        same = bool(np.array_equal(out, ref))
        ms = bench.time_ntt(pr, d, log_n, batch, steps=20, warmup=3)
        pr.set_profiling(True)
        acc = None
        for _ in range(5):
            pr.ntt_(d, log_n, batch)
            pm = pr.last_pass_ms()
            acc = pm if acc is None else [a + b for a, b in zip(acc, pm)]
        pr.set_profiling(False)
        d.free()
        print(json.dumps({"round": rnd, "plan": pr.describe_plan(log_n, batch), "ms": round(ms, 4), "gbps": round(16.0 * n * batch / ms / 1e6, 1),
                          "pass_ms": [round(a / 5, 4) for a in acc], "equals_default_plan": same}), flush=True)
pr.set_plan(log_n, None)
pr.close()

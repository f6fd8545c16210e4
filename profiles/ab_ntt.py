#!/usr/bin/env python3
"""A/B timing of library builds on one box: profiles/ab_ntt.py lib_a.so lib_b.so ...  — alternates the builds `rounds` times (boxes
and DVFS states differ by a few percent: only same-run alternation is comparable) and prints the bench line's value and per-pass
kernel times for each.  Tuning aid; GLP_LIB selects the build (0-kno-blobstreamx_amd/__init__.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
rounds = int(os.environ.get("AB_ROUNDS", "3"))
extra = os.environ.get("AB_ARGS", "--no-cpu --no-sizes --no-prove --steps 30 --warmup 5").split()
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, GLP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             text=True).stdout.strip().splitlines()
        try:
            d = json.loads(out[-1])
            print(json.dumps({"round": r, "lib": os.path.basename(lib), "value": d["value"], "pass_ms": d["roofline"]["pass_ms_profiling_mode"],
                              "kernel_gbps": d["roofline"]["achieved"]}), flush=True)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"round": r, "lib": os.path.basename(lib), "error": str(e), "tail": out[-1:] }), flush=True)

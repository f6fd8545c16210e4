#!/usr/bin/env python3
"""profiles/prove_workload.py — one setup + two proofs of the build-defined circuit at 2^20 rows x
80 wires, for `rocprofv3 --kernel-trace --stats` (measurement aid, not product code)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.prove_bench([(20, 80)])

#!/usr/bin/env python3
"""Summarise rocprofv3 csv output (kernel trace + one --pmc counter) into per-kernel averages.
usage: summarize_pmc.py <dir-with-*_counter_collection.csv or *_kernel_trace.csv> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    for d in sys.argv[1:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = defaultdict(lambda: defaultdict(list))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            print(f"# {f}")
            for k, cs in acc.items():
                for c, v in cs.items():
                    print(f"{k[:70]:70s} {c:12s} n={len(v):3d} mean={sum(v)/len(v):.1f} min={min(v):.1f} max={max(v):.1f}")
        for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            acc = defaultdict(list)
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    acc[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
            print(f"# {f}")
            for k, v in acc.items():
                print(f"{k[:70]:70s} calls={len(v):3d} avg_us={sum(v)/len(v):.1f} min_us={min(v):.1f} max_us={max(v):.1f}")


if __name__ == "__main__":
    main()

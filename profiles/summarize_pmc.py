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


def traffic_json(fetch_dir, write_dir, out_path, elements_by_kernel):
    """per-kernel HBM bytes per launch: FETCH_SIZE (KB) * 1024 * 2 (gfx950 reports exactly half of
    a streaming read: MI355X_MICROARCH.md §HBM; confirmed by the glp_field_op_kernel calibration,
    2 GiB read -> 1,048,6xx KB reported) and WRITE_SIZE (KB) * 1024 (exact)."""
    import json

    def means(d, counter):
        acc = defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] == counter:
                        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in acc.items()}

    fe, wr = means(fetch_dir, "FETCH_SIZE"), means(write_dir, "WRITE_SIZE")
    out = {}
    for k in fe:
        if "glp_" not in k:
            continue
        out[k] = {"fetch_bytes": fe[k] * 1024 * 2, "write_bytes": wr.get(k, 0.0) * 1024,
                  "fetch_size_kb_raw": fe[k], "write_size_kb_raw": wr.get(k, 0.0)}
    with open(out_path, "w") as f:
        json.dump({"note": "per launch; workload = profiles/pmc_workload.py (2^27 elements per launch); FETCH corrected x2",
                   "kernels": out}, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
        traffic_json(sys.argv[2], sys.argv[3], sys.argv[4], None)
    else:
        main()

#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (tools/profile_gpu.sh) into one JSON: per kernel, launch count and
average duration from the kernel trace, and per-launch averages of every PMC counter collected."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    out = {"kernels": {}}
    # what was profiled: the hash of the library sources in this snapshot (bench.py uses the counters only while it still holds)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["csrc_sha256"] = bench.csrc_sha256()
    # kernel trace
    for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        meta = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "")
                dur[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                meta[name] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                      "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y")}
        for name, d in dur.items():
            d.sort()
            out["kernels"].setdefault(name, {}).update({
                "launches": len(d), "avg_us": sum(d) / len(d) / 1e3, "median_us": d[len(d) // 2] / 1e3,
                "min_us": d[0] / 1e3, "max_us": d[-1] / 1e3, **meta[name]})
    # stats file, verbatim rows for the top kernels
    for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as f:
            out["kernel_stats_csv"] = [row for row in csv.DictReader(f)][:8]
    # counters
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        with open(path) as f:
            for row in csv.DictReader(f):
                acc[row.get("Kernel_Name", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, ctrs in acc.items():
            k = out["kernels"].setdefault(name, {})
            c = k.setdefault("counters_per_launch", {})
            for cname, vals in ctrs.items():
                c[cname] = sum(vals) / len(vals)
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

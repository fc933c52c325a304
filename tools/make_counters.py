#!/usr/bin/env python3
"""profiles/counters.json: per-launch PMC averages of the dominant kernel of a bench configuration, taken from a
committed *_summary.json (tools/summarize_prof.py over tools/profile_gpu.sh).  bench.py reads it for
roofline.traffic and roofline.valu.

  tools/make_counters.py profiles/r02_c2_summary.json 'C2_RGB_ASCII_rtx_trace<RTX_K_RGB_ASCII,true>' 'rtx_trace<2, true, 0'

HBM traffic per launch = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 bytes: rocprofv3 reports both in KiB, and on gfx950
FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, "HBM"); the two counters come from
separate --pmc passes.
"""
import json
import os
import sys


def main():
    summary_path, key, match = sys.argv[1:4]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(summary_path) as f:
        summ = json.load(f)
    names = [n for n in summ["kernels"] if match in n]
    if len(names) != 1:
        raise SystemExit("kernel match %r is not unique: %r" % (match, names))
    k = summ["kernels"][names[0]]
    c = k["counters_per_launch"]
    entry = {"rocprof_kernel": names[0], "csrc_sha256": summ.get("csrc_sha256"), "launches": k.get("launches"), "avg_us": k.get("avg_us"), "median_us": k.get("median_us"),
             "source": "%s (rocprofv3 --pmc, one counter group per pass; FETCH_SIZE doubled per MI355X_MICROARCH.md)"
                       % os.path.relpath(os.path.abspath(summary_path), root)}
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY",
                 "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"):
        if name in c:
            entry[name] = c[name]
    if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
        entry["write_bytes"] = c["WRITE_SIZE"] * 1024.0
        entry["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024.0
        entry["fetch_bytes_corrected_x2"] = 2.0 * entry["fetch_bytes_raw"]
        entry["total_bytes"] = entry["write_bytes"] + entry["fetch_bytes_corrected_x2"]
    out_path = os.path.join(root, "profiles", "counters.json")
    data = {}
    if os.path.exists(out_path):
        with open(out_path) as f:
            data = json.load(f)
    data[key] = entry
    with open(out_path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(json.dumps({key: entry}, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The table rows of DESIGN.md section 4.1 from the committed profiles of a tag (default r04_z): per config the algorithmic
bytes per frame, rocprofv3's average launch duration (one launch at a time), the bench line's in-flight figure, the HBM
fraction both ways, executed VALU wave-instructions per launch and active lanes, PMC traffic.

  python tools/design_numbers.py [tag]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_z"
P = os.path.join(ROOT, "profiles")


def load(name):
    with open(os.path.join(P, name)) as f:
        return json.load(f)


rows = [("C1 320×180, 8+1 (brute)", "c1", "bench_C1", "rtx_trace<2, false, 0, false"),
        ("**C2 1920×1080, 1024+1**", "c2", "bench_c2", "rtx_trace<2, true, 0, false"),
        ("C2, BIT_ASCII (12-byte records)", "c2bit", None, "rtx_trace<0, true, 0, false"),
        ("C3 3840×2160, 4096+6", "c3", "bench_C3", "rtx_trace<2, true, 0, false"),
        ("C4 7680×4320, 1024", "c4", "bench_C4", "rtx_trace<2, true, 0, false"),
        ("C5 1920×1080, 65 536 (REFINE)", "c5", "bench_C5", "rtx_trace<2, true, 0, true")]
main = load("%s_bench_c2.json" % tag)
for label, prof, bench, kern in rows:
    s = load("%s_%s_summary.json" % (tag, prof))
    k = next(v for n, v in s["kernels"].items() if kern in n)
    c = k["counters_per_launch"]
    if bench:
        b = load("%s_%s.json" % (tag, bench))
        alg, inflight, mv = b["roofline"]["bytes_per_launch"], b["ms_per_step"] * 1e3, b["timing"].get("moving_view", {})
    else:
        b = main["modes"]["BIT_ASCII"]
        alg, inflight, mv = b["roofline"]["bytes_per_launch"], b["ms_per_step"] * 1e3, {}
    alone = k["avg_us"]
    lanes = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] if c.get("SQ_ACTIVE_INST_VALU") else float("nan")
    traffic = (c.get("WRITE_SIZE", 0) + 2 * c.get("FETCH_SIZE", 0)) * 1024
    turning = " (turning 0.001 rad/frame: %.1f / %.1f)" % (mv["in_flight_ms_per_frame"] * 1e3, mv["alone_ms_per_frame"] * 1e3) if mv.get("in_flight_ms_per_frame") else ""
    print("| %s | %.2f MB | %.2f | %.2f%s | %.0f (%.3f) alone, %.0f (%.3f) in flight | %.2f M at %.1f lanes; traffic %.1f MB |" % (
        label, alg / 1e6, alone, inflight, turning, alg / alone / 1e3, alg / alone / 1e3 / 8000.0, alg / inflight / 1e3, alg / inflight / 1e3 / 8000.0,
        c["SQ_INSTS_VALU"] / 1e6, lanes, traffic / 1e6))
e = main["end_to_end"]
print("\nUpdate: pipelined %.3f ms, blocking %.3f ms, %.2f MB over PCIe at %.1f GB/s" % (e["ms_per_update_pipelined"], e["ms_per_update_blocking"], e["pcie_bytes_per_update"] / 1e6, e["pcie_GBs_pipelined"]))
for name in ("update", "updaterec"):
    s = load("%s_%s_summary.json" % (tag, name))
    for n, v in s["kernels"].items():
        if "rtx" in n and v.get("launches", 0) > 100:
            c = v.get("counters_per_launch", {})
            print("  %-10s %-60s %6.2f us  write %.1f MB fetch x2 %.1f MB" % (name, n[:60], v["avg_us"], c.get("WRITE_SIZE", 0) * 1024 / 1e6, 2 * c.get("FETCH_SIZE", 0) * 1024 / 1e6))
cb = main["cpu_baseline"]
print("\ncpu_baseline: %.2f Mrays/s at %d threads (by threads %r), T=1 %.3f; host %r; line value %.0f = %.0fx" % (
    cb["value"], cb["cores"], cb["by_threads"], cb["single_thread"]["value"], cb["host"], main["value"], main["speedup_vs_cpu_baseline"]))
print("line: ms_per_step %.5f, alone %.5f, driver form %.5f" % (main["ms_per_step"], main["roofline"]["kernel_ms"], load("%s_bench_c2_driver_form.json" % tag)["ms_per_step"]))

#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV of overlapping launches (bench.py's default: 4 frames in flight on 4
streams): per kernel the average duration of one launch, the rate at which launches COMPLETE in the steady state
(microseconds per frame), and the average number of launches executing at once -- the evidence behind
bench.py's roofline.pipelined.  Prints JSON; --excerpt N adds N consecutive rows (start / end relative to the
first, stream / queue id) so that the overlap can be read off directly.

  tools/overlap_from_trace.py gpurun_out/prof_x/trace [--kernel rtx_trace] [--excerpt 24]
"""
import argparse
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--kernel", default="rtx_trace")
    ap.add_argument("--excerpt", type=int, default=24)
    args = ap.parse_args()
    paths = glob.glob(os.path.join(args.root, "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        raise SystemExit("no kernel_trace.csv under %s" % args.root)
    rows = []
    with open(paths[0]) as f:
        for r in csv.DictReader(f):
            if args.kernel in r.get("Kernel_Name", ""):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Kernel_Name"]))
    rows.sort()
    n = len(rows)
    if n < 64:
        raise SystemExit("only %d launches of %r" % (n, args.kernel))
    # steady state: the last half of the run (after the run-in and warm-up)
    tail = rows[n // 2:]
    durs = sorted(e - s for s, e, *_ in tail)
    ends = sorted(e for _, e, *_ in tail)
    span = ends[-1] - ends[0]
    busy = sum(e - s for s, e, *_ in tail)
    # gaps between completions, and the longest run of back-to-back overlapping launches
    overlapped = sum(1 for a, b in zip(tail[:-1], tail[1:]) if b[0] < a[1])
    out = {
        "trace_csv": os.path.relpath(paths[0]),
        "kernel": tail[0][4], "launches_total": n, "launches_in_window": len(tail),
        "launch_duration_us": {"avg": sum(durs) / len(durs) / 1e3, "median": durs[len(durs) // 2] / 1e3, "min": durs[0] / 1e3, "max": durs[-1] / 1e3},
        "completion_interval_us": span / (len(tail) - 1) / 1e3,
        "average_concurrency": busy / float(max(e for _, e, *_ in tail) - min(s for s, *_ in tail)),
        "launches_starting_before_their_predecessor_ends": overlapped,
        "queues": sorted(set(r[2] for r in tail)), "streams": sorted(set(r[3] for r in tail)),
    }
    if args.excerpt:
        t0 = tail[0][0]
        out["excerpt_ns_relative"] = [{"start": s - t0, "end": e - t0, "queue": q, "stream": st} for s, e, q, st, _ in tail[:args.excerpt]]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

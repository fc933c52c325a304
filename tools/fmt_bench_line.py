#!/usr/bin/env python3
"""One bench.py JSON line on stdin -> one readable line (tools/force_dist_gpu.sh)."""
import json
import sys

d = json.loads(sys.stdin.read())
lat = d["timing"].get("frame_latency") or {}
chk = (d["timing"].get("check_rounds") or {}).get("passed")
print("verified", d.get("verified_against_golden"), "check rounds", chk, "us/frame %.2f" % (1e3 * d["ms_per_step"]),
      "latency median/max ms", lat.get("median_ms"), lat.get("max_ms"), "|", d["config"]["parallelism"].split(";")[-1][:40])

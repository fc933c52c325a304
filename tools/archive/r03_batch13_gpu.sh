#!/bin/bash
# Round 3, batch 13: direction-sorted copy of the sphere array for staging (RTX_OPT_SORTED_STORE).
T=${TAG:-r03_o}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/${T}_tests.log
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); mv=d['timing'].get('moving_view',{}); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'verified', d['verified_against_golden'], 'moving', mv.get('in_flight_ms_per_frame'), mv.get('alone_ms_per_frame'))"; }
for c in C5 C3 C2 C4; do for s in 0 -1; do python bench.py --no-cpu-baseline --config $c --sorted-store $s 2>/dev/null | line "$c --sorted-store $s"; done; done
tools/profile_gpu.sh ${T}_c5 --config C5 > gpurun_out/${T}_prof_c5.log 2>&1; echo "prof c5 rc $?"
python3 - <<PY
import json
d=json.load(open("gpurun_out/prof_${T}_c5/summary.json"))
for k,v in d["kernels"].items():
    c=v.get("counters_per_launch",{})
    if "rtx" in k: print(k[:60], "launches", v.get("launches"), "avg_us", round(v.get("avg_us",0),2), "FETCHx2 MB", round(2*c.get("FETCH_SIZE",0)*1024/1e6,2), "WRITE MB", round(c.get("WRITE_SIZE",0)*1024/1e6,2))
PY

#!/bin/bash
# Round 3, batch 2: after the capacity feedback / readiness-aware slot choice / experiment hooks in rtx_experiment.inc.
set -o pipefail
T=${TAG:-r03_d}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/${T}_tests.log
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); mv=d['timing'].get('moving_view',{}); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'verified', d['verified_against_golden'], 'moving in flight / alone', mv.get('in_flight_ms_per_frame'), mv.get('alone_ms_per_frame'), d['config']['kernel'])"; }
for c in C2 C3 C5; do python bench.py --config $c --no-cpu-baseline 2>gpurun_out/${T}_bench_$c.err | tee gpurun_out/${T}_bench_$c.json | line $c; done
python bench.py --no-cpu-baseline --two-level 1 2>/dev/null | tee gpurun_out/${T}_bench_C2_two_level.json | line "C2 --two-level 1"
python bench.py --no-cpu-baseline --config C3 --frames-in-flight 6 2>/dev/null | line "C3 6 in flight"
python tools/moving_camera_gpu.py 0.001 > gpurun_out/${T}_moving.txt 2>&1; cat gpurun_out/${T}_moving.txt
python tools/worst_view_gpu.py > gpurun_out/${T}_worst_view.txt 2>&1; cat gpurun_out/${T}_worst_view.txt
python tools/worst_view_gpu.py stamps > gpurun_out/${T}_worst_view_stamps.txt 2>&1; cat gpurun_out/${T}_worst_view_stamps.txt
tools/profile_gpu.sh ${T}_c5 --config C5 > gpurun_out/${T}_prof_c5.log 2>&1; echo "prof c5 rc $?"
python3 - <<PY
import json
d=json.load(open("gpurun_out/prof_${T}_c5/summary.json"))
for k,v in d["kernels"].items():
    c=v.get("counters_per_launch",{})
    print(k[:60], "launches", v.get("launches"), "avg_us", v.get("avg_us"), "FETCHx2 MB", 2*c.get("FETCH_SIZE",0)*1024/1e6, "WRITE MB", c.get("WRITE_SIZE",0)*1024/1e6, "VALU", c.get("SQ_INSTS_VALU"))
PY

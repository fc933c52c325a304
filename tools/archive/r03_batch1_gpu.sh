#!/bin/bash
# Round 3, first batch after the cell-list reuse / planner refactor: the whole -m gpu suite, bench lines of C2 / C3 / C5 (frames in
# flight and alone), C2 with two-level culling forced on, the N>1 walk with the C4 sub-record, and the rocprofv3 summary of C5.
set -o pipefail
T=${TAG:-r03_c}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/${T}_tests.log
for c in C2 C3 C5; do
  python bench.py --config $c --no-cpu-baseline > gpurun_out/${T}_bench_$c.json 2>gpurun_out/${T}_bench_$c.err; echo "$c rc $?"
  python3 - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench_$c.json"))
print("$c", "in flight %.2f us" % (1e3*d["ms_per_step"]), "alone %.2f us" % (1e3*d["roofline"]["kernel_ms"]), "verified", d["verified_against_golden"], "moving", d["timing"].get("moving_view",{}).get("in_flight_ms_per_frame"), d["timing"].get("moving_view",{}).get("alone_ms_per_frame"), d["config"]["kernel"])
PY
done
for extra in "--two-level 1" "--two-level 1 --tile-order 0" "--config C5 --two-level 1 --frames-in-flight 1"; do
  python bench.py --no-cpu-baseline $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$extra:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'verified', d['verified_against_golden'], 'moving', d['timing'].get('moving_view',{}).get('in_flight_ms_per_frame'), d['timing'].get('moving_view',{}).get('alone_ms_per_frame'))"
done
RTX_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_force_dist_default.json 2> gpurun_out/${T}_force_dist_default.err; echo "force-dist default rc $?"
python3 - <<PY
import json
d=json.load(open("gpurun_out/${T}_force_dist_default.json"))
c=d["configs"]["C4"]
print("force-dist: C2 verified", d["verified_against_golden"], "us/frame %.1f" % (1e3*d["ms_per_step"]), "| C4 verified", c["verified_against_golden"], "check rounds", c["timing"]["check_rounds"]["passed"], "us/frame %.1f" % (1e3*c["ms_per_step"]), "| cpu", d["cpu_baseline"]["value"])
PY
tools/profile_gpu.sh ${T}_c5 --config C5 > gpurun_out/${T}_prof_c5.log 2>&1; echo "prof c5 rc $?"
python3 - <<PY
import json
d=json.load(open("gpurun_out/prof_${T}_c5/summary.json"))
for k,v in d["kernels"].items():
    c=v.get("counters_per_launch",{})
    print(k[:60], "avg_us", v.get("avg_us"), "FETCHx2 MB", 2*c.get("FETCH_SIZE",0)*1024/1e6, "WRITE MB", c.get("WRITE_SIZE",0)*1024/1e6, "VALU", c.get("SQ_INSTS_VALU"))
PY

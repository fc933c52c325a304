#!/bin/bash
# Round 3, batch 9: candidate statistics of the dense plan along the yaw sweep (for the way back to the sparse plan); lists for resting views
# of small scenes (still_only); config 3 with the adaptive plan restricted to one-round grids.
set -o pipefail
T=${TAG:-r03_k}
mkdir -p gpurun_out
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); mv=d['timing'].get('moving_view',{}); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'verified', d['verified_against_golden'], 'moving in flight / alone', mv.get('in_flight_ms_per_frame'), mv.get('alone_ms_per_frame'), mv.get('cell_lists'), d['config']['kernel'])"; }
for c in C2 C3 C1; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tee gpurun_out/${T}_bench_$c.json | line $c; done
python bench.py --no-cpu-baseline --cell-reuse 0 2>/dev/null | line "C2 --cell-reuse 0"
python bench.py --no-cpu-baseline --mode BIT_ASCII 2>/dev/null | line "C2 BIT_ASCII"
echo "=== dense plan, candidates per tile along the sweep"
python tools/worst_view_gpu.py stamps --coarse --no-adapt --subtiles=2 --two-level --refine=1 2>&1 | grep -v amdgpu.ids | grep "yaw pi" | cut -c1-200
echo "=== sparse plan"
python tools/worst_view_gpu.py stamps --coarse --no-adapt 2>&1 | grep -v amdgpu.ids | grep "yaw pi" | cut -c1-200

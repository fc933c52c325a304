#!/bin/bash
# Round 3, batch 5: where does the moving view of config 5 lose its frames in flight?  Budget length / prefetch point sweep; wave features of
# the heaviest tiles of the worst view.
set -o pipefail
T=${TAG:-r03_g}
mkdir -p gpurun_out
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); mv=d['timing'].get('moving_view',{}); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'moving in flight / alone', mv.get('in_flight_ms_per_frame'), mv.get('alone_ms_per_frame'), mv.get('cell_lists'))"; }
for fr in 8 16 32; do for at in 0.5 0.25; do
  RTX_CELL_FRAMES=$fr RTX_CELL_PREFETCH_AT=$at python bench.py --no-cpu-baseline --config C5 --no-verify 2>/dev/null | line "C5 frames $fr prefetch at $at"
done; done
RTX_CELL_FRAMES=16 RTX_CELL_PREFETCH_AT=0.25 python bench.py --no-cpu-baseline --config C5 --no-verify --frames-in-flight 4 2>/dev/null | line "C5 frames 16 at 0.25, 4 in flight"
GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --config C5 --no-verify 2>/dev/null | line "C5 default, 8 hw queues"
GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --config C5 --no-verify --cell-reuse 0 2>/dev/null | line "C5 reuse off, 8 hw queues"
RTX_CELL_FRAMES=16 RTX_CELL_PREFETCH_AT=0.25 python bench.py --no-cpu-baseline --two-level 1 --no-verify 2>/dev/null | line "C2 two-level frames 16 at 0.25"
RTX_CELL_FRAMES=16 RTX_CELL_PREFETCH_AT=0.25 python bench.py --no-cpu-baseline --config C3 --no-verify 2>/dev/null | line "C3 frames 16 at 0.25"
python tools/worst_view_gpu.py stamps > gpurun_out/${T}_worst_view_stamps.txt 2>&1; grep -A9 "pi+1.4\|pi+0.0" gpurun_out/${T}_worst_view_stamps.txt | cut -c1-400

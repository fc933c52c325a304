#!/bin/bash
# Round 3, batch 6: lists for the next stretch built on the render stream when several streams carry frames in flight.
set -o pipefail
T=${TAG:-r03_h}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_reuse.py tests/test_gpu_post.py -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/${T}_tests.log
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); mv=d['timing'].get('moving_view',{}); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'moving in flight / alone', mv.get('in_flight_ms_per_frame'), mv.get('alone_ms_per_frame'), mv.get('cell_lists'))"; }
for fr in 8 16; do
  RTX_CELL_FRAMES=$fr python bench.py --no-cpu-baseline --config C5 --no-verify 2>/dev/null | line "C5 frames $fr"
  RTX_CELL_FRAMES=$fr python bench.py --no-cpu-baseline --config C3 --no-verify 2>/dev/null | line "C3 frames $fr"
  RTX_CELL_FRAMES=$fr python bench.py --no-cpu-baseline --two-level 1 --no-verify 2>/dev/null | line "C2 two-level frames $fr"
done
python bench.py --no-cpu-baseline --config C5 --no-verify --cell-reuse 0 2>/dev/null | line "C5 reuse off"
python bench.py --no-cpu-baseline --no-verify 2>/dev/null | line "C2 default"

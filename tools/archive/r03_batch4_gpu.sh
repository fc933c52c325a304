#!/bin/bash
# Round 3, batch 4: precise reader events for rebuilds (no drain of the frames in flight), depth bound reverted.
set -o pipefail
T=${TAG:-r03_f}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/${T}_tests.log
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); mv=d['timing'].get('moving_view',{}); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'verified', d['verified_against_golden'], 'moving in flight / alone', mv.get('in_flight_ms_per_frame'), mv.get('alone_ms_per_frame'), mv.get('cell_lists'))"; }
for c in C2 C3 C5; do python bench.py --config $c --no-cpu-baseline 2>gpurun_out/${T}_bench_$c.err | tee gpurun_out/${T}_bench_$c.json | line $c; done
python bench.py --no-cpu-baseline --config C5 --cell-reuse 0 2>/dev/null | line "C5 --cell-reuse 0"
python bench.py --no-cpu-baseline --config C3 --cell-reuse 0 2>/dev/null | line "C3 --cell-reuse 0"
python bench.py --no-cpu-baseline --two-level 1 2>/dev/null | line "C2 --two-level 1"
python bench.py --no-cpu-baseline --two-level 1 --cell-reuse 0 2>/dev/null | line "C2 --two-level 1 --cell-reuse 0"
python bench.py --no-cpu-baseline --config C4 2>/dev/null | line "C4"
python bench.py --no-cpu-baseline --config C4 --two-level 1 2>/dev/null | line "C4 --two-level 1"
python bench.py --no-cpu-baseline --config C1 2>/dev/null | line "C1"

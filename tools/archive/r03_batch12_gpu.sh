#!/bin/bash
# Round 3, batch 12: sub-tiles at least 8 pixels wide; C5 by tile width.
T=${TAG:-r03_n}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/${T}_tests.log
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'verified', d['verified_against_golden'])"; }
for t in 0 3 4 5; do python bench.py --no-cpu-baseline --no-moving-view --config C5 --tile $t 2>/dev/null | line "C5 --tile $t"; done
for c in C1 C2 C3 C4; do python bench.py --no-cpu-baseline --no-moving-view --config $c 2>/dev/null | line "$c"; done
for t in 3 4; do for sub in 4 8; do python bench.py --no-cpu-baseline --no-moving-view --config C4 --tile $t --subtiles $sub 2>/dev/null | line "C4 --tile $t --subtiles $sub"; done; done
python bench.py --no-cpu-baseline --no-moving-view --config C3 --subtiles 8 2>/dev/null | line "C3 --subtiles 8"
python bench.py --no-cpu-baseline --no-moving-view --config C3 --subtiles 2 2>/dev/null | line "C3 --subtiles 2"

#!/bin/bash
# Round 3, batch 15: does narrowing the candidate list below the macro tile pay on config 2?  REFINE (per wave and pass) at equal sub-tile counts.
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), d['config']['kernel'])"; }
for sub in 4 2; do for r in 0 1; do
  python bench.py --no-cpu-baseline --no-moving-view --no-verify --subtiles $sub --refine $r 2>/dev/null | line "C2 --subtiles $sub --refine $r"
done; done
python bench.py --no-cpu-baseline --no-moving-view --no-verify 2>/dev/null | line "C2 default"
for sub in 4 8; do for r in 0 1; do
  python bench.py --no-cpu-baseline --no-moving-view --no-verify --config C3 --subtiles $sub --refine $r 2>/dev/null | line "C3 --subtiles $sub --refine $r"
done; done

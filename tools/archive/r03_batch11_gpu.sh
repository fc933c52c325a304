#!/bin/bash
# Round 3, batch 11: sub-tile width (records per row segment) against culling tightness: C4, C3, C2 by --tile; frames in flight sweep for C2.
T=${TAG:-r03_m}
mkdir -p gpurun_out
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'verified', d['verified_against_golden'])"; }
for c in C4 C3 C2; do for t in 0 2 3 4 5 6; do
  python bench.py --no-cpu-baseline --no-moving-view --config $c --tile $t 2>/dev/null | line "$c --tile $t"
done; done
for f in 2 3 4 5 6 8; do python bench.py --no-cpu-baseline --no-moving-view --frames-in-flight $f 2>/dev/null | line "C2 --frames-in-flight $f"; done

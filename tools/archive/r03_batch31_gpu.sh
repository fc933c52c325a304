#!/bin/bash
# Round 3, batch 31: grids of several dispatch rounds go heaviest first while the caller renders on one stream.  Parity, then bench lines.
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
for c in C5 C3 C4 C2; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); m=d['timing']['moving_view']; print('$c', 'in flight', round(1e3*d['ms_per_step'],2), 'alone', round(1e3*d['roofline']['kernel_ms'],2), 'moving', round(1e3*m['in_flight_ms_per_frame'],2), round(1e3*m['alone_ms_per_frame'],2), 'verified', d.get('verified'), d['config']['kernel'])"; done

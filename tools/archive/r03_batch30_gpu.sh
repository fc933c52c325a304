#!/bin/bash
# Round 3, batch 30: the planner picks the sub-tile count of grids of a few dispatch rounds by ceil(rounds) x (sub-tiles + 2)
# (config 3: 10 sub-tiles, 16 x 160, 1.9 rounds instead of 2.3).  Parity, then A/B against the build before.
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
for cfg in "--config C3" "" "--config C4"; do
  echo "== bench $cfg"
  tools/ab_gpu.sh "--no-moving-view --no-verify $cfg" librtx_hip_head.so librtx_hip.so
done
echo "== C3 with the camera turning"
for lib in librtx_hip_head.so librtx_hip.so; do
  RTX_LIB=$lib python bench.py --config C3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); m=d['timing']['moving_view']; print('$lib', 'static', d['ms_per_step'], d['roofline']['kernel_ms'], 'moving', m['in_flight_ms_per_frame'], m['alone_ms_per_frame'], 'verified', d.get('verified'))"
done

#!/bin/bash
# Round 3, batch 16: sub-tile masks (a pass scans only the list entries that can touch its sub-tile).  Parity first, then A/B
# against the build before (librtx_hip_prev.so), separate processes, interleaved.
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
for cfg in "" "--config C3" "--config C1" "--config C4"; do
  echo "== bench $cfg"
  tools/ab_gpu.sh "--no-moving-view --no-verify $cfg" librtx_hip_prev.so librtx_hip.so
done
echo "== moving view C2"
for lib in librtx_hip_prev.so librtx_hip.so; do
  RTX_LIB=$lib python bench.py --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernel_ms'], d['timing'].get('moving_view'))"
done

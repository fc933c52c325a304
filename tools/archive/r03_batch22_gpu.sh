#!/bin/bash
# Round 3, batch 22: the directed test on the build before (must fail) and now (must pass); all GPU tests; A/B timing of the new tile_plane.
set -o pipefail
echo "== directed test, build before (expected: fails)"
RTX_LIB=librtx_hip_prev.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "rolled_camera" 2>&1 | grep -E "AssertionError:|passed|failed" | cut -c1-400
echo "== all GPU tests, this build"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
for cfg in "" "--config C3" "--config C4" "--config C5" "--config C1"; do
  echo "== bench $cfg"
  tools/ab_gpu.sh "--no-moving-view --no-verify $cfg" librtx_hip_prev.so librtx_hip.so
done

#!/bin/bash
# Round 3, batch 17: side planes of the culling pyramids from the edge basis (y P + Qr) instead of fp32 cross products of corner
# directions.  (1) the new 8K test must FAIL on the build before (librtx_hip_prev.so) and pass now; (2) the probe; (3) all GPU
# tests; (4) A/B timing.
set -o pipefail
echo "== the new test on the build before (expected: fails)"
RTX_LIB=librtx_hip_prev.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k thin_tiles 2>&1 | tail -8
echo "== the probe, this build"
timeout -k 10 600 python tools/wide_view_cull_gpu.py 2>&1 | grep -v "mismatching pixels: 0$" | tail -12
echo "== all GPU tests"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 || exit 1
for cfg in "" "--config C3" "--config C4" "--config C5"; do
  echo "== bench $cfg"
  tools/ab_gpu.sh "--no-moving-view --no-verify $cfg" librtx_hip_prev.so librtx_hip.so
done

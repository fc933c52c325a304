#!/bin/bash
# Round 3, batch 19: the 8K probe again with the poison fill ordered before the launch (batches 17/18 raced the two), on the build
# before the edge basis and on this one; then the new test on both.
for lib in librtx_hip_prev.so librtx_hip.so; do
  echo "== probe, $lib"
  RTX_LIB=$lib timeout -k 10 400 python tools/wide_view_cull_gpu.py 2>&1 | grep -v "mismatching pixels: 0$" | cut -c1-700 | tail -12
done
for lib in librtx_hip_prev.so librtx_hip.so; do
  echo "== test, $lib"
  RTX_LIB=$lib timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k thin_tiles 2>&1 | grep -E "AssertionError:|passed|failed" | cut -c1-500
done

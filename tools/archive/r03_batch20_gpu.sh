#!/bin/bash
# Round 3, batch 20: large spheres far off the axis (where the planes' error is not hidden by the margin's kappa |O|^2 term), 4K and 8K,
# turned cameras: the build before the edge basis against this one.
for lib in librtx_hip_prev.so librtx_hip.so; do
  echo "== probe --wide-scene --huge, $lib"
  RTX_LIB=$lib timeout -k 10 500 python tools/wide_view_cull_gpu.py --wide-scene --huge 2>&1 | grep -v "mismatching pixels: 0$" | cut -c1-420 | tail -30
done

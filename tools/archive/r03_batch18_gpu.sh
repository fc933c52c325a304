#!/bin/bash
# Round 3, batch 18: which part of the pyramids loses pixels at 8K with a turned camera?  The probe on variant builds.
for lib in librtx_hip.so librtx_hip_delta.so librtx_hip_noaxis.so librtx_hip_norow.so librtx_hip_nocol.so; do
  echo "== $lib"
  RTX_LIB=$lib timeout -k 10 300 python tools/wide_view_cull_gpu.py 2>&1 | grep -v "mismatching pixels: 0$" | cut -c1-600 | tail -8
done

#!/bin/bash
# Round 3, batch 21: the directed scene (tools/wide_view_directed_gpu.py) on the build before the edge basis and on this one.
for lib in librtx_hip_prev.so librtx_hip.so; do
  echo "== directed scene, $lib"
  RTX_LIB=$lib timeout -k 10 500 python tools/wide_view_directed_gpu.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-300
done

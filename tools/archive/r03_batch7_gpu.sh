#!/bin/bash
# Round 3, batch 7: what configuration do the heavy views want?  C2's scene, yaw sweep, one launch alone.
T=${TAG:-r03_i}
mkdir -p gpurun_out
for opts in "" "--subtiles=2" "--subtiles=3" "--subtiles=2 --two-level" "--subtiles=3 --two-level" "--two-level" "--subtiles=2 --two-level --refine=1" "--subtiles=4 --two-level --refine=1"; do
  echo "=== $opts"
  python tools/worst_view_gpu.py --coarse $opts 2>&1 | grep -v amdgpu.ids | sed 's/rtx_trace<RTX_K_RGB_ASCII,//'
done > gpurun_out/${T}_worst_view_configs.txt 2>&1
cat gpurun_out/${T}_worst_view_configs.txt

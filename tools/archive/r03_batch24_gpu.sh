#!/bin/bash
# Round 3, batch 24: where do config 5's +1.5 % of the edge-basis build come from?  prev = before the change; oldplane = the new KArgs
# with the old cross-product planes; inkernel = the basis computed in the kernel in fp32 (no kernel arguments); this build.
for cfg in "--config C5" ""; do
  echo "== bench $cfg"
  tools/ab_gpu.sh "--no-moving-view --no-verify $cfg" librtx_hip_prev.so librtx_hip_oldplane.so librtx_hip_inkernel.so librtx_hip.so
done

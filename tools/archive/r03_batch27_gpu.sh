#!/bin/bash
# Round 3, batch 27: the dispatch-order lookup as an explicit scalar load (sload) against the form the compiler picks behind the volatile asm (spec).
for cfg in "" "--config C5"; do
  echo "== bench $cfg"
  tools/ab_gpu.sh "--no-moving-view --no-verify $cfg" librtx_hip_cam.so librtx_hip_spec.so librtx_hip_sload.so
done

#!/bin/bash
# Round 3, batch 25: oriented edge planes + one camera plane per frame (no per-tile axis plane, four planes per REFINE region and
# per cell).  All GPU tests, the fuzzer for 2 minutes, then A/B against the build before the edge basis.
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
timeout -k 10 300 python tools/fuzz_cull_gpu.py 120 200000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fuzz2.log | grep -E "DIFF|fuzz:" | cut -c1-600
for cfg in "--config C5" "" "--config C3" "--config C4"; do
  echo "== bench $cfg"
  tools/ab_gpu.sh "--no-moving-view --no-verify $cfg" librtx_hip_prev.so librtx_hip.so
done

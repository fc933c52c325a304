#!/bin/bash
# Round 3, batch 26: the head of the workgroup: first list entries and their geometry requested beside the cell's count, the
# prologue's kernel arguments fetched in one batch.  Parity first, then A/B against the camera-plane build (librtx_hip_cam.so).
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
for cfg in "" "--config C5" "--config C3" "--config C4"; do
  echo "== bench $cfg"
  tools/ab_gpu.sh "--no-moving-view --no-verify $cfg" librtx_hip_cam.so librtx_hip.so
done

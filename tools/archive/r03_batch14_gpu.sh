#!/bin/bash
# Round 3, batch 14: the trace kernels' arguments as before the sorted copies (the host points sph_geom / sph_od at whichever arrays they read);
# A/B against the build before the sorted store.
T=${TAG:-r03_r}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/${T}_tests.log
echo "--- A/B C2"; tools/ab_gpu.sh "--no-moving-view" librtx_hip_prev.so librtx_hip.so
echo "--- A/B C5"; tools/ab_gpu.sh "--config C5 --no-moving-view" librtx_hip_prev.so librtx_hip.so
echo "--- A/B C2 BIT_ASCII"; tools/ab_gpu.sh "--no-moving-view --mode BIT_ASCII --no-verify" librtx_hip_prev.so librtx_hip.so

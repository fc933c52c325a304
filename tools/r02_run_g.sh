#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_post.py tests/test_gpu_c4.py -m gpu -x -q > gpurun_out/r02_g_tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_g_tests.log
for cfg in "--config C5 --steps 400" "--config C3 --steps 400" "--config C5 --steps 400 --frames-in-flight 1" ""; do
  echo "=== $cfg"
  tools/ab_gpu.sh "$cfg" librtx_hip_base.so librtx_hip.so 2>&1 | head -4
done

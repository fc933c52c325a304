#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_post.py -m gpu -x -q > gpurun_out/r02_e_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02_e_tests.log
for cfg in "" "--config C4 --steps 200" "--config C3 --steps 400" "--config C5 --steps 400"; do
  echo "=== $cfg"
  tools/ab_gpu.sh "$cfg" librtx_hip_base.so librtx_hip.so librtx_hip_noext.so 2>&1 | head -6
done
echo "=== C2 tile order off (current lib)"
tools/ab_gpu.sh "--tile-order 0" librtx_hip.so 2>&1 | head -2
echo "=== C4 tile order off (current lib)"
tools/ab_gpu.sh "--config C4 --steps 200 --tile-order 0" librtx_hip.so 2>&1 | head -2

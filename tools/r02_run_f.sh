#!/bin/bash
python -m pytest tests/test_gpu_math.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02_f_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02_f_tests.log
for cfg in "" "--frames-in-flight 1" "--config C4 --steps 200" "--config C3 --steps 400" "--config C5 --steps 400"; do
  echo "=== $cfg"
  tools/ab_gpu.sh "$cfg" librtx_hip_base.so librtx_hip.so 2>&1 | head -4
done

#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_post.py -m gpu -x -q > gpurun_out/r02_p_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02_p_tests.log
for cfg in "" "--frames-in-flight 1" "--config C3 --steps 1000" "--config C5 --steps 1000" "--config C4 --steps 300" "--mode BIT_ASCII"; do echo "=== $cfg"; tools/ab_gpu.sh "$cfg" librtx_hip_prev.so librtx_hip.so 2>&1 | head -4; done

#!/bin/bash
timeout -k 10 200 python tools/dbg_gpu.py 2>&1 | grep -v amdgpu | grep -v " 0 bad"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_post.py tests/test_gpu_c4.py -m gpu -x -q > gpurun_out/r02_j_tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_j_tests.log
ab1() { local ARGS=$1; shift
  for lib in "$@"; do printf "%-22s %-30s " $lib "$ARGS"
    RTX_LIB=$lib timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['verified_against_golden'])"
  done; }
ab1 "--config C5" librtx_hip_base.so librtx_hip.so librtx_hip_cl2048.so librtx_hip_cl4096.so
ab1 "--config C5 --subtiles 1" librtx_hip.so librtx_hip_cl4096.so
ab1 "--config C3" librtx_hip_base.so librtx_hip.so librtx_hip_cs1024.so
ab1 "--config C3 --subtiles 8" librtx_hip.so librtx_hip_cs1024.so
ab1 "--two-level 1" librtx_hip.so librtx_hip_cs1024.so
ab1 "--two-level 1 --frames-in-flight 1" librtx_hip.so librtx_hip_cs1024.so

#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_post.py tests/test_gpu_c4.py tests/test_gpu_console.py -m gpu -x -q > gpurun_out/r02_n_tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_n_tests.log
ab1() { local ARGS=$1; shift
  for lib in "$@"; do printf "%-22s %-30s " $lib "$ARGS"
    RTX_LIB=$lib timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['verified_against_golden'])"
  done; }
for r in 1 2; do
ab1 "" librtx_hip_base.so librtx_hip.so
ab1 "--config C3" librtx_hip_base.so librtx_hip.so
done
ab1 "--mode BIT_ASCII" librtx_hip_base.so librtx_hip.so
ab1 "--config C4 --steps 300" librtx_hip_base.so librtx_hip.so

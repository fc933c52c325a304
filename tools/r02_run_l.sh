#!/bin/bash
ab1() { local ARGS=$1; shift
  for lib in "$@"; do printf "%-26s %-30s " $lib "$ARGS"
    RTX_LIB=$lib timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['verified_against_golden'])"
  done; }
ab1 "--config C5" librtx_hip.so librtx_hip_w2048.so librtx_hip_w4096.so librtx_hip_w4096m512.so librtx_hip_w2048c256.so
ab1 "--config C3" librtx_hip.so librtx_hip_w2048.so librtx_hip_w4096.so librtx_hip_w4096m512.so

#!/bin/bash
ab1() { # args, libs...
  local ARGS=$1; shift
  for lib in "$@"; do
    printf "%-28s " $lib
    RTX_LIB=$lib timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['verified_against_golden'])"
  done
}
echo "=== C5"; ab1 "--config C5" librtx_hip.so librtx_hip_cl512.so librtx_hip_cl2048.so librtx_hip_cl4096.so librtx_hip_mi512.so librtx_hip_mi4096.so
echo "=== C5 subtiles 1 / 4"; ab1 "--config C5 --subtiles 1" librtx_hip.so librtx_hip_cl2048.so librtx_hip_cl4096.so; ab1 "--config C5 --subtiles 4" librtx_hip.so librtx_hip_cl2048.so
echo "=== C3"; ab1 "--config C3" librtx_hip.so librtx_hip_cs1024.so librtx_hip_mi512.so
echo "=== C3 subtiles 8 / 2"; ab1 "--config C3 --subtiles 8" librtx_hip.so librtx_hip_cs1024.so;  ab1 "--config C3 --subtiles 2" librtx_hip.so librtx_hip_cs1024.so

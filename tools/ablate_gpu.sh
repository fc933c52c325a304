#!/bin/bash
# Stage-skipping timing experiment (librtx_hip_ablate.so).  Outputs are wrong by construction; only times matter.
export RTX_LIB=librtx_hip_ablate.so
for m in 0 1 2 4 8 16 32 64 48 112 127 3 ; do
  export RTX_ABLATE=$m
  printf "ablate=%-4s " $m
  timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel'], 'kernel_ms', d['roofline']['kernel_ms'])"
done

#!/bin/bash
# Interleaved A/B of library builds in separate processes: tools/ab_gpu.sh "<bench args>" libA.so libB.so ...
ARGS=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    printf "%-28s " $lib
    RTX_LIB=$lib timeout -k 10 120 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
  done
done

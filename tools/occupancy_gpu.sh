#!/bin/bash
# Experiment (librtx_hip_ablate.so): the trace kernel's time as a function of the workgroups a CU may hold at once,
# capped by padding every workgroup with dynamic LDS (20 KB own + pad; 160 KB per CU).  One launch at a time and
# 4 frames in flight.
export RTX_LIB=librtx_hip_ablate.so
for pad in 0 7000 12000 20000 33000 44000; do
  export RTX_LDS_PAD=$pad
  printf "pad=%-6s " $pad
  timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-verify "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step(4 in flight)', d['ms_per_step'], 'kernel_ms(alone)', d['roofline']['kernel_ms'])"
done

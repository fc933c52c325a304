#!/bin/bash
ab1() { local ARGS=$1; shift
  for lib in "$@"; do printf "%-22s %-40s " $lib "$ARGS"
    RTX_LIB=$lib timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['verified_against_golden'])"
  done; }
ab1 "--config C5 --refine 0" librtx_hip.so
ab1 "--config C5 --refine 0 --subtiles 4" librtx_hip.so
ab1 "--config C5 --refine 1 --subtiles 3" librtx_hip.so
ab1 "--config C5 --tile 4" librtx_hip.so
ab1 "--config C5 --tile 3" librtx_hip.so
ab1 "--config C5 --tile-order 16" librtx_hip.so
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_i_c5/trace -- python3 bench.py --config C5 --steps 30 --warmup 5 --no-cpu-baseline --frames-in-flight 1 > gpurun_out/r02_i_c5.log 2>&1
cat gpurun_out/prof_r02_i_c5/trace/*/*kernel_stats.csv | head -5

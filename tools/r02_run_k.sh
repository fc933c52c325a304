#!/bin/bash
export TMPDIR=/tmp
for c in C5 C3; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_k_$c/trace -- python3 bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --frames-in-flight 1 > gpurun_out/r02_k_$c.log 2>&1
cat gpurun_out/prof_r02_k_$c/trace/*/*kernel_stats.csv | head -3
done

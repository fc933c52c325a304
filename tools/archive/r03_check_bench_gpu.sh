#!/bin/bash
# Round 3: the restructured bench.py on the one-GPU box -- the default line (with timing.moving_view), the driver's short form, and the
# N>1 code through its own launcher at world_size 1 (main config + the C4 sub-record + cpu_baseline).
set -o pipefail
mkdir -p gpurun_out
python bench.py > gpurun_out/r03_b_bench_c2.json 2> gpurun_out/r03_b_bench_c2.err; echo "default rc $?"; cat gpurun_out/r03_b_bench_c2.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_b_bench_c2_driver.json 2>/dev/null; echo "driver form rc $?"
RTX_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_b_force_dist_default.json 2> gpurun_out/r03_b_force_dist_default.err; echo "force-dist default (C2 + C4 sub-record + cpu baseline) rc $?"
tail -3 gpurun_out/r03_b_force_dist_default.err
cat gpurun_out/r03_b_force_dist_default.json
tools/force_dist_short_gpu.sh > gpurun_out/r03_b_force_dist_short.txt 2>&1; cat gpurun_out/r03_b_force_dist_short.txt

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 tools/microbench/store_flush > gpurun_out/r02_b_store_flush.txt 2>&1; echo "store_flush rc $?"
cat gpurun_out/r02_b_store_flush.txt
timeout -k 10 400 tools/ablate_gpu.sh --frames-in-flight 1 > gpurun_out/r02_b_ablation_c2.txt 2>&1; echo "ablate rc $?"
cat gpurun_out/r02_b_ablation_c2.txt
timeout -k 10 100 python tools/stamps_gpu.py 4 > gpurun_out/r02_b_stamps_sub4.txt 2>&1; echo "stamps rc $?"
cat gpurun_out/r02_b_stamps_sub4.txt

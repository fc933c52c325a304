#!/bin/bash
# Round 3, batch 23: the culling fuzzer (tools/fuzz_cull_gpu.py), 2.5 minutes, other seeds, progress line every 20 seeds.
timeout -k 10 700 python tools/fuzz_cull_gpu.py 150 100000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fuzz1.log | grep -E "DIFF|fuzz:|\.\.\. seed .*0," 

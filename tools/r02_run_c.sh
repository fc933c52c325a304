#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_order or c2_full or default_scene or c1_all" > gpurun_out/r02_c_tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_c_tests.log
for args in "--tile-order 0 --frames-in-flight 1" "--tile-order 16 --frames-in-flight 1" "--tile-order 0" "--tile-order 16" "--tile-order 1 --frames-in-flight 1" ; do
  echo "== $args"
  timeout -k 10 120 python bench.py --steps 500 --warmup 50 --no-cpu-baseline $args 2>gpurun_out/r02_c_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'verified', d['verified_against_golden'])" || tail -5 gpurun_out/r02_c_err.txt
done
RTX_LIB=librtx_hip.so timeout -k 10 100 python tools/stamps_gpu.py 4 2>&1 | tail -3

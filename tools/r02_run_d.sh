#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_parity.py tests/test_gpu_post.py -m gpu -x -q > gpurun_out/r02_d_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02_d_tests.log
for args in "--frames-in-flight 1" "" "--config C3 --steps 200" "--config C5 --steps 200" "--config C4 --steps 100" "--mode BIT_ASCII"; do
  echo "== $args"
  timeout -k 10 120 python bench.py --steps 500 --warmup 50 --no-cpu-baseline $args 2>gpurun_out/r02_d_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'verified', d['verified_against_golden'])" || tail -5 gpurun_out/r02_d_err.txt
done

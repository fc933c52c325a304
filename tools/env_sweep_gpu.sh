#!/bin/bash
# tools/env_sweep_gpu.sh "<bench args>" VAR v1 v2 ...   : bench.py under different values of one env var
ARGS=$1; VAR=$2; shift; shift
for v in "$@"; do
  printf "%s=%-6s " $VAR $v
  env $VAR=$v timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms', d['roofline']['kernel_ms'])"
done

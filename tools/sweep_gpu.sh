#!/bin/bash
# Quick A/B on the GPU box: bench.py over kernel variants (no CPU baseline), one JSON line each.
for args in "$@"; do
  echo "== $args"
  timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'Mrays/s', d['value'])"
done

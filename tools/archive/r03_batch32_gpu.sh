#!/bin/bash
# Round 3, batch 32: heaviest first WITHIN each XCD's share of a two-level grid (rtx_order_tiles with the static XCD order as base).
set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
for c in C5 C3; do
  for o in -2 16 0; do
    python bench.py --config $c --no-cpu-baseline --no-moving-view --tile-order $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c tile-order $o:', 'in flight', round(1e3*d['ms_per_step'],2), 'alone', round(1e3*d['roofline']['kernel_ms'],2), d['config']['kernel'])"
  done
done
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_b32/pmc_rd -- python3 bench.py --config C5 --steps 200 --warmup 40 --frames-in-flight 1 --no-cpu-baseline --no-moving-view --no-verify > /dev/null 2>&1; echo "pmc rc $?"
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('gpurun_out/prof_b32/pmc_rd/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rtx_trace' in r.get('Kernel_Name','') and r.get('Counter_Name')=='FETCH_SIZE': rows.append(float(r['Counter_Value']))
if rows:
    rows=rows[len(rows)//2:]
    print('C5 one stream: FETCH_SIZE x2 per launch (KiB counter, doubled as tools/make_counters.py does), later half of the launches: %.2f MB' % (2.0 * 1024.0 * sum(rows) / len(rows) / 1e6))
PY

#!/bin/bash
# Minimize from words, one launch (default) against three (--minimize-chain): per-kernel times inside the Update loop (rocprofv3
# kernel trace, the program directly after `--`) and the loop's own figures.   tools/minimize_ab_gpu.sh [TAG] [forms...]
set -o pipefail
TAG=${1:-min_ab}; shift
FORMS=${*:-fused chain}
export TMPDIR=/tmp
mkdir -p gpurun_out
for form in $FORMS; do
  extra=""; [ $form = chain ] && extra="--minimize-chain"
  for what in update update-async; do
    python bench.py --what $what --physics --no-cpu-baseline $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$form $what ms_per_step', d['ms_per_step'])"
  done
  rm -rf gpurun_out/prof_${TAG}_$form
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$form -o t -- python bench.py --what update --physics --no-cpu-baseline $extra > gpurun_out/${TAG}_$form.log 2>&1
  f=gpurun_out/prof_${TAG}_$form/t_kernel_stats.csv
  [ -f "$f" ] && cp "$f" gpurun_out/${TAG}_${form}_kernel_stats.csv && python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'rtx' in r['Name']: print('  %-70s %6s avg %8.2f min %8.2f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1000, float(r['MinNs'])/1000))"
done

#!/bin/bash
# Per-stage VALU instruction counts: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU under each ablation mask.
export RTX_LIB=librtx_hip_ablate.so TMPDIR=/tmp
OUT=$PWD/gpurun_out/ablate_pmc; mkdir -p $OUT
for m in 0 1 2 4 8 16 32 64 127; do
  export RTX_ABLATE=$m
  rm -rf $OUT/m$m
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/m$m -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 "$@" > $OUT/m$m.log 2>&1
  python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for p in glob.glob("$OUT/m$m/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rtx_trace" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate=%-4s"%"$m", {k: round(sum(v)/len(v)/32400.0,1) for k,v in acc.items()}, "(per pixel-wave)")
PY
done

#!/bin/bash
# Lane utilisation of the trace kernel: active lanes per executed VALU instruction
# (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, rocprofv3's AvgNumActiveThreads), the whole kernel and -- through the
# experiment build's stage masks (32: no shading, 96: no shading and no encoding, 4: no candidate scan) -- stage by stage.
#   tools/lane_util_gpu.sh [configs...]        (default: C2 C3 C5)
# Counters only (no tracing domains beside --pmc).  Raw output: gpurun_out/lane_util/; the digest is printed.
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/lane_util; mkdir -p "$OUT"
CONFIGS=${*:-C2 C3 C5}
digest() { # dir label
python3 - "$1" "$2" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rtx_trace" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    act, thr = m.get("SQ_ACTIVE_INST_VALU", 0.0), m.get("SQ_THREAD_CYCLES_VALU", 0.0)
    print("%-22s %s launches=%d lanes/instr=%.2f (%.1f %% of 64)  " % (sys.argv[2], k.split("(")[0][-40:], len(next(iter(c.values()))), thr / act if act else -1, 100 * thr / act / 64 if act else -1) +
          " ".join("%s=%.4g" % (n, v) for n, v in sorted(m.items())))
PY
}
for cfg in $CONFIGS; do
  d=$OUT/${cfg}_product; rm -rf "$d"
  timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 \
      --output-format csv -d "$d" -- python3 bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 --no-verify > "$d.log" 2>&1 || { echo "$cfg product pass failed"; tail -3 "$d.log"; exit 1; }
  digest "$d" "$cfg product"
  for m in 0 32 96 100; do
    d=$OUT/${cfg}_ablate$m; rm -rf "$d"
    RTX_LIB=librtx_hip_ablate.so RTX_ABLATE=$m timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
        --output-format csv -d "$d" -- python3 bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 --no-verify > "$d.log" 2>&1 || { echo "$cfg ablate $m failed"; tail -3 "$d.log"; exit 1; }
    digest "$d" "$cfg ablate=$m"
  done
done

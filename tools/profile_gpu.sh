#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats, then PMC counters in separate
# passes (never combined with tracing of other domains), for one bench.py configuration.
#   tools/profile_gpu.sh <tag> [bench.py args...]
# Raw output goes to gpurun_out/prof_<tag>/; tools/summarize_prof.py condenses it into profiles/.
set -o pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-moving-view --no-side-legs --frames-in-flight 1 $*"
run() { # name, rocprof flags...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
run trace --kernel-trace --stats &&
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
run pmc_sq2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE &&
run pmc_wr --pmc WRITE_SIZE &&
run pmc_rd --pmc FETCH_SIZE
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.json" && cat "$OUT/summary.json"

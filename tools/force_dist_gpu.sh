#!/bin/bash
# Walks bench.py's N>1 code on a one-GPU box (world_size 1 over RCCL), through bench.py's own launcher (it starts its rank as a
# child process, as `python bench.py --gpus N` does): tools/force_dist_gpu.sh [bench args]
# The numbers are not a bench line (no peer: the all-to-all is a self-copy, and the one GPU traces, copies and expands
# every frame); every run checks the last assembled frame against the golden hash and runs the check rounds.
export RTX_BENCH_FORCE_DIST=1
run() { printf "%-52s " "$*"
  timeout -k 10 300 python bench.py --gpus 1 --steps 2000 --warmup 100 --no-cpu-baseline --sub-configs none "$@" 2>gpurun_out/force_dist.err | tail -1 | python3 tools/fmt_bench_line.py || tail -3 gpurun_out/force_dist.err; }
run --exchange compact "$@"
run --exchange compact --frames-in-flight 4 "$@"
run --exchange compact --graphs 0 "$@"
run --exchange compact --latency "$@"
run --exchange compact --root fixed "$@"
run --exchange compact --root fixed --latency "$@"
run --exchange compact --root fixed --frames-per-root 4 --latency "$@"
run --exchange rounds "$@"
run --exchange rounds --root fixed "$@"
run --exchange p2p "$@"
run --exchange p2p --root fixed "$@"
unset RTX_BENCH_FORCE_DIST
# the same sharded frame behind the C ABI: one process, a device group (here: logical ranks on the one GPU)
for d in 0 0,0,0,0; do
  printf "%-52s " "native group, devices $d, 8 frames per call"
  python bench.py --native --native-devices $d --steps 2000 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('verified', d.get('verified_against_golden'), 'us/frame %.2f' % (1e3*d['ms_per_step']), '| update %.3f ms' % d['end_to_end']['ms_per_update_blocking'])"
done
printf "%-52s " "plain N=1 path (4 frames in flight)"
python bench.py --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('verified', d.get('verified_against_golden'), 'us/frame %.2f' % (1e3*d['ms_per_step']), 'one launch alone %.2f' % (1e3*d['roofline']['kernel_ms']))"

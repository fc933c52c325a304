#!/bin/bash
# Walks bench.py's N>1 code on a one-GPU box (world_size 1 over RCCL): tools/force_dist_gpu.sh [bench args]
# The numbers are not a bench line (no peer, self-copy all-to-all); --verify checks the assembled frame.
export RTX_BENCH_FORCE_DIST=1
for ex in compact rounds p2p; do
  printf "%-8s " $ex
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 100 --warmup 10 --verify --no-cpu-baseline --exchange $ex "$@" 2>gpurun_out/force_dist_$ex.err | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('verified', d.get('verified_against_golden'), 'ms/frame', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], '|', d['config']['parallelism'][:120])"
done

#!/bin/bash
python -m pytest tests/test_gpu_post.py -m gpu -x -q > gpurun_out/r02_m_tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_m_tests.log
export RTX_BENCH_FORCE_DIST=1
run() { printf "%-50s " "$*"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>gpurun_out/r02_m_err.txt | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('verified', d.get('verified_against_golden'), 'ms/frame', d['ms_per_step'], d['timing'].get('frame_latency'), '|', d['config']['parallelism'][-60:])" || tail -5 gpurun_out/r02_m_err.txt; }
run --exchange compact
run --exchange compact --graphs 0
run --exchange compact --root fixed
run --exchange compact --root fixed --latency
run --exchange compact --latency
run --exchange compact --frames-per-root 32
run --exchange rounds
run --exchange rounds --root fixed
run --exchange p2p --root fixed

# The world_size-1 walk of bench.py at the driver's K (tools/force_dist_gpu.sh for the long form)
export RTX_BENCH_FORCE_DIST=1
run() { printf "%-60s " "$*"
  timeout -k 10 300 python bench.py --gpus 1 --no-cpu-baseline --sub-configs none "$@" 2>gpurun_out/force_dist.err | tail -1 | python3 tools/fmt_bench_line.py || tail -3 gpurun_out/force_dist.err; }
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --frames-per-root 4
run --steps 20 --warmup 5 --frames-per-root 2
run --steps 20 --warmup 5 --frames-per-root 1
run --steps 20 --warmup 5 --exchange p2p
run --steps 20 --warmup 5 --graphs 0
run --steps 200 --warmup 5
run --steps 2000 --warmup 100

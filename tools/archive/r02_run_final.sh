#!/bin/bash
# Round-2 final GPU pass (TAG=r02_f by default): the -m gpu suite, the default bench line, and rocprofv3 summaries of C2 (one launch at a
# time and 4 in flight), C3 and C5 at this build.
set -o pipefail
TAG=${TAG:-r02_f}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/${TAG}_tests.log
tail -3 gpurun_out/${TAG}_tests.log
python bench.py > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; echo "bench rc $?"
cat gpurun_out/${TAG}_bench_c2.json
tools/profile_gpu.sh ${TAG}_c2 > gpurun_out/${TAG}_prof_c2.log 2>&1; echo "prof c2 rc $?"
tools/profile_gpu.sh ${TAG}_c3 --config C3 > gpurun_out/${TAG}_prof_c3.log 2>&1; echo "prof c3 rc $?"
tools/profile_gpu.sh ${TAG}_c5 --config C5 > gpurun_out/${TAG}_prof_c5.log 2>&1; echo "prof c5 rc $?"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c2_inflight4/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_inflight4.log 2>&1; echo "inflight trace rc $?"
python3 tools/overlap_from_trace.py gpurun_out/prof_${TAG}_c2_inflight4/trace > gpurun_out/${TAG}_c2_inflight4_overlap.json; echo "overlap rc $?"
for c in C1 C3 C4 C5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2>gpurun_out/${TAG}_bench_$c.err; echo "$c rc $?"; done
python bench.py --frames-in-flight 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_c2_f1.json 2>/dev/null; echo "f1 rc $?"

#!/bin/bash
# Round-4 final GPU pass (TAG=r04_z by default): the -m gpu suite; the default bench line and the driver's short form; rocprofv3
# summaries (kernel trace + PMC groups, each in its own pass, the program directly after `--`) of C2 in RGB_ASCII and BIT_ASCII (one
# launch at a time, and 4 in flight), C1, C3, C4, C5, of the whole Update in its word form and its record form, and of
# rtx_expand_words; bench lines of the other configs; a rank's slab of eight frames as eight launches and as one; the world-size-1
# walk of the N>1 code; the device group's walk (logical ranks on the one GPU).  Copy what is to be kept into profiles/.
set -o pipefail
TAG=${TAG:-r04_z}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/${TAG}_tests.log
tail -3 gpurun_out/${TAG}_tests.log
python bench.py > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; echo "bench rc $?"
cat gpurun_out/${TAG}_bench_c2.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c2_driver_form.json 2>/dev/null; echo "driver form rc $?"
TAG=$TAG tools/r04_profile_gpu.sh c2 c2bit c1 c3 c4 c5 update updaterec updatecopy
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c2_inflight4/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-moving-view --no-side-legs > gpurun_out/${TAG}_inflight4.log 2>&1; echo "inflight trace rc $?"
python3 tools/overlap_from_trace.py gpurun_out/prof_${TAG}_c2_inflight4/trace > gpurun_out/${TAG}_c2_inflight4_overlap.json; echo "overlap rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_expand/trace -- python3 tools/expand_gpu.py > gpurun_out/${TAG}_expand.log 2>&1; echo "expand trace rc $?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_expand/pmc_wr -- python3 tools/expand_gpu.py > /dev/null 2>&1; echo "expand pmc_wr rc $?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_expand/pmc_rd -- python3 tools/expand_gpu.py > /dev/null 2>&1; echo "expand pmc_rd rc $?"
python3 tools/summarize_prof.py gpurun_out/prof_${TAG}_expand > gpurun_out/${TAG}_expand_summary.json; echo "expand summary rc $?"
for c in C1 C3 C4 C5; do python bench.py --config $c --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_bench_$c.json 2>gpurun_out/${TAG}_bench_$c.err; echo "$c rc $?"; done
python bench.py --frames-in-flight 1 --no-cpu-baseline --no-side-legs > gpurun_out/${TAG}_bench_c2_f1.json 2>/dev/null; echo "f1 rc $?"
python bench.py --what update --no-cpu-baseline > gpurun_out/${TAG}_bench_update.json 2>/dev/null; echo "update rc $?"
python bench.py --what update-async --no-cpu-baseline > gpurun_out/${TAG}_bench_update_async.json 2>/dev/null; echo "update-async rc $?"
python tools/batch_slabs_gpu.py C2 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_batch_slabs.txt; echo "batch slabs rc $?"; cat gpurun_out/${TAG}_batch_slabs.txt
RTX_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_force_dist_default.json 2> gpurun_out/${TAG}_force_dist_default.err; echo "force-dist default rc $?"
tools/force_dist_gpu.sh > gpurun_out/${TAG}_force_dist.txt 2>&1; echo "force-dist walk rc $?"; cat gpurun_out/${TAG}_force_dist.txt
tools/native_walk_gpu.sh > gpurun_out/${TAG}_native_walk.txt 2>&1; echo "native walk rc $?"; cat gpurun_out/${TAG}_native_walk.txt
python tools/moving_camera_gpu.py 0 0.001 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_moving.txt; cat gpurun_out/${TAG}_moving.txt

#!/bin/bash
# Round-3 final GPU pass (TAG=r03_z by default): the -m gpu suite; the default bench line and the driver's short form; rocprofv3 summaries
# (kernel trace + PMC groups, each in its own pass, program directly after `--`) of C2 (one launch at a time, and 4 in flight), C1, C3, C4, C5,
# of the whole Update (--what update: rtx_min_count / rtx_min_scatter / rtx_update_spheres) and of rtx_expand_words; bench lines of the other
# configs; the world-size-1 walk of the N>1 code with the C4 sub-record.  Copy what is to be kept from gpurun_out/ into profiles/.
set -o pipefail
TAG=${TAG:-r03_z}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/${TAG}_tests.log
tail -3 gpurun_out/${TAG}_tests.log
python bench.py > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; echo "bench rc $?"
cat gpurun_out/${TAG}_bench_c2.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c2_driver_form.json 2>/dev/null; echo "driver form rc $?"
for c in c2: c1:--config\ C1 c3:--config\ C3 c4:--config\ C4 c5:--config\ C5 update:--what\ update\ --physics; do
  name=${c%%:*}; args=${c#*:}
  tools/profile_gpu.sh ${TAG}_$name $args > gpurun_out/${TAG}_prof_$name.log 2>&1; echo "prof $name rc $?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c2_inflight4/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-moving-view > gpurun_out/${TAG}_inflight4.log 2>&1; echo "inflight trace rc $?"
python3 tools/overlap_from_trace.py gpurun_out/prof_${TAG}_c2_inflight4/trace > gpurun_out/${TAG}_c2_inflight4_overlap.json; echo "overlap rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_expand/trace -- python3 tools/expand_gpu.py > gpurun_out/${TAG}_expand.log 2>&1; echo "expand trace rc $?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_expand/pmc_wr -- python3 tools/expand_gpu.py > /dev/null 2>&1; echo "expand pmc_wr rc $?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_expand/pmc_rd -- python3 tools/expand_gpu.py > /dev/null 2>&1; echo "expand pmc_rd rc $?"
python3 tools/summarize_prof.py gpurun_out/prof_${TAG}_expand > gpurun_out/prof_${TAG}_expand/summary.json; echo "expand summary rc $?"
for c in C1 C3 C4 C5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2>gpurun_out/${TAG}_bench_$c.err; echo "$c rc $?"; done
python bench.py --frames-in-flight 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_c2_f1.json 2>/dev/null; echo "f1 rc $?"
python bench.py --what update --no-cpu-baseline > gpurun_out/${TAG}_bench_update.json 2>/dev/null; echo "update rc $?"
python bench.py --what update-async --no-cpu-baseline > gpurun_out/${TAG}_bench_update_async.json 2>/dev/null; echo "update-async rc $?"
RTX_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_force_dist_default.json 2> gpurun_out/${TAG}_force_dist_default.err; echo "force-dist default rc $?"
tools/force_dist_gpu.sh > gpurun_out/${TAG}_force_dist.txt 2>&1; echo "force-dist walk rc $?"; cat gpurun_out/${TAG}_force_dist.txt
python tools/moving_camera_gpu.py 0 0.001 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_moving.txt; cat gpurun_out/${TAG}_moving.txt
python tools/worst_view_gpu.py --coarse 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_worst_view.txt; cat gpurun_out/${TAG}_worst_view.txt

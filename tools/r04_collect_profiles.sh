#!/bin/bash
# After `TAG=r04_z tools/r04_run_final.sh` on the GPU box: copy what is to be kept from gpurun_out/ into profiles/ and rebuild
# profiles/counters.json (per-launch PMC averages + the hash of the sources they were taken on) from the summaries.
set -e
TAG=${TAG:-r04_z}
cd "$(dirname "$0")/.."
for n in c1 c2 c2bit c3 c4 c5 update updaterec updatecopy; do
  [ -f gpurun_out/${TAG}_${n}_kernel_stats.csv ] && cp gpurun_out/${TAG}_${n}_kernel_stats.csv profiles/
  [ -f gpurun_out/${TAG}_${n}_summary.json ] && cp gpurun_out/${TAG}_${n}_summary.json profiles/
done
for n in expand c2_inflight4; do
  f=$(ls gpurun_out/prof_${TAG}_$n/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" profiles/${TAG}_${n}_kernel_stats.csv
done
[ -f gpurun_out/${TAG}_expand_summary.json ] && cp gpurun_out/${TAG}_expand_summary.json profiles/
for f in bench_c2 bench_c2_driver_form bench_c2_f1 bench_C1 bench_C3 bench_C4 bench_C5 bench_update bench_update_async force_dist_default c2_inflight4_overlap; do
  [ -f gpurun_out/${TAG}_$f.json ] && cp gpurun_out/${TAG}_$f.json profiles/${TAG}_$f.json
done
for f in force_dist native_walk batch_slabs moving fuzz_tail; do [ -f gpurun_out/${TAG}_$f.txt ] && cp gpurun_out/${TAG}_$f.txt profiles/${TAG}_$f.txt; done
tail -4 gpurun_out/${TAG}_tests.log > profiles/${TAG}_gpu_tests_tail.txt
python3 tools/make_counters.py profiles/${TAG}_c2_summary.json 'C2_RGB_ASCII_rtx_trace<RTX_K_RGB_ASCII,true>' 'rtx_trace<2, true, 0, false' > /dev/null
python3 tools/make_counters.py profiles/${TAG}_c2bit_summary.json 'C2_BIT_ASCII_rtx_trace<RTX_K_BIT_ASCII,true>' 'rtx_trace<0, true, 0, false' > /dev/null
python3 tools/make_counters.py profiles/${TAG}_c3_summary.json 'C3_RGB_ASCII_rtx_trace<RTX_K_RGB_ASCII,true>' 'rtx_trace<2, true, 0, false' > /dev/null
python3 tools/make_counters.py profiles/${TAG}_c4_summary.json 'C4_RGB_ASCII_rtx_trace<RTX_K_RGB_ASCII,true>' 'rtx_trace<2, true, 0, false' > /dev/null
python3 tools/make_counters.py profiles/${TAG}_c5_summary.json 'C5_RGB_ASCII_rtx_trace<RTX_K_RGB_ASCII,true,refine>' 'rtx_trace<2, true, 0, true' > /dev/null
python3 tools/make_counters.py profiles/${TAG}_c1_summary.json 'C1_RGB_ASCII_rtx_trace<RTX_K_RGB_ASCII,false>' 'rtx_trace<2, false, 0, false' > /dev/null
python3 - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
c = json.load(open("profiles/counters.json"))
now = bench.csrc_sha256()
for k, v in sorted(c.items()):
    print("%-60s hash %s %s  avg %.2f us  total %.2f MB" % (k, (v.get("csrc_sha256") or "none")[:12], "(current)" if v.get("csrc_sha256") == now else "(STALE)", v.get("avg_us") or 0, (v.get("total_bytes") or 0) / 1e6))
PY

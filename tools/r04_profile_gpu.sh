#!/bin/bash
# Round-4 profiling pass (TAG=r04_a by default): rocprofv3 summaries (kernel trace + PMC groups, each in its own pass, the program
# directly after `--`) of C2 in RGB_ASCII and BIT_ASCII, C5, and the whole Update (--what update: trace as pixel words,
# rtx_minw_count / rtx_minw_scatter, rtx_update_spheres).  Copy what is to be kept from gpurun_out/ into profiles/.
#   tools/r04_profile_gpu.sh [names...]     names from: c2 c2bit c5 update updaterec updatecopy c1 c3 c4
set -o pipefail
TAG=${TAG:-r04_a}
mkdir -p gpurun_out
export TMPDIR=/tmp
NAMES=${*:-c2 c2bit c5 update}
for name in $NAMES; do
  case $name in
    c2) args="" ;;
    c2bit) args="--mode BIT_ASCII" ;;
    c1) args="--config C1" ;;
    c3) args="--config C3" ;;
    c4) args="--config C4" ;;
    c5) args="--config C5" ;;
    update) args="--what update --physics" ;;
    updaterec) args="--what update --physics --update-records" ;;
    updatecopy) args="--what update --physics --update-host-write 0" ;;   # the copy form: the Minimize launch at HBM speed, not at PCIe speed
    *) echo "unknown name $name"; continue ;;
  esac
  tools/profile_gpu.sh ${TAG}_$name $args > gpurun_out/${TAG}_prof_$name.log 2>&1; echo "prof $name rc $?"
  cp gpurun_out/prof_${TAG}_$name/summary.json gpurun_out/${TAG}_${name}_summary.json 2>/dev/null
  f=$(ls gpurun_out/prof_${TAG}_$name/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_${name}_kernel_stats.csv
done

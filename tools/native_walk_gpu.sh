#!/bin/bash
# The device group behind the C ABI (rtx_group_create) on the one-GPU box: `bench.py --native` with device lists that repeat
# device 0 (1, 2, 4, 8 logical ranks), by frames per rtx_submit_frames call.  Every rank shares the one GPU, so the figures show
# the host and exchange overhead of the sharded path -- what more GPUs would have to win back -- not a speed-up.
#   tools/native_walk_gpu.sh [config] [steps]
CFG=${1:-C2}; STEPS=${2:-192}
for d in 0 0,0 0,0,0,0 0,0,0,0,0,0,0,0; do
  for m in 1 8 16; do
   for t in 1 0; do
    [ "$t" = 0 ] && [ "$m" = 16 ] && continue
    printf "threads %s  " $t
    python bench.py --native --config $CFG --native-devices $d --native-frames $m --native-threads $t --no-cpu-baseline --steps $STEPS --warmup 32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
e=d.get('end_to_end') or {}
f=e.get('forms') or {}
print('ranks %d  frames/call %2d  %7.2f us/frame (events)  %7.2f us/frame (wall)  verified %s  gather %d B/frame  update gathered %.3f ms, direct %.3f ms (same bytes: %s)  | %s' % (
    d['logical_ranks'], d['config']['frames_per_call'], d['ms_per_step']*1e3, d['timing']['wall_ms_per_step_median']*1e3,
    d['verified_against_golden'], d['gather_bytes_per_frame'], (f.get('gathered') or {}).get('ms_per_update_blocking', float('nan')),
    (f.get('direct') or {}).get('ms_per_update_blocking', float('nan')), (f.get('direct') or {}).get('same_bytes_as_gathered'), d['config']['parallelism'].split('exchange: ')[1][:40]))
"
   done
  done
done

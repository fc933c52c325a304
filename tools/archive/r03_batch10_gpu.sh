#!/bin/bash
# Round 3, batch 10: dense shading (HitBatch) in the REFINE kernels; A/B against the build before it.
set -o pipefail
T=${TAG:-r03_l}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/${T}_tests.log
echo "--- A/B C5 (6 in flight / alone)"
tools/ab_gpu.sh "--config C5 --no-moving-view" librtx_hip_prev.so librtx_hip.so
echo "--- A/B C5 BIT_ASCII"
tools/ab_gpu.sh "--config C5 --mode BIT_ASCII --no-moving-view --no-verify" librtx_hip_prev.so librtx_hip.so
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); mv=d['timing'].get('moving_view',{}); print('$1:', 'in flight %.2f us' % (1e3*d['ms_per_step']), 'alone %.2f us' % (1e3*d['roofline']['kernel_ms']), 'verified', d['verified_against_golden'], 'moving in flight / alone', mv.get('in_flight_ms_per_frame'), mv.get('alone_ms_per_frame'), d['config']['kernel'])"; }
for c in C2 C5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tee gpurun_out/${T}_bench_$c.json | line $c; done
python tools/moving_camera_gpu.py 0.001 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_moving.txt
python tools/worst_view_gpu.py --coarse 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_worst_view.txt

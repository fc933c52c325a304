#!/bin/bash
for cfg in "" "--config C3 --steps 1000" "--frames-in-flight 1"; do echo "=== $cfg"; tools/ab_gpu.sh "$cfg" librtx_hip_noplanecull.so librtx_hip.so 2>&1 | head -6; done

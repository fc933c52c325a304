#!/bin/bash
# The bench lines of the final pass again (after profiles/counters.json was rebuilt for the current sources, so that roofline.traffic is filled in).
TAG=${TAG:-r03_z}
python bench.py > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; echo "bench rc $?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c2_driver_form.json 2>/dev/null; echo "driver form rc $?"
for c in C1 C3 C4 C5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2>gpurun_out/${TAG}_bench_$c.err; echo "$c rc $?"; done
python bench.py --frames-in-flight 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_c2_f1.json 2>/dev/null; echo "f1 rc $?"

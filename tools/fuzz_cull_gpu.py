#!/usr/bin/env python3
"""Randomised hunt for frames on which a culling plan differs from the brute kernel (every pixel tests every object,
RayTracing.cu:100-136): the long-running front end of tests/fuzz_cases.py (a bounded, fixed-seed share of the same cases runs
inside `pytest -m gpu`: tests/test_gpu_fuzz.py).  Prints one line per finding with the seed that reproduces it.

  python tools/fuzz_cull_gpu.py [seconds] [first_seed] [--physics] [--oracle-rows=N]
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import fuzz_cases as F  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
budget = float(args[0]) if len(args) > 0 else 120.0
seed0 = int(args[1]) if len(args) > 1 else 1
PHYSICS = "--physics" in sys.argv
ORACLE_ROWS = max([int(a.split("=", 1)[1]) for a in sys.argv if a.startswith("--oracle-rows=")] + [0])

t_end = time.time() + budget
seed, bad = seed0, 0
stats = {}
bufs = F.Buffers(torch)
while time.time() < t_end:
    for line in F.cull_case(R, torch, bufs, seed, physics=PHYSICS, oracle_rows=ORACLE_ROWS, stats=stats):
        bad += 1
        print("DIFF " + line, flush=True)
    if seed % 20 == 0:
        print("... seed %d, %d frames, %d findings" % (seed, stats.get("frames", 0), bad), flush=True)
    seed += 1
print("fuzz: seeds %d..%d, %d frames compared with the brute kernel, %d rows with the oracle, %d findings; kernels %r" % (
    seed0, seed - 1, stats.get("frames", 0), stats.get("oracle_rows", 0), bad, stats.get("kernels", {})))

#!/usr/bin/env python3
"""Randomised check of the paths around the trace kernel against the whole frame rendered in one launch: the same frame as
row slabs (the unit a sharded run renders: random cuts, every slab its own launch), and as compact pixel words expanded into
records (what travels between GPUs), Minimize from the words against Minimize of the records, and a three-frame slab call (the
batched kernel where the plan allows) against a launch per frame.  All five character modes.  The long-running front end of tests/fuzz_cases.py
(tests/test_gpu_fuzz.py runs a bounded share inside `pytest -m gpu`).

  python tools/fuzz_paths_gpu.py [seconds] [first_seed] [--oracle-rows=N]
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
ORACLE_ROWS = max([int(a.split("=", 1)[1]) for a in sys.argv if a.startswith("--oracle-rows=")] + [0])

t_end = time.time() + budget
seed, bad = seed0, 0
stats = {}
while time.time() < t_end:
    for line in F.paths_case(R, torch, seed, oracle_rows=ORACLE_ROWS, stats=stats):
        bad += 1
        print("DIFF " + line, flush=True)
    if seed % 50 == 0:
        print("... seed %d, %d comparisons, %d findings" % (seed, stats.get("comparisons", 0), bad), flush=True)
    seed += 1
print("fuzz paths: seeds %d..%d, %d comparisons, %d findings" % (seed0, seed - 1, stats.get("comparisons", 0), bad))

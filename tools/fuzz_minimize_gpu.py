#!/usr/bin/env python3
"""Randomised hunt on the Minimize passes (round 4: both forms as one launch with a look-back, the `lead` form device groups use):
made-up pixel words -- runs of equal colour, misses, empty slots from none to nearly all, W from 1 to 4000, up to 2.5 M slots -- through
rtx_minimize_words and, expanded to records, through rtx_minimize; one launch, the chain of launches, and blocks that give up, all
against the CPU oracle's Minimize; every fourth case a device group of 2..9 logical ranks whose ranks minimise their own rows
(RTX_OPT_GROUP_UPDATE) against a single-device context.  The long-running front end of tests/test_gpu_fuzz.py's two bounded tests.

  python tools/fuzz_minimize_gpu.py [seconds] [seed]
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import oracle as O  # noqa: E402
import util as U  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = R.Context(4000, 700)
t0 = time.time()
cases = groups = findings = 0
while time.time() - t0 < budget:
    cases += 1
    if cases % 4 == 0:
        ranks = int(rng.integers(2, 10))
        W = int(rng.choice([1, 2, 5, 64, 333, int(rng.integers(3, 1200))]))
        H = int(rng.integers(1, 300))
        p = R.camera_params(W, H)
        sph, pl = R.synth_scene(int(rng.integers(1, 1000)), int(rng.integers(1, 600)), 1, p.element1, p.element2)
        with R.Context(W, H, devices=[0] * ranks) as g, R.Context(W, H) as one:
            for c in (g, one):
                c.set_scene(sph, pl)
            g.set_option(R.OPT_GROUP_UPDATE, 1)
            g.set_option(R.OPT_GROUP_THREADS, int(rng.integers(0, 2)))
            g.set_option(R.OPT_MINIMIZE_FUSED, int(rng.choice([-1, 0, 2])))
            mode = int(rng.integers(0, 5))
            a = g.update(p, mode).copy()
            b = one.update(p, mode).copy()
            if a.size != b.size or not np.array_equal(a, b):
                findings += 1
                print("FINDING group: ranks %d, %d x %d, mode %s, case %d of seed %d" % (ranks, W, H, R.MODE_NAMES[mode], cases, seed), flush=True)
        groups += 1
        continue
    W = int(rng.choice([1, 2, 3, 7, 64, 255, 256, 257, 1024, 1920, int(rng.integers(4, 4000))]))
    H = int(rng.integers(1, max(2, min(700, 2500000 // W))))
    mode = int(rng.integers(0, 5))
    holes = float(rng.choice([0.0, 0.0, 0.01, 0.3, 0.9, 0.999]))
    runs = float(rng.choice([0.0, 0.5, 0.9, 0.99]))
    S = 20 if mode >= R.RGB_ASCII else 12
    hw = U.random_words(rng, W, H, runs=runs, holes=holes)
    if mode < R.RGB_ASCII:
        hw = np.where((hw != 0) & (hw != 0xFFFFFFFF), hw & np.uint32(0xFF0000FF), hw).astype(np.uint32)
    frame = np.zeros(20 * W * H, dtype=np.uint8)
    frame[:S * W * H] = U.words_to_records(hw, S, ord("3") if mode in (R.RGB_ASCII, R.BIT_ASCII) else ord("4"))
    want = O.minimize(mode, frame, W, H)
    words = torch.from_numpy(hw.view(np.int32)).cuda()
    recs = torch.from_numpy(frame).cuda()
    for fused in (1, 0, 2):
        ctx.set_option(R.OPT_MINIMIZE_FUSED, fused)
        for form in ("words", "records"):
            dst = torch.full((S * W * H + 16,), 0xEE, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            if form == "words":
                n = ctx.minimize_words(mode, W, H, words.data_ptr(), d_out=dst.data_ptr())
            else:
                n = ctx.minimize(mode, W, H, recs.data_ptr(), dst.data_ptr())
            got = dst.cpu().numpy()
            if n != want.size or not np.array_equal(got[:n], want) or not (got[n:] == 0xEE).all():
                findings += 1
                print("FINDING %s: %d x %d, mode %s, holes %g, runs %g, fused %d, case %d of seed %d" % (form, W, H, R.MODE_NAMES[mode], holes, runs, fused, cases, seed), flush=True)
    if cases % 50 == 0:
        print("... %d cases (%d groups), %d findings" % (cases, groups, findings), flush=True)
print("fuzz minimize: seed %d, %d cases (%d of them device groups), %d findings, fallbacks on purpose %d" % (seed, cases, groups, findings, ctx.get_option(R.STAT_MINIMIZE_FALLBACKS)))
ctx.close()

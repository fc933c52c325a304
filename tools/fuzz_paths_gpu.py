#!/usr/bin/env python3
"""Randomised check of the paths around the trace kernel against the whole frame rendered in one launch: the same frame as
row slabs (the unit a sharded run renders: random cuts, every slab its own launch), and as compact pixel words expanded into
records (what travels between GPUs).  All five character modes, the scenes and cameras of tools/fuzz_cull_gpu.py.

  python tools/fuzz_paths_gpu.py [seconds] [first_seed]
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SIZES = [(1920, 1080), (1280, 720), (640, 360), (333, 77), (2560, 300), (97, 1201), (3840, 2160), (400, 150), (17, 9)]

# general_matrix() and scene() of the culling fuzzer (its module body is a script: take the two functions only)
_src = open(os.path.join(ROOT, "tools", "fuzz_cull_gpu.py")).read()
_ns = {"np": np, "R": R}
exec(_src[_src.index("def general_matrix"):_src.index("t_end = time.time()")], _ns)
general_matrix, scene = _ns["general_matrix"], _ns["scene"]

t_end = time.time() + budget
seed = seed0
frames = bad = 0
while time.time() < t_end:
    g = np.random.default_rng(seed)
    W, H = SIZES[int(g.integers(0, len(SIZES)))]
    pos = [float(v) for v in g.uniform(-30, 30, 3)]
    p = R.camera_params(W, H, pos, (0.0, float(np.pi), 0.0))
    M = general_matrix(g)
    for i in range(3):
        for j in range(3):
            p.inv_v[4 * i + j] = float(M[i, j])
    sph, pl = scene(g, p, np.array([[p.inv_v[4 * i + j] for j in range(3)] for i in range(3)], dtype=np.float64), pos, W, H)
    mode = int(g.integers(0, 5))
    S = 20 if mode >= R.RGB_ASCII else 12
    zt = 0 if mode >= R.RGB_ASCII else R.RENDER_ZERO_TAIL
    want = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
    got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
    words = torch.empty(W * H, dtype=torch.int32, device="cuda")
    c = R.Context(W, H)
    try:
        c.set_scene(sph, pl)
        want.fill_(0xEE)
        got.fill_(0xEE)
        words.fill_(0x5A5A5A5A)
        torch.cuda.synchronize()
        c.render_rows(p, mode, 0, H, d_out=want.data_ptr(), out_row_base=0, flags=zt)
        c.synchronize()
        # (a) the same frame as slabs
        k = int(g.integers(1, min(9, H) + 1))
        cuts = sorted(set([0, H] + [int(v) for v in g.integers(1, H, k - 1)])) if H > 1 else [0, H]
        order = list(range(len(cuts) - 1))
        g.shuffle(order)
        for i in order:
            c.render_rows(p, mode, cuts[i], cuts[i + 1] - cuts[i], d_out=got.data_ptr(), out_row_base=0)
        c.synchronize()
        frames += 1
        if not torch.equal(got[:S * W * H], want[:S * W * H]):
            bad += 1
            print("DIFF seed %d: slabs %r differ from the frame, %dx%d mode %d, %d spheres %d planes (%s)" % (seed, cuts, W, H, mode, len(sph), len(pl), c.last_kernel), flush=True)
        # (b) compact words, expanded
        got.fill_(0xEE)
        torch.cuda.synchronize()
        c.render_rows(p, mode, 0, H, d_out=words.data_ptr(), out_row_base=0, flags=R.RENDER_COMPACT)
        c.synchronize()
        nseg = int(g.integers(1, 5))
        bounds = sorted(set([0, W * H] + [int(v) for v in g.integers(1, W * H, nseg - 1)])) if W * H > 1 else [0, W * H]
        c.expand(mode, words.data_ptr(), got.data_ptr(), [(bounds[i], bounds[i], bounds[i + 1] - bounds[i]) for i in range(len(bounds) - 1)])
        c.synchronize()
        frames += 1
        if not torch.equal(got[:S * W * H], want[:S * W * H]):
            bad += 1
            print("DIFF seed %d: expanded compact words differ from the frame, %dx%d mode %d segments %r (%s)" % (seed, W, H, mode, bounds, c.last_kernel), flush=True)
    finally:
        c.close()
    if seed % 50 == 0:
        print("... seed %d, %d comparisons, %d differing" % (seed, frames, bad), flush=True)
    seed += 1
print("fuzz paths: seeds %d..%d, %d comparisons, %d differing" % (seed0, seed - 1, frames, bad))

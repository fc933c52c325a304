import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
R = importlib.import_module("raytracing-in-windows-console_amd")
for (w, h) in ((7, 513), (9, 513), (16, 513), (7, 257), (333, 77)):
    p = R.camera_params(w, h, (1.0, 2.0, -3.0), (0.1, 3.0, 0.0))
    sph, pl = R.synth_scene(77, 300, 2, p.element1, p.element2)
    ctx = R.Context(3840, 2160)
    ctx.set_scene(sph, pl)
    ctx.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
    want = ctx.render_to_host(p, R.RGB_ASCII)
    ctx.set_option(R.OPT_KERNEL, R.KERNEL_BINNED)
    for two in (0, 1):
        for sub in (1, 2, 4):
            for refine in (0, 1):
                for cap in (0, 1):
                    if two == 0 and cap: continue
                    ctx.set_option(R.OPT_TWO_LEVEL, two); ctx.set_option(R.OPT_SUBTILES, sub); ctx.set_option(R.OPT_REFINE, refine)
                    ctx.set_option(R.OPT_CELL_CAPACITY, cap)
                    got = ctx.render_to_host(p, R.RGB_ASCII)
                    bad = np.flatnonzero(got != want)
                    rows = sorted(set((bad // (20 * w)).tolist()))
                    print("%dx%d two %d sub %d refine %d cap %d: %d bad bytes, rows %s" % (w, h, two, sub, refine, cap, bad.size, (rows[:3], rows[-3:]) if rows else ""))
    ctx.close()

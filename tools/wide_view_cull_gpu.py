#!/usr/bin/env python3
"""Are the culling pyramids sound where they are thinnest?  At 4K / 8K the reference's horizontal tan-extent (element1 =
0.577 H / 100) reaches 12 / 25, so a tile at the frame's left or right edge spans 1e-4 rad horizontally; its top and bottom
planes are cross products of two nearly parallel corner directions.  Rotated cameras (general matrices), culling kernels
against the brute kernel, mismatching pixels counted on the GPU.

  python tools/wide_view_cull_gpu.py [--quick]
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
quick = "--quick" in sys.argv
rng = np.random.default_rng(7)
total_bad = 0
for (W, H, n) in ([(7680, 4320, 1024)] if quick else [(1920, 1080, 1024), (3840, 2160, 4096), (7680, 4320, 1024), (7680, 4320, 16384)]):
    p0 = R.camera_params(W, H)
    sph, pl = R.synth_scene(100 + n, n, 1, p0.element1, p0.element2)
    a, b = R.Context(W, H), R.Context(W, H)
    for c in (a, b):
        c.set_scene(sph, pl)
    b.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
    got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
    want = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
    for view in range(3 if quick else 6):
        rot = (float(rng.uniform(-0.3, 0.3)), float(np.pi + rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.3, 0.3)))
        pos = tuple(float(v) for v in rng.uniform(-2, 2, 3))
        p = R.camera_params(W, H, pos, rot)
        b.render_rows(p, R.RGB_ASCII, 0, H, d_out=want.data_ptr(), out_row_base=0)
        b.synchronize()
        for name, opts in (("auto", {}), ("one-level", {R.OPT_TWO_LEVEL: 0}), ("two-level", {R.OPT_TWO_LEVEL: 1}),
                           ("two-level refine", {R.OPT_TWO_LEVEL: 1, R.OPT_REFINE: 1, R.OPT_SUBTILES: 2})):
            a.set_option(R.OPT_TWO_LEVEL, -1)
            a.set_option(R.OPT_REFINE, -1)
            a.set_option(R.OPT_SUBTILES, 0)
            for k, v in opts.items():
                a.set_option(k, v)
            got.fill_(0xEE)
            a.render_rows(p, R.RGB_ASCII, 0, H, d_out=got.data_ptr(), out_row_base=0)
            a.synchronize()
            diff = (got.view(H, W, 20) != want.view(H, W, 20)).any(dim=2)
            bad = int(diff.sum().item())
            total_bad += bad
            where = ""
            if bad:
                ys, xs = torch.nonzero(diff, as_tuple=True)
                where = " columns %d..%d rows %d..%d" % (int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max()))
            print("%dx%d n=%d view %d rot (%.2f %.2f %.2f) %-18s %-40s mismatching pixels: %d%s" % (W, H, n, view, rot[0], rot[1], rot[2], name, a.last_kernel, bad, where), flush=True)
    a.close()
    b.close()
print("total mismatching pixels:", total_bad)

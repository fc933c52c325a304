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
def wide_scene(W, H, n, seed):
    """Large spheres far off the view axis of the DEFAULT camera (yaw pi): tangent 0.4 element1 .. element1 to either side, 80 .. 200 away,
    radius 1 .. 8.  (App. D scenes shrink their spheres with cos^1.5 towards the frame's edge: the culling margin's kappa |O|^2 term,
    sized for the fp32 test's own sloppiness, then dwarfs any error of the planes.)"""
    g = np.random.default_rng(seed)
    p0 = R.camera_params(W, H)
    xt = g.uniform(0.4 * float(p0.element1), float(p0.element1), n) * g.choice([-1.0, 1.0], n)
    yt = g.uniform(-1.0, 1.0, n) * float(p0.element2)
    L = g.uniform(80.0, 200.0, n)
    d = np.stack([-xt, yt, np.ones(n)], axis=1)      # yaw pi: inverseVMatrix = diag(-1, 1, 1): w = (-vx, vy, 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sph = np.zeros((n, 7), dtype=np.float32)
    sph[:, 0:3] = d * L[:, None]
    sph[:, 3] = g.uniform(1.0, 8.0, n)
    if "--huge" in sys.argv:
        # few, huge and far: the margin's slack kappa |O|^2 / 2r is smallest, and a silhouette that is nearly straight over hundreds of
        # columns crosses the tile rows' boundaries at every depth
        L = g.uniform(200.0, 230.0, n)
        sph[:, 0:3] = d * L[:, None]
        sph[:, 3] = g.uniform(10.0, 25.0, n)
    sph[:, 4:7] = np.floor(g.uniform(1, 256, (n, 3)))
    return sph, np.zeros((0, 11), dtype=np.float32)


adversarial = "--wide-scene" in sys.argv
for (W, H, n) in ([(7680, 4320, 64), (7680, 4320, 300)] if "--huge" in sys.argv else [(7680, 4320, 1024)] if quick else [(1920, 1080, 1024), (3840, 2160, 4096), (7680, 4320, 1024), (7680, 4320, 16384)]):
    p0 = R.camera_params(W, H)
    sph, pl = wide_scene(W, H, n, 100 + n) if adversarial else R.synth_scene(100 + n, n, 1, p0.element1, p0.element2)
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
            torch.cuda.synchronize()   # (the fill runs on torch's stream, the launch on the context's: order them)
            a.render_rows(p, R.RGB_ASCII, 0, H, d_out=got.data_ptr(), out_row_base=0)
            a.synchronize()
            diff = (got.view(H, W, 20) != want.view(H, W, 20)).any(dim=2)
            bad = int(diff.sum().item())
            total_bad += bad
            where = ""
            if bad:
                ys, xs = torch.nonzero(diff, as_tuple=True)
                g3, w3 = got.view(H, W, 20), want.view(H, W, 20)
                lost = int(((g3[..., 2] == ord('4')) & (w3[..., 2] == ord('3')) & diff).sum().item())   # a hit the culling kernel did not see
                y0, x0 = int(ys[0]), int(xs[0])
                where = " columns %d..%d rows %d..%d; %d of them hits the brute kernel sees and this one does not; first (%d, %d): got %r want %r" % (
                    int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max()), lost, x0, y0, bytes(g3[y0, x0].cpu().numpy()), bytes(w3[y0, x0].cpu().numpy()))
            print("%dx%d n=%d view %d rot (%.2f %.2f %.2f) %-18s %-40s mismatching pixels: %d%s" % (W, H, n, view, rot[0], rot[1], rot[2], name, a.last_kernel, bad, where), flush=True)
    a.close()
    b.close()
print("total mismatching pixels:", total_bad)

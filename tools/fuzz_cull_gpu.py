#!/usr/bin/env python3
"""Randomised hunt for frames on which a culling plan differs from the brute kernel (every pixel tests every object,
RayTracing.cu:100-136).  Unlike the parity tests it goes for the corners the directed 8K scene came from: general camera
matrices (roll, slight non-orthonormality), fields of view from a fifth to twice the reference's, frames up to 8K, spheres
that are large and far / tiny / around and behind the camera / containing it, up to 20 planes, every tile shape, lists that
outlive the frame while the camera creeps.  Prints one line per differing frame with the seed that reproduces it.

  python tools/fuzz_cull_gpu.py [seconds] [first_seed] [--physics]
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
PHYSICS = "--physics" in sys.argv
SIZES = [(1920, 1080), (3840, 2160), (7680, 4320), (1280, 720), (640, 360), (333, 77), (2560, 300), (97, 1201)]


def general_matrix(g):
    a, b, c = g.uniform(-0.6, 0.6), g.uniform(0, 2 * np.pi), g.uniform(-0.6, 0.6)
    cx, sx, cy, sy, cz, sz = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    m = np.array([[cy * cz + sy * sx * sz, -cy * sz + sy * sx * cz, sy * cx], [cx * sz, cx * cz, -sx],
                  [-sy * cz + cy * sx * sz, sy * sz + cy * sx * cz, cy * cx]])
    if g.random() < 0.2:
        m = m * (1.0 + g.uniform(-2e-4, 2e-4, (3, 3)))      # slightly off orthonormal (still inside the reuse policy's epsilon)
    return m


def scene(g, p, M, pos, W, H):
    kind = g.integers(0, 5)
    n = int(g.choice([1, 7, 60, 700, 3000, 12000]))
    e1, e2 = float(p.element1), float(p.element2)
    xt = g.uniform(-1.1, 1.1, n) * e1
    yt = g.uniform(-1.1, 1.1, n) * e2
    if kind == 1:                                            # towards the left / right edge
        xt = g.uniform(0.5, 1.05, n) * e1 * g.choice([-1.0, 1.0], n)
    d = np.stack([xt, yt, np.ones(n)], axis=1) @ M.T        # w = M (vx, vy, 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    L = g.uniform(5.0, 240.0, n)
    r = np.abs(g.normal(0, 1, n)) * g.choice([0.02, 0.5, 3.0, 15.0], n) + 1e-3
    if kind == 2:                                            # all around, also behind and containing the camera
        d = g.normal(0, 1, (n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        L = g.uniform(0.0, 60.0, n)
    if kind == 3:                                            # large and far: the margin's slack is smallest
        L = g.uniform(150.0, 245.0, n)
        r = g.uniform(5.0, 30.0, n)
    sph = np.zeros((n, 7), dtype=np.float32)
    sph[:, 0:3] = np.asarray(pos)[None, :] + d * L[:, None]
    sph[:, 3] = r
    sph[:, 4:7] = np.floor(g.uniform(1, 256, (n, 3)))
    npl = int(g.choice([0, 0, 1, 2, 6, 20]))
    pl = np.zeros((npl, 11), dtype=np.float32)
    for i in range(npl):
        c = np.asarray(pos) + (M @ np.array([g.uniform(-1, 1) * e1, g.uniform(-1, 1) * e2, 1.0])) * g.uniform(1, 20)
        nrm = g.normal(0, 1, 3)
        pl[i] = [c[0], c[1], c[2], nrm[0], nrm[1], nrm[2], g.integers(1, 256), g.integers(1, 256), g.integers(1, 256), g.uniform(1, 300), g.uniform(1, 300)]
    return sph, pl


t_end = time.time() + budget
seed = seed0
frames = bad_frames = 0
covered = 0.0          # sum over frames of the fraction of pixels with a visible hit (is the fuzzer looking at anything?)
kernels = {}
bufs = {}
while time.time() < t_end:
    g = np.random.default_rng(seed)
    W, H = SIZES[int(g.integers(0, len(SIZES)))]
    pos = [float(v) for v in g.uniform(-30, 30, 3)]
    p = R.camera_params(W, H, pos, (0.0, float(np.pi), 0.0))
    M = general_matrix(g)
    fov = float(g.choice([1.0, 1.0, 0.2, 0.5, 2.0]))
    p.element1 = float(p.element1) * fov
    p.element2 = float(p.element2) * fov
    for i in range(3):
        for j in range(3):
            p.inv_v[4 * i + j] = float(M[i, j])
    sph, pl = scene(g, p, np.array([[p.inv_v[4 * i + j] for j in range(3)] for i in range(3)], dtype=np.float64), pos, W, H)
    if (W, H) not in bufs:
        bufs[(W, H)] = (torch.empty(20 * W * H, dtype=torch.uint8, device="cuda"), torch.empty(20 * W * H, dtype=torch.uint8, device="cuda"))
    got, want = bufs[(W, H)]
    a, b = R.Context(W, H), R.Context(W, H)
    try:
        for c in (a, b):
            c.set_scene(sph, pl)
        b.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
        a.set_option(R.OPT_KERNEL, R.KERNEL_BINNED)
        opts = {R.OPT_TILE_LOG2_W: int(g.choice([0, 0, 2, 3, 4, 5, 6])), R.OPT_SUBTILES: int(g.choice([0, 0, 1, 2, 3, 4, 5, 8])),
                R.OPT_TWO_LEVEL: int(g.choice([-1, 0, 1])), R.OPT_REFINE: int(g.choice([-1, 0, 1]))}
        for k, v in opts.items():
            a.set_option(k, v)
        mode = int(g.choice([R.RGB_ASCII, R.RGB_ASCII, R.BIT_ASCII, R.RGB_NORMALS]))
        creep = g.random() < 0.5
        # physics: a third of the creeping runs also step their spheres between frames (Sphere::Update, Sphere.cu:15-23: y moves by
        # speed * mover * dt and is clamped to +-10) on both contexts alike; the lists' position budget has to cover it
        physics = creep and PHYSICS and g.random() < 0.6 and len(sph) <= 3000
        if physics:
            movers = g.choice([-1, 1], len(sph))
            speeds = g.uniform(0.5, 4.0, len(sph))
            for i in range(len(sph)):
                for c in (a, b):
                    c.set_sphere_motion(i, int(movers[i]), float(speeds[i]))
        for f in range((8 if physics else 4) if creep else 1):
            if physics:
                dt = float(g.choice([0.004, 0.016, 0.033]))
                for c in (a, b):
                    c.update_objects(dt)
            if f:
                # creep: a small turn about a random axis and a small step, so that lists built for an earlier frame are reused
                w = g.normal(0, 1, 3) * 2e-4
                dR = np.array([[1, -w[2], w[1]], [w[2], 1, -w[0]], [-w[1], w[0], 1]])
                M = M @ dR
                for i in range(3):
                    for j in range(3):
                        p.inv_v[4 * i + j] = float(M[i, j])
                for i in range(3):
                    p.cam_pos[i] = float(p.cam_pos[i]) + float(g.normal(0, 1) * 1e-3)
            got.fill_(0xEE)
            want.fill_(0xEE)
            torch.cuda.synchronize()
            flags = 0 if mode >= R.RGB_ASCII else 1
            b.render_rows(p, mode, 0, H, d_out=want.data_ptr(), out_row_base=0, flags=flags)
            a.render_rows(p, mode, 0, H, d_out=got.data_ptr(), out_row_base=0, flags=flags)
            a.synchronize()
            b.synchronize()
            frames += 1
            kernels[a.last_kernel] = kernels.get(a.last_kernel, 0) + 1
            if frames % 16 == 0:   # (sampled: the count costs a pass over the frame)
                S_ = 20 if mode >= R.RGB_ASCII else 12
                vis = want[:S_ * W * H].view(H, W, S_)[..., 2] == (ord('3') if mode in (R.RGB_ASCII, R.BIT_ASCII) else ord('4'))
                covered += 16.0 * float(vis.float().mean().item()) if mode in (R.RGB_ASCII, R.BIT_ASCII) else 0.0
            if not torch.equal(got, want):
                S = 20 if mode >= R.RGB_ASCII else 12
                diff = (got[:S * W * H].view(H, W, S) != want[:S * W * H].view(H, W, S)).any(dim=2)
                ys, xs = torch.nonzero(diff, as_tuple=True)
                bad_frames += 1
                print("DIFF seed %d frame %d: %dx%d fov x%.1f mode %d, %d spheres %d planes, options %r, %s: %d pixels, rows %d..%d columns %d..%d" % (
                    seed, f, W, H, fov, mode, len(sph), len(pl), opts, a.last_kernel, int(diff.sum()), int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())), flush=True)
    finally:
        a.close()
        b.close()
    if seed % 20 == 0:
        print("... seed %d, %d frames, %d differing" % (seed, frames, bad_frames), flush=True)
    seed += 1
print("fuzz: seeds %d..%d, %d frames compared, %d differing; mean visible-hit coverage of the ASCII-mode frames sampled ~%.0f %%; kernels %r" % (
    seed0, seed - 1, frames, bad_frames, 100.0 * covered / max(1, frames) / 0.75, kernels))

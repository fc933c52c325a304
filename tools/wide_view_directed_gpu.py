#!/usr/bin/env python3
"""A directed scene for the culling pyramids' side planes at 8K (see tools/wide_view_cull_gpu.py for the undirected probe, which
finds nothing: the margin's kappa |O|^2 / 2r term hides a plane error unless a large, far sphere's extreme point falls within a few
rows of a tile boundary at the frame's edge).  For a turned camera the script emulates, in fp32 on the host, the plane the kernels
USED to build for every 16 x 16 tile of the frame's first 16 columns (cross product of two corner directions), picks the tiles whose
own boundary-row pixel rays lie furthest OUTSIDE that plane, and places one sphere per such tile (r = 20, 200 away) whose extreme
point pokes 2.5 rows into the tile.  The brute kernel sees those caps; a culling kernel with the old planes drops them.

  python tools/wide_view_directed_gpu.py        # one level, 16 x 16 tiles (OPT_TILE_LOG2_W 4, OPT_SUBTILES 1), against brute
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

f = np.float32


def directed_scene(R, W, H, rot, pos, want=12, tile=16):
    """(spheres, report): spheres poking into the worst tiles of columns [0, tile) under the old plane formula."""
    p = R.camera_params(W, H, pos, (0.0, float(np.pi), 0.0))
    # a general rotation (pitch, yaw, roll about the view axis) written straight into inverseVMatrix: the reference's own camera keeps
    # m[4] = 0 (no roll: Camera3D's matrix), which happens to keep the old formula's error near 4e-6 rad -- still most
    # of the half pixel -- but the C ABI takes any matrix (rtx_params::inv_v)
    cx, sx, cy, sy, cz, sz = np.cos(rot[0]), np.sin(rot[0]), np.cos(rot[1]), np.sin(rot[1]), np.cos(rot[2]), np.sin(rot[2])
    M = np.array([[cy * cz + sy * sx * sz, -cy * sz + sy * sx * cz, sy * cx], [cx * sz, cx * cz, -sx],
                  [-sy * cz + cy * sx * sz, sy * sz + cy * sx * cz, cy * cx]]).astype(np.float32)
    for i in range(3):
        for j in range(3):
            p.inv_v[4 * i + j] = float(M[i, j])
    for i in range(3):
        p.cam_pos[i] = float(pos[i])
    e1, e2 = f(p.element1), f(p.element2)
    o = np.array(pos, dtype=np.float64)

    def view_dir32(cx, cy):
        vx, vy = f(cx) * e1, f(cy) * e2
        return np.array([f(f(f(M[k, 0] * vx) + f(M[k, 1] * vy)) + M[k, 2]) for k in range(3)], dtype=np.float32)

    def dir64(cx, cy):
        d = M.astype(np.float64) @ np.array([float(cx) * float(e1), float(cy) * float(e2), 1.0])
        return d / np.linalg.norm(d)

    found = []
    for by in range(H // tile):
        col0, row0 = 0, by * tile
        x0, x1 = f(2 * col0 - 1 - W) / f(W), f(2 * (col0 + tile) - 1 - W) / f(W)
        y0, y1 = f(H - 2 * row0 + 1) / f(H), f(H - 2 * (row0 + tile) + 1) / f(H)
        axis = view_dir32(f(0.5) * (x0 + x1), f(0.5) * (y0 + y1)).astype(np.float64)
        for k, (ya, row_in) in ((0, (y0, row0)), (2, (y1, row0 + tile - 1))):
            va, vb = (view_dir32(x0, ya), view_dir32(x1, ya)) if k == 0 else (view_dir32(x1, ya), view_dir32(x0, ya))
            n = np.array([f(f(va[1] * vb[2]) - f(va[2] * vb[1])), f(f(va[2] * vb[0]) - f(va[0] * vb[2])), f(f(va[0] * vb[1]) - f(va[1] * vb[0]))], dtype=np.float64)
            if n @ axis < 0:
                n = -n
            n /= np.linalg.norm(n)
            cy = (H - 2 * row_in) / H
            worst = min(n @ dir64((2 * c - W) / W, cy) for c in range(col0, col0 + tile))
            found.append((worst, by, k))
    found.sort()
    sph = []
    report = []
    used_rows = set()
    for worst, by, k in found:
        if len(sph) >= want or worst > -2.0e-5:
            break
        if any(abs(by - u) < 6 for u in used_rows):
            continue   # keep the spheres' caps apart
        used_rows.add(by)
        row0 = by * tile
        # the TRUE plane of that edge (double), inward normal
        ya = (H - 2 * row0 + 1) / H if k == 0 else (H - 2 * (row0 + tile) + 1) / H
        x0, x1 = (2 * 0 - 1 - W) / W, (2 * tile - 1 - W) / W
        a, b = dir64(x0, ya), dir64(x1, ya)
        nt = np.cross(a, b)
        nt /= np.linalg.norm(nt)
        inside = dir64((x0 + x1) / 2, (H - 2 * (row0 + tile // 2)) / H)
        if nt @ inside < 0:
            nt = -nt
        # extreme point: 2.5 rows inside the tile from that edge, column 8, 200 away
        row_q = row0 + 2.0 if k == 0 else row0 + tile - 1 - 2.0
        dq = dir64((2 * 8 - W) / W, (H - 2 * row_q) / H)
        r, L = 20.0, 200.0
        c = o + L * dq - r * nt
        sph.append([c[0], c[1], c[2], r, 200.0, 60.0 + 10 * len(sph), 90.0])
        report.append("tile row %d, %s edge: old plane leaves its boundary-row rays %.1e rad outside" % (by, "top" if k == 0 else "bottom", -worst))
    return p, np.array(sph, dtype=np.float32).reshape(-1, 7), report


if __name__ == "__main__":
    import torch
    R = importlib.import_module("raytracing-in-windows-console_amd")
    W, H = 7680, 4320
    total = 0
    for rot, pos in (((-0.27, 2.98, 0.21), (0.5, -1.0, 0.3)), ((0.22, 3.3, -0.25), (0.0, 0.0, 0.0)), ((0.1, 2.8, 0.3), (1.0, 2.0, -1.0))):
        p, sph, report = directed_scene(R, W, H, rot, pos)
        print("camera rot %r: %d directed spheres" % (rot, len(sph)))
        for line in report:
            print("   ", line)
        if not len(sph):
            continue
        pl = np.zeros((0, 11), dtype=np.float32)
        got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        want = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        # one sphere per frame: each of them covers the whole height of the first columns (r = 20 at 200 is 0.1 rad; a row is 1e-5)
        for i in range(min(6, len(sph))):
            a, b = R.Context(W, H), R.Context(W, H)
            for c in (a, b):
                c.set_scene(sph[i:i + 1], pl)
            b.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
            a.set_option(R.OPT_KERNEL, R.KERNEL_BINNED)
            a.set_option(R.OPT_TWO_LEVEL, 0)
            a.set_option(R.OPT_TILE_LOG2_W, 4)
            a.set_option(R.OPT_SUBTILES, 1)
            got.fill_(0xEE)
            torch.cuda.synchronize()
            b.render_rows(p, R.RGB_ASCII, 0, H, d_out=want.data_ptr(), out_row_base=0)
            a.render_rows(p, R.RGB_ASCII, 0, H, d_out=got.data_ptr(), out_row_base=0)
            a.synchronize()
            b.synchronize()
            diff = (got.view(H, W, 20) != want.view(H, W, 20)).any(dim=2)
            hit = (want.view(H, W, 20)[..., 2] == ord('3'))
            bad = int(diff.sum().item())
            total += bad
            rows = ""
            if bad:
                ys, xs = torch.nonzero(diff, as_tuple=True)
                rows = ", rows %d..%d columns %d..%d" % (int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()))
            print("   sphere %d, %s: pixels the brute kernel shades: %d (in columns 0..15: %d); pixels that differ: %d%s" % (
                i, a.last_kernel, int(hit.sum().item()), int(hit[:, :16].sum().item()), bad, rows), flush=True)
            a.close()
            b.close()
    print("total differing pixels:", total)

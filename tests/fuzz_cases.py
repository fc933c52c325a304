"""Randomised cases for the culling plans and the paths around the trace kernel (round 3's tools/fuzz_cull_gpu.py and
tools/fuzz_paths_gpu.py as importable functions: those scripts are now the long-running front ends of this module, and
tests/test_gpu_fuzz.py runs a bounded, fixed-seed share of it inside `pytest -m gpu`).

A case is a seed: a frame size, a general camera matrix (roll, slight non-orthonormality -- any matrix is legal through
rtx_params::inv_v), a field of view from a fifth to twice the reference's, a scene (spheres large and far / tiny / around
and behind the camera / containing it, up to 20 planes), a culling plan (tile shape, sub-tile count, one or two levels,
refinement), a mode; optionally a camera that creeps so that cell lists are reused, and physics steps in between
(Sphere::Update, Sphere.cu:15-23).  Checked: the culling kernel against the brute kernel (every pixel tests every object,
RayTracing.cu:100-136), whole frame, on the GPU; and sampled rows of the culling kernel's frame against the CPU oracle.
"""
import numpy as np

import oracle as O
import util as U

CULL_SIZES = [(1920, 1080), (3840, 2160), (7680, 4320), (1280, 720), (640, 360), (333, 77), (2560, 300), (97, 1201)]
PATH_SIZES = [(1920, 1080), (1280, 720), (640, 360), (333, 77), (2560, 300), (97, 1201), (3840, 2160), (400, 150), (17, 9)]


def general_matrix(g):
    a, b, c = g.uniform(-0.6, 0.6), g.uniform(0, 2 * np.pi), g.uniform(-0.6, 0.6)
    cx, sx, cy, sy, cz, sz = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    m = np.array([[cy * cz + sy * sx * sz, -cy * sz + sy * sx * cz, sy * cx], [cx * sz, cx * cz, -sx],
                  [-sy * cz + cy * sx * sz, sy * sz + cy * sx * cz, cy * cx]])
    if g.random() < 0.2:
        m = m * (1.0 + g.uniform(-2e-4, 2e-4, (3, 3)))      # slightly off orthonormal (still inside the reuse policy's epsilon)
    return m


def scene(g, p, M, pos, W, H, max_spheres=12000):
    kind = g.integers(0, 5)
    n = int(g.choice([1, 7, 60, 700, 3000, 12000]))
    n = min(n, max_spheres)
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


def _set_matrix(p, M):
    for i in range(3):
        for j in range(3):
            p.inv_v[4 * i + j] = float(M[i, j])


def _matrix_of(p):
    return np.array([[p.inv_v[4 * i + j] for j in range(3)] for i in range(3)], dtype=np.float64)


class Buffers:
    """Device frame buffers by size, reused from case to case."""

    def __init__(self, torch):
        self.torch = torch
        self.by_size = {}

    def pair(self, W, H):
        if (W, H) not in self.by_size:
            t = self.torch
            self.by_size[(W, H)] = (t.empty(20 * W * H, dtype=t.uint8, device="cuda"), t.empty(20 * W * H, dtype=t.uint8, device="cuda"))
        return self.by_size[(W, H)]


def cull_case(R, torch, bufs, seed, sizes=CULL_SIZES, physics=True, oracle_rows=0, max_spheres=12000, stats=None):
    """One seed of the culling fuzzer.  Returns a list of findings (strings; empty = every frame of the case agreed).
    oracle_rows > 0: that many rows of the last frame of the case are also compared with the CPU oracle (rows spread over
    the frame, the first and the last included), the scene stepped alike when physics runs."""
    g = np.random.default_rng(seed)
    W, H = sizes[int(g.integers(0, len(sizes)))]
    pos = [float(v) for v in g.uniform(-30, 30, 3)]
    p = R.camera_params(W, H, pos, (0.0, float(np.pi), 0.0))
    M = general_matrix(g)
    fov = float(g.choice([1.0, 1.0, 0.2, 0.5, 2.0]))
    p.element1 = float(p.element1) * fov
    p.element2 = float(p.element2) * fov
    _set_matrix(p, M)
    sph, pl = scene(g, p, _matrix_of(p), pos, W, H, max_spheres)
    got, want = bufs.pair(W, H)
    found = []
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
        S = 20 if mode >= R.RGB_ASCII else 12
        creep = g.random() < 0.5
        # physics: some of the creeping runs also step their spheres between frames (y moves by speed * mover * dt and is
        # clamped to +-10) on both contexts alike; the lists' position budget has to cover it -- also for |mover| > 1
        phys = creep and physics and g.random() < 0.6 and len(sph) <= 3000
        sc = None
        if oracle_rows:
            sc = O.Scene.from_arrays(sph, pl)
        if phys:
            movers = g.choice([-1, 1, 1, 3, -2], len(sph))
            speeds = g.uniform(0.5, 4.0, len(sph))
            for i in range(len(sph)):
                for c in (a, b):
                    c.set_sphere_motion(i, int(movers[i]), float(speeds[i]))
                if sc is not None:
                    sc.objects()[i].mover = int(movers[i])
                    sc.objects()[i].speed = float(speeds[i])
        for f in range((8 if phys else 4) if creep else 1):
            if phys:
                dt = float(g.choice([0.004, 0.016, 0.033]))
                for c in (a, b):
                    c.update_objects(dt)
                if sc is not None:
                    O.lib().orc_update_objects(sc.ptrs(), sc.count, dt)
            if f:
                # creep: a small turn about a random axis and a small step, so that lists built for an earlier frame are reused
                w = g.normal(0, 1, 3) * 2e-4
                dR = np.array([[1, -w[2], w[1]], [w[2], 1, -w[0]], [-w[1], w[0], 1]])
                M = M @ dR
                _set_matrix(p, M)
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
            if stats is not None:
                stats["frames"] = stats.get("frames", 0) + 1
                stats.setdefault("kernels", {})
                stats["kernels"][a.last_kernel] = stats["kernels"].get(a.last_kernel, 0) + 1
            if not torch.equal(got, want):
                diff = (got[:S * W * H].view(H, W, S) != want[:S * W * H].view(H, W, S)).any(dim=2)
                ys, xs = torch.nonzero(diff, as_tuple=True)
                found.append("seed %d frame %d: culling kernel differs from the brute kernel: %dx%d fov x%.1f mode %d, %d spheres %d planes, options %r, %s: "
                             "%d pixels, rows %d..%d columns %d..%d" % (seed, f, W, H, fov, mode, len(sph), len(pl), opts, a.last_kernel, int(diff.sum()),
                                                                      int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())))
        if oracle_rows and sc is not None:
            # sampled rows of the LAST frame (after the creep and the physics steps) against the oracle
            rows = sorted(set([0, H - 1] + [int(v) for v in g.integers(0, H, max(0, oracle_rows - 2))]))
            op = U.oracle_params(p)
            host = got[:S * W * H].view(H, W * S).cpu().numpy()
            for r in rows:
                ref = O.render_row(op, sc, mode, r)
                if stats is not None:
                    stats["oracle_rows"] = stats.get("oracle_rows", 0) + 1
                    stats["oracle_visible"] = stats.get("oracle_visible", 0) + int((ref.reshape(W, S)[:, 2] == ord('3')).sum())
                if not np.array_equal(host[r], ref):
                    found.append("seed %d: row %d of the culling kernel's last frame differs from the oracle: %dx%d mode %d, %d spheres %d planes, "
                                 "options %r, %s: %s" % (seed, r, W, H, mode, len(sph), len(pl), opts, a.last_kernel,
                                                         U.first_diff(host[r], ref, S, W)))
    finally:
        a.close()
        b.close()
    return found


def paths_case(R, torch, seed, sizes=PATH_SIZES, oracle_rows=0, max_spheres=12000, stats=None):
    """One seed of the paths fuzzer: the same frame as row slabs (random cuts, every slab its own launch, shuffled) and as
    compact pixel words expanded into records (random segments), against the frame rendered in one launch; all five
    character modes.  oracle_rows > 0: rows of that one-launch frame are also compared with the CPU oracle."""
    g = np.random.default_rng(seed)
    W, H = sizes[int(g.integers(0, len(sizes)))]
    pos = [float(v) for v in g.uniform(-30, 30, 3)]
    p = R.camera_params(W, H, pos, (0.0, float(np.pi), 0.0))
    M = general_matrix(g)
    _set_matrix(p, M)
    sph, pl = scene(g, p, _matrix_of(p), pos, W, H, max_spheres)
    mode = int(g.integers(0, 5))
    S = 20 if mode >= R.RGB_ASCII else 12
    zt = 0 if mode >= R.RGB_ASCII else R.RENDER_ZERO_TAIL
    want = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
    got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
    words = torch.empty(W * H, dtype=torch.int32, device="cuda")
    found = []
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
        if stats is not None:
            stats["comparisons"] = stats.get("comparisons", 0) + 2
        if not torch.equal(got[:S * W * H], want[:S * W * H]):
            found.append("seed %d: slabs %r differ from the frame, %dx%d mode %d, %d spheres %d planes (%s)" % (seed, cuts, W, H, mode, len(sph), len(pl), c.last_kernel))
        # (b) compact words, expanded
        got.fill_(0xEE)
        torch.cuda.synchronize()
        c.render_rows(p, mode, 0, H, d_out=words.data_ptr(), out_row_base=0, flags=R.RENDER_COMPACT)
        c.synchronize()
        nseg = int(g.integers(1, 5))
        bounds = sorted(set([0, W * H] + [int(v) for v in g.integers(1, W * H, nseg - 1)])) if W * H > 1 else [0, W * H]
        c.expand(mode, words.data_ptr(), got.data_ptr(), [(bounds[i], bounds[i], bounds[i + 1] - bounds[i]) for i in range(len(bounds) - 1)])
        c.synchronize()
        if not torch.equal(got[:S * W * H], want[:S * W * H]):
            found.append("seed %d: expanded compact words differ from the frame, %dx%d mode %d segments %r (%s)" % (seed, W, H, mode, bounds, c.last_kernel))
        # (c) Minimize from the words against Minimize of the records (the two GPU passes; small frames also against the oracle's)
        m1 = torch.empty(20 * W * H + 16, dtype=torch.uint8, device="cuda")
        m2 = torch.empty(20 * W * H + 16, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        n1 = c.minimize(mode, W, H, d_in=want.data_ptr(), d_out=m1.data_ptr())
        n2 = c.minimize_words(mode, W, H, words.data_ptr(), d_out=m2.data_ptr())
        if stats is not None:
            stats["comparisons"] = stats.get("comparisons", 0) + 1
        if n1 != n2 or not torch.equal(m1[:n1], m2[:n2]):
            found.append("seed %d: Minimize from words (%d bytes) differs from Minimize of the records (%d bytes), %dx%d mode %d" % (seed, n2, n1, W, H, mode))
        elif W * H <= 400 * 150:
            ref = O.minimize(mode, want.cpu().numpy(), W, H)
            if ref.size != n2 or not np.array_equal(ref, m2[:n2].cpu().numpy()):
                found.append("seed %d: Minimize from words differs from the oracle's Minimize, %dx%d mode %d" % (seed, W, H, mode))
        del m1, m2
        # (d) a slab of three frames with three cameras: one call on one stream (the batched kernel where the plan allows) against
        # a launch per frame
        if W * H <= 3840 * 2160:
            cams = []
            for i in range(3):
                q = R.Params.from_buffer_copy(p)
                q.cam_pos[0] = float(q.cam_pos[0]) + 0.37 * i
                q.inv_v[1] = float(q.inv_v[1]) + 1e-3 * i
                cams.append(q)
            r0 = int(g.integers(0, H))
            rows = int(g.integers(1, H - r0 + 1))
            nb = rows * W * S
            a3 = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(3)]
            b3 = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(3)]
            for t in a3 + b3:
                t.fill_(0xEE)
            torch.cuda.synchronize()
            c.submit_slabs(cams, mode, r0, rows, [t.data_ptr() for t in a3], r0, [None] * 3)
            for i in range(3):
                c.render_rows(cams[i], mode, r0, rows, d_out=b3[i].data_ptr(), out_row_base=r0)
            c.synchronize()
            if stats is not None:
                stats["comparisons"] = stats.get("comparisons", 0) + 1
                stats["batched"] = c.get_option(R.STAT_BATCHED_LAUNCHES) + stats.get("batched", 0)
            for i in range(3):
                if not torch.equal(a3[i], b3[i]):
                    found.append("seed %d: frame %d of a three-frame slab call (rows %d+%d) differs from its own launch, %dx%d mode %d (%s)" % (seed, i, r0, rows, W, H, mode, c.last_kernel))
            del a3, b3
        if oracle_rows:
            sc = O.Scene.from_arrays(sph, pl)
            op = U.oracle_params(p)
            host = want[:S * W * H].view(H, W * S).cpu().numpy()
            for r in sorted(set([0, H - 1] + [int(v) for v in g.integers(0, H, max(0, oracle_rows - 2))])):
                ref = O.render_row(op, sc, mode, r)
                if stats is not None:
                    stats["oracle_rows"] = stats.get("oracle_rows", 0) + 1
                if not np.array_equal(host[r], ref):
                    found.append("seed %d: row %d of the frame differs from the oracle, %dx%d mode %d, %d spheres %d planes (%s): %s"
                                 % (seed, r, W, H, mode, len(sph), len(pl), c.last_kernel, U.first_diff(host[r], ref, S, W)))
    finally:
        c.close()
    return found

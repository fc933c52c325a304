"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed goldens.

Bar: the ANSI character buffer is byte-exact (character indices, colour digits, layout).  The
hit distances / colours behind it are fp32 and required to agree within 1e-5 relative
(BASELINE.json north_star); since both sides evaluate the same IEEE operations in the same order
the bytes agree exactly, which implies the tolerance.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
import util as U

pytestmark = pytest.mark.gpu

TOL_REL = 1e-5  # north_star tolerance on distances/colours; byte equality below is the stricter check


@pytest.fixture(scope="module")
def R():
    return U.pkg()


@pytest.fixture(scope="module")
def ctx(R):
    c = R.Context(3840, 2160)
    yield c
    c.close()


def assert_same(got, want, mode, W, what):
    S = 20 if mode >= O.RGB_ASCII else 12
    assert np.array_equal(got, want), "%s: %s" % (what, U.first_diff(got, want, S, W))


KERNELS = ["brute", "binned"]


def set_kernel(R, ctx, kernel, tile=0, subtiles=0, two_level=-1, refine=-1, tile_order=None):
    if tile_order is not None:
        ctx.set_option(R.OPT_TILE_ORDER, tile_order)
    ctx.set_option(R.OPT_KERNEL, {"auto": R.KERNEL_AUTO, "brute": R.KERNEL_BRUTE, "binned": R.KERNEL_BINNED}[kernel])
    ctx.set_option(R.OPT_TILE_LOG2_W, tile)
    ctx.set_option(R.OPT_SUBTILES, subtiles)
    ctx.set_option(R.OPT_TWO_LEVEL, two_level)
    ctx.set_option(R.OPT_REFINE, refine)


# ---------------------------------------------------------------- reference default scene

SURVEY_8C_400x150 = {O.BIT_ASCII: "566f369b48c48349", O.BIT_PIXEL: "2600c441a058a41f", O.RGB_ASCII: "dd3497ccdb38ff6e",
                     O.RGB_PIXEL: "08bda1486917bf70"}
SURVEY_8C_1080P = {O.BIT_ASCII: "b366f64565c06fa1", O.BIT_PIXEL: "454b2ee3b30179c4", O.RGB_ASCII: "71e4385fd8fe0844",
                   O.RGB_PIXEL: "0cef41476e6725c5"}


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mode", range(5))
@pytest.mark.parametrize("res", [(400, 150), (1920, 1080)])
def test_default_scene_matches_oracle_and_survey_hash(R, ctx, res, mode, kernel):
    w, h = res
    set_kernel(R, ctx, kernel)
    ctx.set_reference_default_scene()
    p = R.camera_params(w, h)
    got = ctx.render_to_host(p, mode)
    want = O.render(U.oracle_params(p), O.Scene.reference_default(), mode, threads=8)
    assert_same(got, want, mode, w, "default scene %dx%d %s %s" % (w, h, O.MODE_NAMES[mode], kernel))
    known = (SURVEY_8C_400x150 if res == (400, 150) else SURVEY_8C_1080P).get(mode)
    if known:  # the reference's own known answers (RGB_NORMALS excluded: they used the x86 conversion)
        assert O.fnv1a64(got, O.FNV_OFFSET_SURVEY) == known


# ---------------------------------------------------------------- BASELINE configs

@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mode", range(5))
def test_c1_all_modes(R, ctx, mode, kernel):
    set_kernel(R, ctx, kernel)
    p, sph, pl = R.config_inputs("C1")
    ctx.set_scene(sph, pl)
    got = ctx.render_to_host(p, mode)
    want = O.render(U.oracle_params(p), O.Scene.from_arrays(sph, pl), mode)
    assert_same(got, want, mode, int(p.x), "C1 %s %s" % (O.MODE_NAMES[mode], kernel))
    assert O.fnv1a64(got) == U.load_golden()["C1_%s" % O.MODE_NAMES[mode]]["frame_fnv1a64"]


def test_c1_against_committed_frame(R, ctx):
    set_kernel(R, ctx, "auto")
    p, sph, pl = R.config_inputs("C1")
    ctx.set_scene(sph, pl)
    got = ctx.render_to_host(p, R.RGB_ASCII)
    want = np.load(os.path.join(U.GOLDEN_DIR, "c1_rgb_ascii_frame.npz"))["frame"]
    assert_same(got, want, R.RGB_ASCII, int(p.x), "C1 frame fixture")


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mode", [O.BIT_ASCII, O.RGB_ASCII])
def test_c2_full_size(R, ctx, mode, kernel):
    set_kernel(R, ctx, kernel)
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    got = ctx.render_to_host(p, mode)
    gold = U.load_golden()["C2_%s" % O.MODE_NAMES[mode]]
    if O.fnv1a64(got) != gold["frame_fnv1a64"]:
        want = O.render(U.oracle_params(p), O.Scene.from_arrays(sph, pl), mode, threads=16)
        assert_same(got, want, mode, int(p.x), "C2 %s %s" % (O.MODE_NAMES[mode], kernel))
        pytest.fail("frame equals the oracle but not the committed golden hash")
    W, H = int(p.x), int(p.y)
    rec = got[:(20 if mode >= 2 else 12) * W * H].reshape(H, W, -1)
    assert int((rec[:, :W - 1, 2] == ord("3")).sum()) == gold["foreground_pixels"]


@pytest.mark.parametrize("name", ["C3", "C5"])
def test_large_configs_against_golden_hash(R, ctx, name):
    gold = U.load_golden().get("%s_RGB_ASCII" % name)
    if gold is None:
        pytest.skip("golden for %s not generated (make_golden.py --big)" % name)
    set_kernel(R, ctx, "auto")
    p, sph, pl = R.config_inputs(name)
    ctx.set_scene(sph, pl)
    got = ctx.render_to_host(p, R.RGB_ASCII)
    assert O.fnv1a64(got) == gold["frame_fnv1a64"]


def test_c3_brute_equals_binned(R, ctx):
    p, sph, pl = R.config_inputs("C3")
    ctx.set_scene(sph, pl)
    set_kernel(R, ctx, "brute")
    a = ctx.render_to_host(p, R.RGB_ASCII)
    for two in (0, 1):
        set_kernel(R, ctx, "binned", two_level=two)
        b = ctx.render_to_host(p, R.RGB_ASCII)
        assert_same(b, a, R.RGB_ASCII, int(p.x), "C3 binned (two-level %d) vs brute" % two)


@pytest.mark.parametrize("two", [0, 1, 2])
@pytest.mark.parametrize("mode", [O.BIT_ASCII, O.RGB_ASCII, O.RGB_NORMALS])
def test_two_level_culling_on_c2(R, ctx, mode, two):
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    set_kernel(R, ctx, "binned", two_level=two)
    got = ctx.render_to_host(p, mode)
    if mode == O.RGB_NORMALS:
        set_kernel(R, ctx, "brute")
        assert_same(got, ctx.render_to_host(p, mode), mode, int(p.x), "C2 normals two-level %d vs brute" % two)
    else:
        assert O.fnv1a64(got) == U.load_golden()["C2_%s" % O.MODE_NAMES[mode]]["frame_fnv1a64"]


def test_two_level_row_slabs(R, ctx):
    """Two-level culling inside row-slab launches (cells are laid out from the slab's first row)."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    W, H = int(p.x), int(p.y)
    set_kernel(R, ctx, "binned", two_level=1)
    dst = torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for (r0, r1) in ((0, 135), (135, 541), (541, 1080)):
        ctx.render_rows(p, R.RGB_ASCII, r0, r1 - r0, d_out=dst.data_ptr(), out_row_base=0)
    ctx.synchronize()
    assert O.fnv1a64(dst.cpu().numpy()) == U.load_golden()["C2_RGB_ASCII"]["frame_fnv1a64"]


# ---------------------------------------------------------------- properties and edge cases

@pytest.mark.parametrize("subtiles", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("tile", [2, 3, 4, 5, 6])
def test_every_tile_shape_gives_the_same_frame(R, ctx, tile, subtiles):
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    set_kernel(R, ctx, "binned", tile, subtiles)
    got = ctx.render_to_host(p, R.RGB_ASCII)
    assert O.fnv1a64(got) == U.load_golden()["C2_RGB_ASCII"]["frame_fnv1a64"]


def test_row_slabs_assemble_to_the_full_frame(R, ctx):
    """rendering row slabs (global row index in ray generation) into one
    buffer equals the one-launch frame, for 1/2/4/8-way splits and a ragged one."""
    import torch
    set_kernel(R, ctx, "auto")
    p, sph, pl = R.config_inputs("C1")
    ctx.set_scene(sph, pl)
    W, H = int(p.x), int(p.y)
    full = ctx.render_to_host(p, R.RGB_ASCII)
    for parts in (1, 2, 4, 8, 7):
        # full-frame destination
        dst = torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the context renders on its own non-blocking stream
        bounds = [H * i // parts for i in range(parts + 1)]
        for i in range(parts):
            ctx.render_rows(p, R.RGB_ASCII, bounds[i], bounds[i + 1] - bounds[i], d_out=dst.data_ptr(), out_row_base=0)
        ctx.synchronize()
        assert np.array_equal(dst.cpu().numpy(), full), "parts=%d" % parts
        # slab-local destinations, concatenated (what each rank holds before the gather)
        pieces = []
        for i in range(parts):
            rows = bounds[i + 1] - bounds[i]
            slab = torch.zeros(20 * W * rows, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # the context renders on its own non-blocking stream
            ctx.render_rows(p, R.RGB_ASCII, bounds[i], rows, d_out=slab.data_ptr(), out_row_base=bounds[i])
            ctx.synchronize()
            pieces.append(slab.cpu().numpy())
        assert np.array_equal(np.concatenate(pieces), full), "slabs parts=%d" % parts


def test_c2_row_slabs_with_the_librarys_own_choices(R, ctx):
    """Config 2 as the row slabs of 2 / 3 / 4 / 8 ranks with every option on auto: the sub-tile count follows the
    slab's size (3 / 2 / 2 / 1), slabs from half a megapixel up are balanced (orders switched while the frames
    repeat).  Each slab is rendered 1, 5 and 9 times in a row; the assembled frame is the golden one every time."""
    import torch
    set_kernel(R, ctx, "auto", tile_order=-1)
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    W, H = int(p.x), int(p.y)
    gold = U.load_golden()["C2_RGB_ASCII"]["frame_fnv1a64"]
    dst = torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda")
    first = None
    for parts in (2, 3, 4, 8):
        bounds = [H * i // parts for i in range(parts + 1)]
        for rep in range(3):
            dst.fill_(0xEE)
            torch.cuda.synchronize()
            for i in range(parts):
                # the same slab several times in a row, as a rank renders it frame after frame (a stream keeps one
                # dispatch order, for the grid it saw last)
                for _ in range(1 + 4 * rep):
                    ctx.render_rows(p, R.RGB_ASCII, bounds[i], bounds[i + 1] - bounds[i], d_out=dst.data_ptr(), out_row_base=0)
            ctx.synchronize()
            if first is None:
                assert O.fnv1a64(dst.cpu().numpy()) == gold
                first = dst.clone()
            else:
                assert torch.equal(dst, first), "parts=%d, repetition %d" % (parts, rep)


def test_mode_switch_leaves_reference_zero_semantics(R, ctx):
    """RGB frame, then an 8-bit frame in the same buffer: bytes past 12*W*H must read as zero, as after
    the reference's per-frame memset (RayTracingManager.cu:86)."""
    set_kernel(R, ctx, "auto")
    ctx.set_reference_default_scene()
    p = R.camera_params(400, 150)
    ctx.render_to_host(p, R.RGB_PIXEL)
    got = ctx.render_to_host(p, R.BIT_PIXEL)
    want = O.render(U.oracle_params(p), O.Scene.reference_default(), O.BIT_PIXEL)
    assert_same(got, want, O.BIT_PIXEL, 400, "8-bit after RGB")
    # smaller frame after a larger one
    p2 = R.camera_params(200, 75)
    got2 = ctx.render_to_host(p2, R.BIT_ASCII)
    want2 = O.render(U.oracle_params(p2), O.Scene.reference_default(), O.BIT_ASCII)
    assert_same(got2, want2, O.BIT_ASCII, 200, "smaller 8-bit frame")


def test_sdl_mode_writes_nothing(R, ctx):
    import torch
    ctx.set_reference_default_scene()
    p = R.camera_params(64, 32)
    dst = torch.full((20 * 64 * 32,), 7, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the context renders on its own non-blocking stream
    ctx.render_rows(p, R.SDL, 0, 32, d_out=dst.data_ptr())
    ctx.synchronize()
    assert bool((dst == 7).all())


def test_invalid_mode_and_arguments_are_reported(R, ctx):
    p = R.camera_params(64, 32)
    with pytest.raises(R.RtxError) as e:
        ctx.render(p, 6)
    assert e.value.status == R.ERR_INVALID_MODE
    with pytest.raises(R.RtxError) as e:
        ctx.render(p, -1)
    assert e.value.status == R.ERR_INVALID_MODE
    big = R.camera_params(4000, 4000)
    with pytest.raises(R.RtxError) as e:
        ctx.render(big, R.RGB_ASCII)
    assert e.value.status == R.ERR_TOO_LARGE


def test_empty_scene_and_tiny_frames(R, ctx):
    set_kernel(R, ctx, "auto")
    ctx.scene_clear()
    for (w, h) in ((1, 1), (1, 5), (2, 1), (3, 3), (65, 5), (17, 33)):
        p = R.camera_params(w, h)
        for mode in (R.BIT_ASCII, R.RGB_ASCII):
            got = ctx.render_to_host(p, mode)
            want = O.render(U.oracle_params(p), O.Scene(), mode)
            assert_same(got, want, mode, w, "empty scene %dx%d" % (w, h))
    ctx.set_reference_default_scene()
    for (w, h) in ((1, 1), (2, 2), (63, 3), (64, 4), (65, 5), (129, 31)):
        p = R.camera_params(w, h)
        for kernel in KERNELS:
            set_kernel(R, ctx, kernel)
            got = ctx.render_to_host(p, R.RGB_ASCII)
            want = O.render(U.oracle_params(p), O.Scene.reference_default(), O.RGB_ASCII)
            assert_same(got, want, O.RGB_ASCII, w, "default scene %dx%d %s" % (w, h, kernel))


def test_interleaved_creation_order_breaks_ties_like_the_reference(R, ctx):
    """Coincident objects: the earliest created one wins (strict '<' in creation order,
    RayTracing.cu:123), whether it is a sphere or a plane."""
    p = R.camera_params(160, 60)
    for order in ("ssp", "pss", "sps"):
        ctx.scene_clear()
        sc = O.Scene()
        si = 0
        for ch in order:
            if ch == "s":
                col = [(250.0, 10.0, 10.0), (10.0, 250.0, 10.0)][si]
                si += 1
                ctx.add_sphere(6.0, (0.0, 0.0, 30.0), col)
                sc.add_sphere(6.0, (0.0, 0.0, 30.0), col)
            else:
                ctx.add_plane((0.0, -2.0, 30.0), (0.0, 1.0, 0.0), (90.0, 90.0, 200.0), 30.0, 30.0)
                sc.add_plane((0.0, -2.0, 30.0), (0.0, 1.0, 0.0), (90.0, 90.0, 200.0), 30.0, 30.0)
        for kernel in KERNELS:
            set_kernel(R, ctx, kernel)
            got = ctx.render_to_host(p, R.RGB_PIXEL)
            want = O.render(U.oracle_params(p), sc, O.RGB_PIXEL)
            assert_same(got, want, O.RGB_PIXEL, 160, "order %s %s" % (order, kernel))


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_and_cameras_binned_vs_brute_vs_oracle(R, ctx, seed):
    """Randomised stress of the culling margin: tiny and huge spheres, spheres around and behind the
    camera, a camera inside spheres, far-away origins, arbitrary poses."""
    rng = np.random.default_rng(1000 + seed)
    w, h = [(257, 97), (400, 150), (320, 180), (191, 203), (640, 64), (96, 301)][seed]
    pos = rng.uniform(-30, 30, 3) if seed % 2 else rng.uniform(-3000, 3000, 3)
    rot = (rng.uniform(-1.2, 1.2), rng.uniform(0, 6.28), 0.0)
    p = R.camera_params(w, h, [float(v) for v in pos], [float(v) for v in rot])
    n = 700
    centres = pos + rng.normal(0, 1, (n, 3)) * rng.choice([5.0, 40.0, 300.0], (n, 1))
    radii = np.abs(rng.normal(0, 1, n)) * rng.choice([0.01, 0.5, 5.0, 60.0], n) + 1e-4
    cols = np.floor(rng.uniform(1, 256, (n, 3)))
    sph = np.concatenate([centres, radii[:, None], cols], axis=1).astype(np.float32)
    pl = np.array([[pos[0], pos[1] - 8, pos[2], 0, 1, 0, 100, 120, 140, 400, 400],
                   [pos[0], pos[1] + 50, pos[2], 0.2, -1, 0.1, 30, 200, 90, 300, 500]], dtype=np.float32)
    ctx.set_scene(sph, pl)
    sc = O.Scene.from_arrays(sph, pl)
    want = O.render(U.oracle_params(p), sc, O.RGB_ASCII, threads=8)
    for kernel in KERNELS:
        for tile in ((0, 6) if kernel == "brute" else (0, 2, 4, 6)):
            for sub in ((0,) if kernel == "brute" else (1, 2, 4, 8)):
                for two in ((0,) if kernel == "brute" else (0, 1, 2)):
                    for refine in ((0, 1) if kernel == "binned" and sub <= 4 else (0,)):
                        set_kernel(R, ctx, kernel, tile, sub, two, refine)
                        got = ctx.render_to_host(p, R.RGB_ASCII)
                        assert_same(got, want, O.RGB_ASCII, w, "random scene %d %s tile %d sub %d two-level %d refine %d"
                                    % (seed, kernel, tile, sub, two, refine))
                        # (REFINE is declined for macro tiles beyond 64 x 64 pixels -- its tables are that large)
                        if refine and tile in (0, 4) and sub <= 2:
                            assert ctx.last_kernel.endswith(",refine>")
                        if not refine:
                            assert not ctx.last_kernel.endswith(",refine>")


def test_per_pixel_values_within_tolerance(R, ctx):
    """Decodes colours back from the RGB records and compares them with the oracle's float colours:
    the digits are the truncated floats, so |decoded - colour| < 1 and exact after truncation."""
    set_kernel(R, ctx, "auto")
    p, sph, pl = R.config_inputs("C1")
    ctx.set_scene(sph, pl)
    W, H = int(p.x), int(p.y)
    got = ctx.render_to_host(p, R.RGB_PIXEL).reshape(H, W, 20)
    _, px = O.render(U.oracle_params(p), O.Scene.from_arrays(sph, pl), O.RGB_PIXEL, want_pixels=True)
    vis = px["distance"] <= p.cam_far
    vis[:, -1] = False

    def dec(d):
        d = d.astype(np.int32)
        return np.where(d[..., 0] > 0, d[..., 0] - 48, 0) * 100 + np.where(d[..., 1] > 0, d[..., 1] - 48, 0) * 10 + d[..., 2] - 48

    for ch, off in ((0, 7), (1, 11), (2, 15)):
        val = dec(got[..., off:off + 3])[vis]
        want = np.trunc(px["color"][..., ch][vis]).astype(np.int32)
        assert np.array_equal(val, want)
        assert np.all(np.abs(val - px["color"][..., ch][vis]) <= 1.0 + TOL_REL * 255)


def test_candidate_list_overflow_path(R, ctx):
    """More candidates per macro tile than the LDS list holds (hundreds of spheres stacked on the same
    pixels): the kernel falls back to one scene walk per sub-tile and must still pick, per pixel, the
    nearest sphere with the lowest creation index."""
    rng = np.random.default_rng(5)
    n = 1500
    # all spheres straddle the view axis at increasing depth, so every tile keeps most of them
    z = np.sort(rng.uniform(20, 200, n))[::-1]  # far ones created first
    centres = np.stack([rng.normal(0, 1.0, n), rng.normal(0, 0.3, n), z], axis=1)
    radii = rng.uniform(2.0, 6.0, n)
    cols = np.floor(rng.uniform(1, 256, (n, 3)))
    sph = np.concatenate([centres, radii[:, None], cols], axis=1).astype(np.float32)
    sph[100] = sph[99]  # an exact duplicate: the tie goes to the earlier one
    pl = np.zeros((0, 11), dtype=np.float32)
    ctx.set_scene(sph, pl)
    sc = O.Scene.from_arrays(sph, pl)
    p = R.camera_params(256, 96)
    want = O.render(U.oracle_params(p), sc, O.RGB_ASCII, threads=8)
    for kernel, tile, sub, two in (("brute", 0, 0, 0), ("binned", 0, 1, 0), ("binned", 0, 4, 0), ("binned", 3, 8, 1), ("binned", 5, 16, 0),
                                   ("binned", 0, 4, 1)):
        set_kernel(R, ctx, kernel, tile, sub, two)
        got = ctx.render_to_host(p, R.RGB_ASCII)
        assert_same(got, want, O.RGB_ASCII, 256, "overflow %s tile %d sub %d two-level %d" % (kernel, tile, sub, two))


def test_many_planes_beyond_the_lds_table(R, ctx):
    """More planes than the kernel hoists into LDS (16): the rest take the direct Plane::Trace form."""
    rng = np.random.default_rng(11)
    ctx.scene_clear()
    sc = O.Scene()
    for i in range(40):
        pos = (float(rng.uniform(-40, 40)), float(rng.uniform(-25, -2)), float(rng.uniform(20, 150)))
        nrm = (float(rng.normal(0, 0.2)), 1.0, float(rng.normal(0, 0.2)))
        col = [float(v) for v in np.floor(rng.uniform(1, 256, 3))]
        w, h = float(rng.uniform(5, 60)), float(rng.uniform(5, 60))
        ctx.add_plane(pos, nrm, col, w, h)
        sc.add_plane(pos, nrm, col, w, h)
        if i % 5 == 0:
            ctx.add_sphere(3.0, (pos[0], pos[1] + 6, pos[2]), col)
            sc.add_sphere(3.0, (pos[0], pos[1] + 6, pos[2]), col)
    p = R.camera_params(320, 120)
    want = O.render(U.oracle_params(p), sc, O.RGB_ASCII, threads=4)
    for kernel in KERNELS:
        set_kernel(R, ctx, kernel)
        got = ctx.render_to_host(p, R.RGB_ASCII)
        assert_same(got, want, O.RGB_ASCII, 320, "40 planes %s" % kernel)


@pytest.mark.parametrize("res", [(333, 77), (1025, 3), (7, 513), (2049, 5), (64, 64), (129, 257), (5, 513), (9, 513), (7, 257), (16, 513)])
def test_odd_frame_sizes_with_culling(R, ctx, res):
    """Frame sizes that are not multiples of any tile dimension, with enough spheres for the binned kernel
    and (forced) two-level culling; full frames and ragged row slabs."""
    import torch
    w, h = res
    p = R.camera_params(w, h, (1.0, 2.0, -3.0), (0.1, 3.0, 0.0))
    sph, pl = R.synth_scene(77, 300, 2, p.element1, p.element2)
    ctx.set_scene(sph, pl)
    sc = O.Scene.from_arrays(sph, pl)
    want = O.render(U.oracle_params(p), sc, O.RGB_ASCII, threads=4)
    for two in (0, 1):
        for sub in (0, 1, 3, 4):
            set_kernel(R, ctx, "binned", 0, sub, two)
            got = ctx.render_to_host(p, R.RGB_ASCII)
            assert_same(got, want, O.RGB_ASCII, w, "%dx%d two-level %d sub %d" % (w, h, two, sub))
    set_kernel(R, ctx, "binned", two_level=1)
    dst = torch.zeros(20 * w * h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    cuts = sorted(set([0, h // 3, h // 2 + 1 if h > 2 else h, h]))
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        ctx.render_rows(p, R.RGB_ASCII, r0, r1 - r0, d_out=dst.data_ptr(), out_row_base=0)
    ctx.synchronize()
    assert_same(dst.cpu().numpy(), want, O.RGB_ASCII, w, "%dx%d slabs" % (w, h))


@pytest.mark.parametrize("config,kernel", [("C1", "auto"), ("C1", "brute"), ("C2", "auto")])
def test_hit_distances_colours_normals_against_the_oracle(R, ctx, config, kernel):
    """RTX_RENDER_VALUES: the floats behind the records (distance, shadingValue, normal, colour) next to the
    oracle's per-pixel values.  north_star's tolerance is 1e-5 (relative) on hit distances and colours; the two
    sides evaluate the same IEEE operations in the same order, so the stricter check -- bit equality -- holds too."""
    import torch
    set_kernel(R, ctx, kernel)
    p, sph, pl = R.config_inputs(config)
    ctx.set_scene(sph, pl)
    W, H = int(p.x), int(p.y)
    rows = H if config == "C1" else 96          # the oracle's per-pixel path is single-threaded
    row0 = 0 if config == "C1" else 500
    vals = torch.zeros(W * rows * 8, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.render_rows(p, R.RGB_ASCII, row0, rows, d_out=vals.data_ptr(), out_row_base=row0, flags=R.RENDER_VALUES)
    ctx.synchronize()
    got = vals.cpu().numpy().reshape(rows, W, 8)
    _, px = O.render(U.oracle_params(p), O.Scene.from_arrays(sph, pl), O.RGB_ASCII, want_pixels=True, row0=row0, rows=rows)
    px = px[row0:row0 + rows]
    traced = np.ones((rows, W), dtype=bool)
    traced[:, -1] = False                      # column W-1 is never traced (RayTracing.cu:187)
    hit = (px["hit"] != 0) & traced
    assert hit.sum() > 0.3 * traced.sum()
    dist_g, dist_o = got[..., 0], px["distance"]
    # the stated tolerance
    assert np.all(np.abs(dist_g[hit] - dist_o[hit]) <= TOL_REL * np.abs(dist_o[hit]))
    assert np.all(np.abs(got[..., 5:8][hit] - px["color"][hit]) <= TOL_REL * 255.0)
    assert np.all(np.abs(got[..., 2:5][hit] - px["normal"][hit]) <= TOL_REL)
    # and the stricter one
    assert np.array_equal(dist_g[traced].view(np.uint32), dist_o[traced].view(np.uint32))     # misses: 99999999.f on both sides
    assert np.array_equal(got[..., 1][hit].view(np.uint32), px["shading_value"][hit].view(np.uint32))
    assert np.array_equal(got[..., 2:5][hit].view(np.uint32), px["normal"][hit].view(np.uint32))
    assert np.array_equal(got[..., 5:8][hit].view(np.uint32), px["color"][hit].view(np.uint32))
    assert not got[:, -1, :].any()


@pytest.mark.parametrize("mode", [O.BIT_ASCII, O.RGB_ASCII, O.RGB_NORMALS])
@pytest.mark.parametrize("sub,tile", [(1, 4), (2, 4), (2, 3), (2, 6), (1, 2), (4, 4), (4, 3), (3, 4)])
def test_per_wave_refinement_gives_the_same_frame(R, ctx, mode, sub, tile):
    """RTX_OPT_REFINE: each wave narrows the workgroup's candidate list to its own 64 pixels before scanning it.
    C2 (long lists when forced to few, large tiles) with and without it, against the golden hash / the oracle."""
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    frames = []
    for refine in (0, 1):
        set_kernel(R, ctx, "binned", tile, sub, 0, refine)
        frames.append(ctx.render_to_host(p, mode))
        if tile == 6:   # 64-pixel-wide sub-tiles: two of them side by side are beyond the REFINE kernels' 64 x 64 tables
            assert not ctx.last_kernel.endswith(",refine>") or bool(refine)
        else:
            assert ctx.last_kernel.endswith(",refine>") == bool(refine)
    set_kernel(R, ctx, "auto")
    assert np.array_equal(frames[0], frames[1])
    g = U.load_golden().get("C2_%s" % O.MODE_NAMES[mode])
    if g:
        assert O.fnv1a64(frames[1]) == g["frame_fnv1a64"]


def test_refinement_with_more_survivors_than_a_wave_keeps(R, ctx):
    """Hundreds of spheres stacked on the same pixels: a wave's refined list overflows its 192 slots and the
    wave must fall back to the whole list."""
    rng = np.random.default_rng(11)
    n = 600
    z = np.sort(rng.uniform(20, 200, n))[::-1]
    centres = np.stack([rng.normal(0, 0.5, n), rng.normal(0, 0.2, n), z], axis=1)
    radii = rng.uniform(2.0, 5.0, n)
    cols = np.floor(rng.uniform(1, 256, (n, 3)))
    sph = np.concatenate([centres, radii[:, None], cols], axis=1).astype(np.float32)
    pl = np.zeros((0, 11), dtype=np.float32)
    ctx.set_scene(sph, pl)
    p = R.camera_params(192, 64)
    want = O.render(U.oracle_params(p), O.Scene.from_arrays(sph, pl), O.RGB_ASCII, threads=8)
    for sub in (1, 2, 4):
        set_kernel(R, ctx, "binned", 4, sub, 0, 1)
        got = ctx.render_to_host(p, R.RGB_ASCII)
        assert_same(got, want, O.RGB_ASCII, 192, "refine overflow sub %d" % sub)
    set_kernel(R, ctx, "auto")


# ---------------------------------------------------------------- heaviest-first dispatch order (RTX_OPT_TILE_ORDER)

def _render_into_poisoned_buffer(R, ctx, p, mode, buf):
    """Whole frame into a caller buffer that starts as 0xEE everywhere: a macro tile that no workgroup renders (an
    order that is not a permutation) leaves its bytes poisoned."""
    import torch
    buf.fill_(0xEE)
    torch.cuda.synchronize()
    ctx.render_rows(p, mode, 0, int(p.y), d_out=buf.data_ptr(), out_row_base=0)
    ctx.synchronize()
    return buf.cpu().numpy()


@pytest.mark.parametrize("period", [-1, 1, 3])
def test_tile_order_keeps_every_frame_identical(R, ctx, period):
    """The order the workgroups take the macro tiles in is derived from the previous frames' work estimates; any
    permutation must give the same bytes.  C2 over many frames (order refreshed with period 1 / 3 / the library's
    default: five sub-tiles per workgroup so the grid is one dispatch round, balanced by rtx_balance_tiles on a stream
    of its own), the first frames against the golden hash, the rest against the first on the device."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    W, H = int(p.x), int(p.y)
    buf = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
    gold = U.load_golden()["C2_RGB_ASCII"]["frame_fnv1a64"]
    assert ctx.get_option(R.OPT_TILE_ORDER) == -1   # the default: auto
    set_kernel(R, ctx, "auto" if period < 0 else "binned", tile_order=period)
    first = None
    for i in range(150 if period < 0 else 40):
        # (auto: the balancing passes run on their own stream, every fourth frame at first and every 64th from frame 64
        # on; each switches the launches to the other half of the order buffer three frames later)
        if i < 7:
            got = _render_into_poisoned_buffer(R, ctx, p, R.RGB_ASCII, buf)
            assert O.fnv1a64(got) == gold, "frame %d" % i
            if first is None:
                first = torch.from_numpy(got).cuda()
        else:
            buf.fill_(0xEE)
            torch.cuda.synchronize()
            ctx.render_rows(p, R.RGB_ASCII, 0, H, d_out=buf.data_ptr(), out_row_base=0)
            ctx.synchronize()
            assert torch.equal(buf, first), "frame %d" % i
    # and queued back to back, as a renderer submits them (passes, waits and switches with launches in flight)
    bufs = [torch.full_like(buf, 0xEE) for _ in range(3)]
    torch.cuda.synchronize()
    for i in range(100):
        ctx.render_rows(p, R.RGB_ASCII, 0, H, d_out=bufs[i % 3].data_ptr(), out_row_base=0)
    ctx.synchronize()
    for b in bufs:
        assert torch.equal(b, first)
    set_kernel(R, ctx, "binned", tile_order=0)
    assert O.fnv1a64(_render_into_poisoned_buffer(R, ctx, p, R.RGB_ASCII, buf)) == gold
    set_kernel(R, ctx, "auto", tile_order=-1)


@pytest.mark.parametrize("mode", [O.RGB_ASCII, O.BIT_ASCII, O.RGB_NORMALS])
def test_every_frame_of_four_streams_in_flight_is_the_frame(R, mode):
    """Four render streams, frames queued round-robin without waiting (the bench's default form), from a fresh context so
    that the balancing passes and order switches of the first 64 launches per stream fall into the checked frames: every
    one of 512 frames lands in its own buffer of a ring and is compared with the golden frame on the device."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    c = R.Context(W, H)
    try:
        c.set_scene(sph, pl)
        S = 20 if mode >= O.RGB_ASCII else 12
        c.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)      # the frame to compare with: brute kernel, frame order
        c.render(p, mode)
        want = torch.from_numpy(c.read_frame(20 * W * H)[:S * W * H]).cuda()
        g = U.load_golden().get("C2_%s" % O.MODE_NAMES[mode])
        if g:
            full = np.zeros(20 * W * H, dtype=np.uint8)
            full[:S * W * H] = want.cpu().numpy()
            assert O.fnv1a64(full) == g["frame_fnv1a64"]
        c.set_option(R.OPT_KERNEL, R.KERNEL_AUTO)
        streams = [torch.cuda.Stream() for _ in range(4)]
        ring = [torch.empty(S * W * H, dtype=torch.uint8, device="cuda") for _ in range(32)]
        for rnd in range(16 if mode == O.RGB_ASCII else 6):
            for b in ring:
                b.fill_(0xEE)
            torch.cuda.synchronize()
            for i, b in enumerate(ring):
                c.render_rows(p, mode, 0, H, d_out=b.data_ptr(), out_row_base=0, stream=streams[i % 4].cuda_stream)
            torch.cuda.synchronize()
            for i, b in enumerate(ring):
                assert torch.equal(b, want), "round %d, frame %d" % (rnd, i)
    finally:
        c.close()


def test_auto_order_follows_a_camera_that_creeps_stops_and_jumps(R):
    """The library's own policy (orders used near the view they were measured on, refreshed when stale, none while the
    view moves fast; spheres moved by rtx_update_objects age them too): 120 frames of a camera that creeps, rests, jumps
    and races, with the spheres bouncing now and then -- every frame equals the brute kernel's frame of the same state."""
    import torch
    rng = np.random.default_rng(5)
    p0, sph, pl = R.config_inputs("C2")
    W, H = int(p0.x), int(p0.y)
    a = R.Context(W, H)
    b = R.Context(W, H)
    try:
        for c in (a, b):
            c.set_scene(sph, pl)
        b.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
        yaw, pos = np.pi, np.zeros(3)
        buf = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        ref = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        for i in range(120):
            phase = i // 20
            if phase in (0, 3):
                yaw += 1.0e-4                      # creeping: an order stays usable for tens of frames
            elif phase == 2 and i % 20 == 0:
                yaw += 0.3                         # a jump: every order is stale at once
                pos = rng.uniform(-2, 2, 3)
            elif phase == 4:
                yaw += 0.01                        # racing: no order is worth deriving
            if phase == 5 and i % 3 == 0:
                for c in (a, b):
                    c.update_objects(0.05)         # the scene moves under a resting camera
            p = R.camera_params(W, H, [float(v) for v in pos], (0.0, float(yaw), 0.0))
            torch.cuda.synchronize()
            a.render_rows(p, R.RGB_ASCII, 0, H, d_out=buf.data_ptr(), out_row_base=0)
            b.render_rows(p, R.RGB_ASCII, 0, H, d_out=ref.data_ptr(), out_row_base=0)
            a.synchronize()
            b.synchronize()
            assert torch.equal(buf, ref), "frame %d" % i
    finally:
        a.close()
        b.close()


def test_tile_order_option_values(R, ctx):
    for ok in (-1, 0, 1, 64, 1 << 20):
        ctx.set_option(R.OPT_TILE_ORDER, ok)
        assert ctx.get_option(R.OPT_TILE_ORDER) == ok
    for bad in (-2, -100, (1 << 20) + 1):
        with pytest.raises(R.RtxError) as e:
            ctx.set_option(R.OPT_TILE_ORDER, bad)
        assert e.value.status == R.ERR_INVALID_ARGUMENT
    ctx.set_option(R.OPT_TILE_ORDER, -1)


def test_contexts_end_with_a_balancing_pass_in_flight(R):
    """rtx_destroy right after launches whose balancing pass (own stream) is still queued, and option changes between
    such launches: nothing may be freed under a running pass, and every frame stays the frame."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    gold = U.load_golden()["C2_RGB_ASCII"]["frame_fnv1a64"]
    for n_frames in (1, 2, 3, 5):
        c = R.Context(1920, 1080)
        c.set_scene(sph, pl)
        for i in range(n_frames):
            c.render(p, R.RGB_ASCII)
            if i == 1:
                c.set_option(R.OPT_TILE_ORDER, 0)
                c.render(p, R.RGB_ASCII)
                c.set_option(R.OPT_TILE_ORDER, -1)
        if n_frames == 5:
            assert O.fnv1a64(c.read_frame(20 * 1920 * 1080)) == gold
        c.close()   # no synchronize before it
    torch.cuda.synchronize()


def test_tile_order_with_a_moving_camera_changing_grids_and_slabs(R, ctx):
    """Stale estimates (the camera moves every frame), a different tile grid every few frames (frame size, sub-tile
    count, two-level culling on and off) and row slabs: always the frame the brute kernel renders in frame order."""
    import torch
    rng = np.random.default_rng(77)
    sph, pl = R.synth_scene(9, 900, 2, 6.0, 0.577)
    ctx.set_scene(sph, pl)
    buf = torch.empty(20 * 1920 * 1080, dtype=torch.uint8, device="cuda")
    for step in range(14):
        w, h = [(1920, 1080), (1283, 721), (1920, 1080), (960, 1080)][(step // 3) % 4]
        pos = [float(v) for v in rng.uniform(-3, 3, 3)]
        rot = (float(rng.uniform(-0.2, 0.2)), float(np.pi + rng.uniform(-0.3, 0.3)), 0.0)
        p = R.camera_params(w, h, pos, rot)
        set_kernel(R, ctx, "brute", tile_order=0)
        want = ctx.render_to_host(p, R.RGB_ASCII)
        set_kernel(R, ctx, "binned", subtiles=(0, 2, 8)[step % 3], two_level=(0, 1)[(step // 2) % 2], tile_order=1)
        got = _render_into_poisoned_buffer(R, ctx, p, R.RGB_ASCII, buf)[:20 * w * h]
        assert_same(got, want, R.RGB_ASCII, w, "tile order, moving camera, step %d" % step)
        # the same frame again as three ragged slabs (each slab shape has its own tile grid and order)
        buf.fill_(0xEE)
        torch.cuda.synchronize()
        for r0, r1 in ((0, h // 3), (h // 3, h // 2 + 5), (h // 2 + 5, h)):
            ctx.render_rows(p, R.RGB_ASCII, r0, r1 - r0, d_out=buf.data_ptr(), out_row_base=0)
        ctx.synchronize()
        assert_same(buf.cpu().numpy()[:20 * w * h], want, R.RGB_ASCII, w, "tile order, slabs, step %d" % step)
    set_kernel(R, ctx, "auto", tile_order=-1)


# ---------------------------------------------------------------- compact coarse-cell lists (rtx_bin_cells)

def test_cell_lists_that_do_not_fit_fall_back_to_the_whole_scene(R, ctx):
    """RTX_OPT_CELL_CAPACITY far below what the cells need: every overflowing cell's workgroups stage the whole scene
    instead of the (truncated) list -- slower, the same frame."""
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    gold = U.load_golden()["C2_RGB_ASCII"]["frame_fnv1a64"]
    try:
        for cap in (1, 7, 64, 0):
            ctx.set_option(R.OPT_CELL_CAPACITY, cap)
            for sub in (0, 2):
                set_kernel(R, ctx, "binned", subtiles=sub, two_level=1)
                assert O.fnv1a64(ctx.render_to_host(p, R.RGB_ASCII)) == gold, "capacity %d, sub-tiles %d" % (cap, sub)
                assert O.fnv1a64(ctx.render_to_host(p, R.RGB_ASCII)) == gold, "capacity %d, second frame (counter buffers alternate)" % cap
    finally:
        ctx.set_option(R.OPT_CELL_CAPACITY, 0)
        set_kernel(R, ctx, "auto")


def test_a_million_spheres_and_scratch_proportional_to_the_scene(R):
    """1 048 576 spheres on a small frame: the two-level kernel (compact lists: 4 ns / cells + 1024 entries per cell,
    i.e. O(spheres) words of scratch, where round 1 used cells x ns) against the brute kernel, byte for byte; and
    config 5's scratch stays below 8 MB."""
    import torch
    w, h = 320, 128
    p = R.camera_params(w, h)
    n = 1 << 20
    sph, pl = R.synth_scene(11, n, 0, p.element1, p.element2)
    with R.Context(1920, 1080) as c:
        c.set_scene(sph, pl)
        c.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
        want = c.render_to_host(p, R.RGB_ASCII)
        assert int((want.reshape(h, w, 20)[:, :w - 1, 2] == ord("3")).sum()) > 0.2 * w * h   # the scene does cover the frame
        c.set_option(R.OPT_KERNEL, R.KERNEL_BINNED)
        for two in (-1, 1):
            c.set_option(R.OPT_TWO_LEVEL, two)
            got = c.render_to_host(p, R.RGB_ASCII)
            assert_same(got, want, R.RGB_ASCII, w, "1M spheres, two-level %d" % two)
        # config 5 (65 536 spheres at 1080p): device memory taken by the first two-level frames
        p5, sph5, pl5 = R.config_inputs("C5")
        c.set_scene(sph5, pl5)
        c.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
        c.render_rows(p5, R.RGB_ASCII, 0, 8)          # uploads the scene
        c.synchronize()
        c.set_option(R.OPT_KERNEL, R.KERNEL_AUTO)
        c.set_option(R.OPT_TWO_LEVEL, -1)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        got5 = c.render_to_host(p5, R.RGB_ASCII)
        c.render(p5, R.RGB_ASCII)
        c.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        assert O.fnv1a64(got5) == U.load_golden()["C5_RGB_ASCII"]["frame_fnv1a64"]
        assert free0 - free1 <= 8 * 1024 * 1024, "two-level scratch for C5 took %.1f MB" % ((free0 - free1) / 1e6)


# ---------------------------------------------------------------- per-tile plane visibility (plane_invisible)

@pytest.mark.parametrize("seed", range(8))
def test_planes_left_out_of_a_tile_are_never_hit_there(R, ctx, seed):
    """The culling kernels leave a plane out of a macro tile's table when no pixel ray of the tile can hit it (facing
    away, behind the origin, or its bounded extent off the tile).  Random bounded planes of every orientation --
    grazing, behind the camera, containing the camera, tiny and huge -- and random cameras, all tile shapes: the
    binned frames must equal the oracle's (and the brute kernel's, which tests every plane at every pixel)."""
    rng = np.random.default_rng(31000 + seed)
    w, h = [(400, 150), (333, 97), (640, 200), (129, 257), (1280, 90), (257, 129), (512, 512), (97, 403)][seed]
    pos = rng.uniform(-20, 20, 3)
    rot = (float(rng.uniform(-1.4, 1.4)), float(rng.uniform(0, 6.28)), 0.0)
    p = R.camera_params(w, h, [float(v) for v in pos], rot)
    ctx.scene_clear()
    sc = O.Scene()
    for i in range(14):
        kind = i % 7
        centre = pos + rng.normal(0, 1, 3) * rng.choice([3.0, 30.0, 120.0])
        nrm = rng.normal(0, 1, 3)
        if kind == 0:
            nrm = np.array([0.0, 1.0, 0.0])                       # floors and ceilings
        elif kind == 1:
            nrm = np.array([0.0, -1.0, 0.0])
        elif kind == 2:
            centre = pos + np.array([0.0, -1e-3, 0.0])            # the camera (almost) on the plane
            nrm = np.array([0.0, 1.0, 0.0])
        elif kind == 3:
            nrm = np.array([1.0, rng.normal(0, 0.02), 0.0])       # walls: bounded in x by a thin slab
        width, height = float(rng.choice([0.5, 8.0, 60.0, 500.0])), float(rng.choice([0.5, 8.0, 60.0, 500.0]))
        col = [float(v) for v in np.floor(rng.uniform(1, 256, 3))]
        args = ([float(v) for v in centre], [float(v) for v in nrm], col, width, height)
        ctx.add_plane(*args)
        sc.add_plane(*args)
        if i % 3 == 0:
            sph = ([float(v) for v in centre + rng.normal(0, 2, 3)], float(rng.uniform(0.5, 4.0)), col)
            ctx.add_sphere(sph[1], sph[0], sph[2])
            sc.add_sphere(sph[1], sph[0], sph[2])
    want = O.render(U.oracle_params(p), sc, O.RGB_ASCII, threads=8)
    assert int((want.reshape(h, w, 20)[:, :w - 1, 2] == ord("3")).sum()) > 0, "the test scene shows nothing"
    for kernel, tile, sub in (("brute", 0, 0), ("binned", 0, 0), ("binned", 2, 1), ("binned", 6, 4), ("binned", 4, 16), ("binned", 3, 8)):
        set_kernel(R, ctx, kernel, tile, sub)
        got = ctx.render_to_host(p, R.RGB_ASCII)
        assert_same(got, want, O.RGB_ASCII, w, "planes seed %d %s tile %d sub %d" % (seed, kernel, tile, sub))
    set_kernel(R, ctx, "auto")


# ---------------------------------------------------------------- extreme magnitudes

@pytest.mark.parametrize("scale", [1e18, 3e18, 1e-18, 1e-16, 1e10, 1e-10])
def test_scenes_at_extreme_scales(R, ctx, scale):
    """The kernels reject a sphere on s*s - a*cc instead of the reference's b*b - 4a*cc (one multiply fewer); the two
    agree bit for bit except where a product overflows or goes subnormal (|o - c| around 1e19 or 1e-19).  There the
    rejection is only taken when it is certain, and otherwise the literal arithmetic decides: the whole scene scaled to
    those magnitudes (camera far plane and light stay where they are, as in the reference) must still give the oracle's
    frame from every kernel.  Also exercises the culling margins with infinite and NaN intermediates."""
    rng = np.random.default_rng(12)
    w, h = 200, 75
    p = R.camera_params(w, h, (0.0, 0.0, 0.0), (0.05, 3.1, 0.0))
    n = 120
    centres = np.stack([rng.uniform(-60, 60, n), rng.uniform(-20, 20, n), rng.uniform(20, 200, n)], axis=1) * scale
    radii = rng.uniform(2, 15, n) * scale
    cols = np.floor(rng.uniform(1, 256, (n, 3)))
    sph = np.concatenate([centres, radii[:, None], cols], axis=1).astype(np.float32)
    pl = np.array([[0, -30 * scale, 125 * scale, 0, 1, 0, 100, 100, 100, 3000 * scale, 250 * scale]], dtype=np.float32)
    ctx.set_scene(sph, pl)
    want = O.render(U.oracle_params(p), O.Scene.from_arrays(sph, pl), O.RGB_ASCII, threads=4)
    for kernel, two in (("brute", 0), ("binned", 0), ("binned", 1)):
        set_kernel(R, ctx, kernel, two_level=two)
        got = ctx.render_to_host(p, R.RGB_ASCII)
        assert_same(got, want, O.RGB_ASCII, w, "scale %g %s two-level %d" % (scale, kernel, two))
    set_kernel(R, ctx, "auto")


@pytest.mark.parametrize("W,H,n", [(7680, 4320, 1024), (7680, 4320, 16384), (3840, 2160, 4096)])
def test_wide_frames_with_turned_cameras_culling_equals_brute(R, W, H, n):
    """4K and 8K frames seen by turned cameras: every plan -- default, one level, two levels, refined -- against the brute kernel
    (every pixel tests every object, RayTracing.cu:100-136).  The reference's horizontal tangent extent grows with the frame
    (element1 = 0.577 H / 100: 25 at 8K), so tiles at the left and right edge are angularly thin (a 16-column tile at 8K: 1.7e-4
    rad); App. D scenes shrink their spheres towards the edge, which is why this test never caught the plane error that the
    directed test below constructs."""
    import torch
    rng = np.random.default_rng(7 + n)
    p0 = R.camera_params(W, H)
    sph, pl = R.synth_scene(100 + n, n, 1, p0.element1, p0.element2)
    a, b = R.Context(W, H), R.Context(W, H)
    try:
        for c in (a, b):
            c.set_scene(sph, pl)
        b.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
        got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        want = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        for view in range(5):
            rot = (float(rng.uniform(-0.3, 0.3)), float(np.pi + rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.3, 0.3)))
            pos = tuple(float(v) for v in rng.uniform(-2, 2, 3))
            p = R.camera_params(W, H, pos, rot)
            b.render_rows(p, R.RGB_ASCII, 0, H, d_out=want.data_ptr(), out_row_base=0)
            b.synchronize()
            for name, opts in (("auto", {}), ("one level", {R.OPT_TWO_LEVEL: 0}), ("two levels", {R.OPT_TWO_LEVEL: 1}),
                               ("two levels, refined", {R.OPT_TWO_LEVEL: 1, R.OPT_REFINE: 1, R.OPT_SUBTILES: 2})):
                a.set_option(R.OPT_TWO_LEVEL, -1)
                a.set_option(R.OPT_REFINE, -1)
                a.set_option(R.OPT_SUBTILES, 0)
                for k, v in opts.items():
                    a.set_option(k, v)
                got.fill_(0xEE)
                torch.cuda.synchronize()   # (the fill runs on torch's stream, the launch on the context's: order them)
                a.render_rows(p, R.RGB_ASCII, 0, H, d_out=got.data_ptr(), out_row_base=0)
                a.synchronize()
                if not torch.equal(got, want):
                    diff = (got.view(H, W, 20) != want.view(H, W, 20)).any(dim=2)
                    ys, xs = torch.nonzero(diff, as_tuple=True)
                    raise AssertionError("%dx%d, %d spheres, view %d rot %r, %s (%s): %d pixels differ from the brute kernel's, columns %d..%d rows %d..%d"
                                         % (W, H, n, view, rot, name, a.last_kernel, int(diff.sum()), int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())))
    finally:
        a.close()
        b.close()


def test_side_planes_of_thin_tiles_with_a_rolled_camera_at_8k(R):
    """A directed scene for the culling pyramids' side planes (tools/wide_view_directed_gpu.py builds it): a camera matrix with roll
    (any matrix is legal through rtx_params::inv_v; the reference's own Camera3D never rolls), 7680 x 4320, 16 x 16 tiles; for the
    tiles of the first 16 columns the host emulates the fp32 cross product of corner directions that rounds 1-2 took as a side
    plane's normal, picks the tiles whose own boundary-row pixel rays it leaves furthest outside (5e-5 .. 7e-5 rad; half a pixel
    is 5.3e-6 there), and puts one sphere (r = 20, 200 away) per frame so that its extreme point pokes 2.5 rows into such a tile.
    On the build before rtxplan::edge_basis the culling kernel dropped those rows in 5 of 6 frames (14 .. 46 pixels each, whole
    tile widths); with y P + Qr / -x P + Qc every frame equals the brute kernel's."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from wide_view_directed_gpu import directed_scene
    W, H = 7680, 4320
    p, sph, report = directed_scene(R, W, H, (0.1, 2.8, 0.3), (1.0, 2.0, -1.0))
    assert len(sph) >= 6, report
    pl = np.zeros((0, 11), dtype=np.float32)
    got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
    want = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
    for i in range(6):
        a, b = R.Context(W, H), R.Context(W, H)
        try:
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
            assert a.last_kernel == "rtx_trace<RTX_K_RGB_ASCII,true>"
            shaded = int((want.view(H, W, 20)[:, :16, 2] == ord('3')).sum().item())
            assert shaded > 1000, "the directed sphere is not where it was put"
            if not torch.equal(got, want):
                diff = (got.view(H, W, 20) != want.view(H, W, 20)).any(dim=2)
                ys, xs = torch.nonzero(diff, as_tuple=True)
                raise AssertionError("directed sphere %d (%s): %d pixels differ from the brute kernel's, rows %d..%d columns %d..%d"
                                     % (i, report[i], int(diff.sum()), int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())))
        finally:
            a.close()
            b.close()

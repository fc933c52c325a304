"""GPU Minimize, UpdateObjects and the whole-Update call against the oracle
(RayTracingManager.cu:10-44, 76-154, 167-319)."""
import numpy as np
import pytest

import oracle as O
import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    return U.pkg()


@pytest.fixture(scope="module")
def ctx(R):
    c = R.Context(1920, 1080)
    yield c
    c.close()


@pytest.mark.parametrize("mode", range(5))
@pytest.mark.parametrize("res", [(400, 150), (97, 41)])
def test_minimize_matches_oracle_default_scene(R, ctx, res, mode):
    import torch
    w, h = res
    ctx.set_reference_default_scene()
    p = R.camera_params(w, h)
    ctx.render(p, mode)
    dst = torch.zeros(20 * w * h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the context renders on its own non-blocking stream
    n = ctx.minimize(mode, w, h, d_out=dst.data_ptr())
    frame = ctx.read_frame(20 * w * h)
    want = O.minimize(mode, frame, w, h)
    got = dst.cpu().numpy()[:n]
    assert n == want.size
    assert np.array_equal(got, want)


def test_minimize_c2_full_size_against_golden(R, ctx):
    import torch
    gold = U.load_golden()
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    W, H = int(p.x), int(p.y)
    for mode in (R.RGB_ASCII, R.BIT_ASCII):
        ctx.render(p, mode)
        dst = torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the context renders on its own non-blocking stream
        n = ctx.minimize(mode, W, H, d_out=dst.data_ptr())
        g = gold["C2_%s" % R.MODE_NAMES[mode]]
        assert n == g["minimized_bytes"]
        assert O.fnv1a64(dst.cpu().numpy()[:n]) == g["minimized_fnv1a64"]


def test_minimize_sdl_and_empty_rows(R, ctx):
    import torch
    w, h = 64, 9
    ctx.set_reference_default_scene()
    p = R.camera_params(w, h)
    # SDL: nothing is written, the minimised stream is one newline per row
    ctx.render(p, R.RGB_ASCII)
    ctx.render(p, R.SDL)
    dst = torch.zeros(20 * w * h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the context renders on its own non-blocking stream
    n = ctx.minimize(R.SDL, w, h, d_out=dst.data_ptr())
    assert bytes(dst.cpu().numpy()[:n]) == b"\n" * h
    # a frame with only some rows rendered (others NUL): colour persistence skips the empty rows
    frame = torch.zeros(20 * w * h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the context renders on its own non-blocking stream
    ctx.render_rows(p, R.RGB_PIXEL, 2, 2, d_out=frame.data_ptr(), out_row_base=0)
    ctx.render_rows(p, R.RGB_PIXEL, 6, 1, d_out=frame.data_ptr(), out_row_base=0)
    ctx.synchronize()
    n = ctx.minimize(R.RGB_PIXEL, w, h, d_in=frame.data_ptr(), d_out=dst.data_ptr())
    want = O.minimize(O.RGB_PIXEL, frame.cpu().numpy(), w, h)
    assert np.array_equal(dst.cpu().numpy()[:n], want)


def test_minimize_width_one_and_two(R, ctx):
    import torch
    ctx.set_reference_default_scene()
    for (w, h) in ((1, 5), (2, 7), (3, 3)):
        p = R.camera_params(w, h)
        ctx.render(p, R.BIT_ASCII)
        dst = torch.zeros(20 * w * h + 4, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the context renders on its own non-blocking stream
        n = ctx.minimize(R.BIT_ASCII, w, h, d_out=dst.data_ptr())
        want = O.minimize(O.BIT_ASCII, ctx.read_frame(20 * w * h), w, h)
        assert np.array_equal(dst.cpu().numpy()[:n], want)


def test_update_is_the_reference_update(R, ctx):
    """rtx_update = params, zero semantics, trace, minimise, copy: the bytes the reference hands to
    PrintMachine::SetDataInBackBuffer (RayTracingManager.cu:150)."""
    ctx.set_reference_default_scene()
    p = R.camera_params(400, 150)
    for mode in (R.BIT_ASCII, R.RGB_ASCII, R.RGB_NORMALS, R.BIT_PIXEL):
        got = ctx.update(p, mode)
        frame = O.render(U.oracle_params(p), O.Scene.reference_default(), mode)
        want = O.minimize(mode, frame, 400, 150)
        assert np.array_equal(got, want), R.MODE_NAMES[mode]


def test_update_objects_matches_oracle_over_many_steps(R, ctx):
    """Sphere::Update (Sphere.cu:15-23) with double dt, through several bounces; 1500 spheres so the
    launch shape that breaks the reference past 1024 objects is exercised.

    Sphere::Update is declared with a `long double dt` parameter (Sphere.cu:15).  In device code -- where the reference
    calls it, from the UpdateObjects kernel (RayTracingManager.cu:10-44) -- `long double` IS `double` (CUDA and HIP
    alike have no 80-bit type on the device), and the caller passes a double anyway (RayTracingManager.cu:76,103).
    So both the oracle and the kernel evaluate y += (double)(speed * mover) * dt in binary64: that is the reference's
    device arithmetic, not a simplification; an x87 host evaluation would differ and is not what the reference runs."""
    rng = np.random.default_rng(7)
    n = 1500
    sph = np.concatenate([rng.uniform(-40, 40, (n, 3)), rng.uniform(0.5, 3, (n, 1)), np.floor(rng.uniform(1, 256, (n, 3)))],
                         axis=1).astype(np.float32)
    ctx.set_scene(sph, np.zeros((0, 11), dtype=np.float32))
    ctx.add_plane((0.0, -3.0, 30.0), (0.0, 1.0, 0.0), (100.0, 100.0, 100.0), 10.0, 20.0)
    sc = O.Scene.from_arrays(sph, np.zeros((0, 11), dtype=np.float32))
    sc.add_plane((0.0, -3.0, 30.0), (0.0, 1.0, 0.0), (100.0, 100.0, 100.0), 10, 20)
    speeds = (rng.integers(100, 400, n) / 100.0).astype(np.float32)  # Sphere.cu:11-12: (rand() % 300 + 100) / 100
    for i in range(n):
        ctx.set_sphere_motion(i, -1, float(speeds[i]))
        sc.objects()[i].speed = float(speeds[i])
        sc.objects()[i].mover = -1
    for dt in (0.016, 0.25, 1.7, 0.0333333, 3.0, 0.5, 0.016):
        ctx.update_objects(dt)
        O.lib().orc_update_objects(sc.ptrs(), sc.count, dt)
    for i in list(range(0, n, 37)) + [n - 1]:
        t, v = ctx.get_object(i)
        o = sc.objects()[i]
        assert t == 2
        assert v[1] == np.float32(o.center.y) and int(v[7]) == o.mover, i
    t, v = ctx.get_object(n)
    assert t == 1 and v[1] == -3.0
    # and the frame after physics equals the oracle's frame of the moved scene
    p = R.camera_params(200, 75)
    got = ctx.render_to_host(p, R.RGB_ASCII)
    want = O.render(U.oracle_params(p), sc, O.RGB_ASCII, threads=4)
    assert np.array_equal(got, want)


def test_update_with_physics_default_scene(R, ctx):
    ctx.set_reference_default_scene()
    sc = O.Scene.reference_default()
    p = R.camera_params(400, 150)
    for k in range(3):
        got = ctx.update(p, R.RGB_ASCII, dt=0.05, run_physics=True)
        O.lib().orc_update_objects(sc.ptrs(), sc.count, 0.05)
        frame = O.render(U.oracle_params(p), sc, O.RGB_ASCII)
        want = O.minimize(O.RGB_ASCII, frame, 400, 150)
        assert np.array_equal(got, want), k


def test_reference_api_surface_end_to_end(R, tmp_path):
    """examples/headless_engine.cpp drives the path through the reference's own class names
    (RayTracingManager, Scene3D, Camera3D, PrintMachine over include/rtx_compat.hpp), three frames with
    physics; its back buffer must be what the oracle's Update sequence produces."""
    import os
    import subprocess
    exe = os.path.join(R.PKG_DIR, "headless_engine")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    out = tmp_path / "frame.bin"
    for mode, frames, dt in ((R.BIT_ASCII, 3, 0.05), (R.RGB_ASCII, 2, 0.25)):
        subprocess.check_call([exe, "400", "150", str(frames), str(mode), str(dt), str(out)])
        got = np.fromfile(out, dtype=np.uint8)
        sc = O.Scene.reference_default()
        p = O.camera_params(400, 150)
        for _ in range(frames):
            O.lib().orc_update_objects(sc.ptrs(), sc.count, dt)
            frame = O.render(p, sc, mode)
        want = O.minimize(mode, frame, 400, 150)
        assert np.array_equal(got, want)


def test_frames_in_flight_are_each_complete(R, ctx):
    """rtx_submit_frames: several frames queued on different streams into different buffers, with different
    cameras, may overlap on the GPU; each must equal the frame rendered alone."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    W, H = int(p.x), int(p.y)
    cams = [R.camera_params(W, H, (float(i), 0.5 * i, -1.0 * i), (0.02 * i, float(np.float32(np.pi)) + 0.03 * i, 0.0)) for i in range(4)]
    want = [O.fnv1a64(ctx.render_to_host(c, R.RGB_ASCII)) for c in cams]
    streams = [torch.cuda.Stream() for _ in range(4)]
    bufs = [torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    for _ in range(3):  # the same ring three times over
        ctx.submit_frames(cams, R.RGB_ASCII, [b.data_ptr() for b in bufs], [s.cuda_stream for s in streams])
    torch.cuda.synchronize()
    for i in range(4):
        assert O.fnv1a64(bufs[i].cpu().numpy()) == want[i], i
    assert want[0] == U.load_golden()["C2_RGB_ASCII"]["frame_fnv1a64"]


def test_slab_rounds_fork_and_join(R, ctx):
    """rtx_submit_slabs as one rank of a 4-way row split uses it: the rank's rows of 4 consecutive frames
    (4 cameras) queued on 2 streams, forked from and joined into a third stream on which the slabs are then
    consumed at once (here: copied into whole frames, standing in for the RCCL exchange).  Done for each of
    the 4 'ranks' in turn, the frames must equal the ones rendered in one piece."""
    import importlib
    import torch
    sharding = importlib.import_module("raytracing-in-windows-console_amd.sharding")
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    W, H, S, N = int(p.x), int(p.y), 20, 4
    cams = [R.camera_params(W, H, (0.5 * i, 0.25 * i, 0.0), (0.01 * i, float(np.float32(np.pi)) - 0.02 * i, 0.0)) for i in range(N)]
    want = [O.fnv1a64(ctx.render_to_host(c, R.RGB_ASCII)) for c in cams]
    bounds = sharding.row_bounds(H, N)
    after = torch.cuda.Stream()
    rstreams = [torch.cuda.Stream() for _ in range(2)]
    frames = [torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda") for _ in range(N)]
    torch.cuda.synchronize()
    for rep in range(3):      # the send buffer is reused at once: the fork must order the next slabs after the copies
        for g in range(N):
            r0, rows = bounds[g], bounds[g + 1] - bounds[g]
            slab = S * W * rows
            if rep == 0 and g == 0:
                send = torch.zeros(N * S * W * (bounds[1] - bounds[0] + 1), dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
            ctx.submit_slabs(cams, R.RGB_ASCII, r0, rows, [send.data_ptr() + j * slab for j in range(N)], r0,
                             [rstreams[j % 2].cuda_stream for j in range(N)], after.cuda_stream)
            with torch.cuda.stream(after):
                for j in range(N):
                    frames[j][r0 * W * S:(r0 + rows) * W * S].copy_(send[j * slab:(j + 1) * slab], non_blocking=True)
    torch.cuda.synchronize()
    for j in range(N):
        assert O.fnv1a64(frames[j].cpu().numpy()) == want[j], j


def test_frames_in_flight_with_two_level_culling(R, ctx):
    """Large scene (pre-pass + per-stream scratch): overlapping frames must not share culling lists."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    ctx.set_scene(sph, pl)
    ctx.set_option(R.OPT_TWO_LEVEL, 1)
    try:
        W, H = int(p.x), int(p.y)
        cams = [R.camera_params(W, H, (3.0 * i, 0.0, 0.0), (0.0, float(np.float32(np.pi)) - 0.05 * i, 0.0)) for i in range(3)]
        want = [O.fnv1a64(ctx.render_to_host(c, R.RGB_ASCII)) for c in cams]
        streams = [torch.cuda.Stream() for _ in range(3)]
        bufs = [torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda") for _ in range(3)]
        torch.cuda.synchronize()
        for _ in range(4):
            ctx.submit_frames(cams, R.RGB_ASCII, [b.data_ptr() for b in bufs], [s.cuda_stream for s in streams])
        torch.cuda.synchronize()
        for i in range(3):
            assert O.fnv1a64(bufs[i].cpu().numpy()) == want[i], i
    finally:
        ctx.set_option(R.OPT_TWO_LEVEL, -1)


def test_pinned_host_buffer_roundtrip(R, ctx):
    ctx.set_reference_default_scene()
    p = R.camera_params(400, 150)
    a = np.array(ctx.update(p, R.RGB_PIXEL), copy=True)
    b = np.array(ctx.update(p, R.RGB_PIXEL), copy=True)
    assert a.size > 0 and np.array_equal(a, b)


def test_pipelined_update_matches_synchronous_update(R, ctx):
    """rtx_update_begin/_end with two frames in flight (copy of frame k overlapping the trace of k+1) must
    deliver, frame by frame, the bytes of the synchronous Update, physics included."""
    ctx.set_reference_default_scene()
    sc = O.Scene.reference_default()
    p = R.camera_params(400, 150)
    nbytes = 20 * 400 * 150
    bufs = [ctx.host_alloc(nbytes) for _ in range(2)]
    try:
        want = []
        for k in range(5):
            O.lib().orc_update_objects(sc.ptrs(), sc.count, 0.04)
            mode = [R.RGB_ASCII, R.BIT_ASCII, R.RGB_PIXEL, R.SDL, R.BIT_PIXEL][k]
            frame = O.render(U.oracle_params(p), sc, mode)
            want.append(O.minimize(mode, frame, 400, 150))
        tickets = []
        got = []
        for k in range(5):
            mode = [R.RGB_ASCII, R.BIT_ASCII, R.RGB_PIXEL, R.SDL, R.BIT_PIXEL][k]
            if len(tickets) == 2:  # two in flight: retire the older one
                t, idx = tickets.pop(0)
                n = ctx.update_end(t)
                got.append(np.array(bufs[idx][1][:n], copy=True))
            t = ctx.update_begin(p, mode, bufs[k % 2][0], dt=0.04, run_physics=True)
            tickets.append((t, k % 2))
        for t, idx in tickets:
            n = ctx.update_end(t)
            got.append(np.array(bufs[idx][1][:n], copy=True))
        assert len(got) == 5
        for k in range(5):
            assert np.array_equal(got[k], want[k]), k
    finally:
        for ptr, _ in bufs:
            ctx.host_free(ptr)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("config", ["C1", "odd"])
def test_compact_words_expand_to_the_same_records(R, ctx, mode, config):
    """RTX_RENDER_COMPACT + rtx_expand == the records rtx_render_rows writes == the oracle's, every character mode;
    the words themselves are checked through the numpy restatement of the 4-byte form (tests/util.py)."""
    import torch
    if config == "C1":
        p, sph, pl = R.config_inputs("C1")
    else:
        _, sph, pl = R.config_inputs("C1")
        p = R.camera_params(333, 77)      # odd width: record rows are not 16-byte multiples (slow store path)
    ctx.set_scene(sph, pl)
    W, H = int(p.x), int(p.y)
    S = 20 if mode >= R.RGB_ASCII else 12
    want = O.render(U.oracle_params(p), O.Scene.from_arrays(sph, pl), mode)[:S * W * H]
    words = torch.full((W * H,), 0x7F7F7F7F, dtype=torch.int32, device="cuda")
    out = torch.full((S * W * H + 16,), 0xEE, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.render_rows(p, mode, 0, H, d_out=words.data_ptr(), out_row_base=0, flags=R.RENDER_COMPACT)
    ctx.expand(mode, words.data_ptr(), out.data_ptr(), [(0, 0, W * H)])
    ctx.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[:S * W * H], want), U.first_diff(got[:S * W * H], want, S, W)
    assert (got[S * W * H:] == 0xEE).all()          # nothing written past the last record
    w = words.cpu().numpy().view(np.uint32)
    assert np.array_equal(U.words_to_records(w, S, ord('3') if mode in (0, 2) else ord('4')), want)
    assert (w.reshape(H, W)[:, W - 1] == 0xFFFFFFFF).all()
    # several segments, out of order, into a destination that starts mid-buffer: rows swapped pairwise
    segs = []
    for r in range(H):
        segs.append((r * W, (r ^ 1) * W if (r ^ 1) < H else r * W, W))
    out2 = torch.zeros(S * W * H, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.expand(mode, words.data_ptr(), out2.data_ptr(), segs)
    ctx.synchronize()
    got2 = out2.cpu().numpy().reshape(H, W * S)
    wantr = want.reshape(H, W * S)
    for r in range(H):
        rr = (r ^ 1) if (r ^ 1) < H else r
        assert np.array_equal(got2[rr], wantr[r]), r


def test_compact_rejects_what_it_cannot_do(R, ctx):
    import torch
    ctx.set_reference_default_scene()
    p = R.camera_params(64, 16)
    buf = torch.zeros(20 * 64 * 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(R.RtxError):
        ctx.render_rows(p, R.RGB_ASCII, 0, 16, flags=R.RENDER_COMPACT)                       # context's own buffer
    with pytest.raises(R.RtxError):
        ctx.render_rows(p, R.SDL, 0, 16, d_out=buf.data_ptr(), flags=R.RENDER_COMPACT)      # SDL writes nothing
    with pytest.raises(R.RtxError):
        ctx.expand(R.SDL, buf.data_ptr(), buf.data_ptr(), [(0, 0, 16)])
    with pytest.raises(R.RtxError):
        ctx.expand(R.RGB_ASCII, buf.data_ptr() + 2, buf.data_ptr(), [(0, 0, 16)])            # misaligned


def test_hip_graph_replays_a_round_of_slabs_and_their_expansion(R, ctx):
    """rtx_graph_begin / _end / _launch: a round of the row-sharded loop -- the slab launches of several frames forked over
    render streams and joined back, then the expansion of the compact words into records -- recorded once and replayed
    with one host call each; every replay must leave exactly the bytes the direct calls leave.  What cannot be recorded
    (the two-level pre-pass) is refused and the stream leaves capture mode."""
    import torch
    p0, sph, pl = R.config_inputs("C2")
    ctx.set_option(R.OPT_KERNEL, R.KERNEL_AUTO)
    ctx.set_option(R.OPT_TWO_LEVEL, -1)
    ctx.set_scene(sph, pl)
    W, H = int(p0.x), int(p0.y)
    row0, rows, n = 405, 270, 5
    params = [R.camera_params(W, H, pos=(0.1 * i, 0.0, 0.0), rot=(0.0, np.pi + 0.02 * i, 0.0)) for i in range(n)]
    main = torch.cuda.Stream()
    streams = [torch.cuda.Stream() for _ in range(3)]
    words = torch.zeros(n, W * rows, dtype=torch.int32, device="cuda")
    recs = torch.zeros(n, 20 * W * rows, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    segs = [(0, 0, W * rows)]

    def queue_round():
        ctx.submit_slabs(params, R.RGB_ASCII, row0, rows, [words[i].data_ptr() for i in range(n)], row0,
                         [streams[i % 3].cuda_stream for i in range(n)], after=main.cuda_stream, flags=R.RENDER_COMPACT)
        for i in range(n):
            ctx.expand(R.RGB_ASCII, words[i].data_ptr(), recs[i].data_ptr(), segs, stream=main.cuda_stream)

    queue_round()                         # also uploads the scene: a capture must not have to
    torch.cuda.synchronize()
    want = recs.cpu().numpy().copy()
    for i in range(n):                    # the records are what the oracle renders for that frame's camera
        if i in (0, n - 1):
            o = O.render(U.oracle_params(params[i]), O.Scene.from_arrays(sph, pl), O.RGB_ASCII, threads=8, )
            assert np.array_equal(want[i], o[20 * W * row0:20 * W * (row0 + rows)])
    ctx.graph_begin(main.cuda_stream)
    queue_round()
    g = ctx.graph_end(main.cuda_stream)
    replay = ctx.graph_launcher(g, main.cuda_stream)
    for _ in range(3):
        words.zero_()
        recs.fill_(0xEE)
        torch.cuda.synchronize()
        replay()
        torch.cuda.synchronize()
        assert np.array_equal(recs.cpu().numpy(), want)
    ctx.graph_destroy(g)
    # not recordable: the two-level pre-pass; the error is reported and the stream is usable afterwards
    ctx.set_option(R.OPT_TWO_LEVEL, 1)
    ctx.graph_begin(main.cuda_stream)
    with pytest.raises(R.RtxError) as e:
        ctx.render_rows(params[0], R.RGB_ASCII, row0, rows, d_out=recs[0].data_ptr(), out_row_base=row0, stream=main.cuda_stream)
    assert e.value.status == R.ERR_INVALID_ARGUMENT
    try:
        ctx.graph_destroy(ctx.graph_end(main.cuda_stream))
    except R.RtxError:
        pass                              # an empty or invalidated capture may be reported; either way it has ended
    ctx.set_option(R.OPT_TWO_LEVEL, -1)
    recs.fill_(0xEE)
    torch.cuda.synchronize()
    queue_round()
    torch.cuda.synchronize()
    assert np.array_equal(recs.cpu().numpy(), want)


# ---- Minimize from compact pixel words (rtx_minimize_words; what rtx_update runs by default, RTX_OPT_UPDATE_WORDS)

@pytest.mark.parametrize("mode", range(5))
@pytest.mark.parametrize("res", [(400, 150), (97, 41), (1, 5), (2, 7), (3, 3), (1030, 9)])
def test_minimize_from_words_is_minimize_of_the_records(R, ctx, res, mode):
    """The words the trace kernel stores (RTX_RENDER_COMPACT) through rtx_minimize_words against the oracle's Minimize8bit /
    MinimizeRGB (RayTracingManager.cu:181-319) of the oracle's frame: same bytes, same length."""
    import torch
    w, h = res
    ctx.set_reference_default_scene()
    p = R.camera_params(w, h)
    words = torch.empty(w * h, dtype=torch.int32, device="cuda")
    dst = torch.zeros(20 * w * h + 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.render_rows(p, mode, 0, h, d_out=words.data_ptr(), out_row_base=0, flags=R.RENDER_COMPACT)
    ctx.synchronize()
    n = ctx.minimize_words(mode, w, h, words.data_ptr(), d_out=dst.data_ptr())
    want = O.minimize(mode, O.render(U.oracle_params(p), O.Scene.reference_default(), mode), w, h)
    assert n == want.size and np.array_equal(dst.cpu().numpy()[:n], want)


def test_minimize_from_words_with_empty_slots_and_rows(R, ctx):
    """Words of a frame that was only partly rendered (0xffffffff = an empty slot: NUL as a record): colour persistence skips
    the empty slots, across rows and across blocks of the pass, exactly as the byte scan does on the expanded records."""
    import torch
    w, h = 300, 40
    ctx.set_reference_default_scene()
    p = R.camera_params(w, h)
    rng = np.random.default_rng(5)
    for mode in (R.RGB_ASCII, R.BIT_PIXEL):
        S = 20 if mode >= R.RGB_ASCII else 12
        words = torch.empty(w * h, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.render_rows(p, mode, 0, h, d_out=words.data_ptr(), out_row_base=0, flags=R.RENDER_COMPACT)
        ctx.synchronize()
        hw = words.cpu().numpy().view(np.uint32).copy().reshape(h, w)
        hw[3:9, :] = 0xFFFFFFFF                          # whole rows empty (longer than a block of the pass: 6 x 300 > 1024)
        hw[20, 5:290] = 0xFFFFFFFF                       # most of a row
        holes = rng.random((h, w)) < 0.3
        holes[:, w - 1] = False
        hw[holes] = 0xFFFFFFFF
        hw[0, 0] = 0xFFFFFFFF                            # the frame's first slot
        words.copy_(torch.from_numpy(hw.reshape(-1).view(np.int32)))
        dst = torch.zeros(20 * w * h, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        n = ctx.minimize_words(mode, w, h, words.data_ptr(), d_out=dst.data_ptr())
        frame = np.zeros(20 * w * h, dtype=np.uint8)
        frame[:S * w * h] = U.words_to_records(hw.reshape(-1), S, ord("3") if mode in (R.RGB_ASCII, R.BIT_ASCII) else ord("4"))
        want = O.minimize(mode, frame, w, h)
        assert n == want.size and np.array_equal(dst.cpu().numpy()[:n], want), R.MODE_NAMES[mode]


def test_update_from_words_and_from_records_hand_over_the_same_stream(R):
    """rtx_update traces pixel words and minimises from them by default (RTX_OPT_UPDATE_WORDS): the stream is the golden one
    (C2, both record sizes) and the frame buffer is left alone; with the option off the records are written and minimised, as
    the reference does, and the frame buffer holds the frame."""
    gold = U.load_golden()
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    with R.Context(W, H) as c:
        c.set_scene(sph, pl)
        assert c.get_option(R.OPT_UPDATE_WORDS) == -1
        for mode in (R.RGB_ASCII, R.BIT_ASCII):
            g = gold["C2_%s" % R.MODE_NAMES[mode]]
            got = c.update(p, mode)
            assert "compact" in c.last_kernel
            assert len(got) == g["minimized_bytes"] and O.fnv1a64(got) == g["minimized_fnv1a64"]
        assert not c.read_frame(20 * W * H).any()                      # never written
        c.set_option(R.OPT_UPDATE_WORDS, 0)
        for mode in (R.RGB_ASCII, R.BIT_ASCII):
            g = gold["C2_%s" % R.MODE_NAMES[mode]]
            got = c.update(p, mode)
            assert "compact" not in c.last_kernel
            assert len(got) == g["minimized_bytes"] and O.fnv1a64(got) == g["minimized_fnv1a64"]
            assert O.fnv1a64(c.read_frame(20 * W * H)) == g["frame_fnv1a64"]
        # SDL has no words: one newline per row either way
        c.set_option(R.OPT_UPDATE_WORDS, -1)
        assert bytes(c.update(p, R.SDL)) == b"\n" * H


# ---- the one-launch form of Minimize from words (rtx_minw_fused, RTX_OPT_MINIMIZE_FUSED)

@pytest.mark.parametrize("res", [(1024, 1), (1024, 63), (1024, 64), (1024, 65), (1024, 129), (1920, 1080), (3840, 2160), (977, 331), (5, 3)])
def test_fused_minimize_is_the_three_launch_minimize(R, ctx, res):
    """One launch with a two-level look-back (64 blocks per group: frames of 1, 63, 64, 65, 129, 2025 and 8100 blocks) against the
    three launches (RTX_OPT_MINIMIZE_FUSED = 0) on the same words, every mode family, with and without empty slots; the small
    sizes also against the oracle's Minimize of the expanded records."""
    import torch
    w, h = res
    rng = np.random.default_rng(w * 7919 + h)
    for mode, holes in ((R.RGB_ASCII, 0.0), (R.BIT_PIXEL, 0.0), (R.RGB_NORMALS, 0.05)):
        S = 20 if mode >= R.RGB_ASCII else 12
        hw = U.random_words(rng, w, h, runs=0.7, holes=holes)
        if mode < R.RGB_ASCII:
            hw = np.where((hw != 0) & (hw != 0xFFFFFFFF), hw & np.uint32(0xFF0000FF), hw).astype(np.uint32)   # an index and a glyph
        words = torch.from_numpy(hw.view(np.int32)).cuda()
        out = {}
        for fused in (0, 1):
            ctx.set_option(R.OPT_MINIMIZE_FUSED, fused)
            dst = torch.full((S * w * h + 16,), 0xEE, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            n = ctx.minimize_words(mode, w, h, words.data_ptr(), d_out=dst.data_ptr())
            got = dst.cpu().numpy()
            assert (got[n:] == 0xEE).all(), "bytes written past the stream"
            out[fused] = got[:n].copy()
        ctx.set_option(R.OPT_MINIMIZE_FUSED, -1)
        assert out[0].size == out[1].size and np.array_equal(out[0], out[1]), (res, R.MODE_NAMES[mode])
        assert ctx.get_option(R.STAT_MINIMIZE_FALLBACKS) == 0
        if w * h <= 1024 * 129:
            frame = np.zeros(20 * w * h, dtype=np.uint8)
            frame[:S * w * h] = U.words_to_records(hw, S, ord("3") if mode in (R.RGB_ASCII, R.BIT_ASCII) else ord("4"))
            want = O.minimize(mode, frame, w, h)
            assert out[1].size == want.size and np.array_equal(out[1], want), (res, R.MODE_NAMES[mode])


def test_fused_minimize_that_gives_up_is_redone(R):
    """RTX_OPT_MINIMIZE_FUSED = 2: a third of the blocks give up on purpose, a group's last block among them poisons its total, the
    failure word reaches the host and the frame is minimised by the three launches: same stream, the fallback is counted -- through
    rtx_minimize_words, rtx_update and the pipelined rtx_update_begin / _end."""
    import torch
    gold = U.load_golden()
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    g = gold["C2_RGB_ASCII"]
    with R.Context(W, H) as c:
        c.set_scene(sph, pl)
        assert c.get_option(R.OPT_MINIMIZE_FUSED) == -1
        got = c.update(p, R.RGB_ASCII)
        assert len(got) == g["minimized_bytes"] and O.fnv1a64(got) == g["minimized_fnv1a64"]
        assert c.get_option(R.STAT_MINIMIZE_FALLBACKS) == 0
        c.set_option(R.OPT_MINIMIZE_FUSED, 2)
        got = c.update(p, R.RGB_ASCII)
        assert len(got) == g["minimized_bytes"] and O.fnv1a64(got) == g["minimized_fnv1a64"]
        assert c.get_option(R.STAT_MINIMIZE_FALLBACKS) == 1
        host = [c.host_alloc(20 * W * H) for _ in range(2)]
        try:
            t0 = c.update_begin(p, R.RGB_ASCII, host[0][0])
            t1 = c.update_begin(p, R.RGB_ASCII, host[1][0])
            for t, hb in ((t0, host[0]), (t1, host[1])):
                n = c.update_end(t)
                assert n == g["minimized_bytes"] and O.fnv1a64(hb[1][:n]) == g["minimized_fnv1a64"]
        finally:
            for ptr, _ in host:
                c.host_free(ptr)
        assert c.get_option(R.STAT_MINIMIZE_FALLBACKS) == 3
        c.set_option(R.OPT_MINIMIZE_FUSED, 1)
        got = c.update(p, R.RGB_ASCII)
        assert len(got) == g["minimized_bytes"] and O.fnv1a64(got) == g["minimized_fnv1a64"]
        assert c.get_option(R.STAT_MINIMIZE_FALLBACKS) == 3
        with pytest.raises(R.RtxError):
            c.set_option(R.OPT_MINIMIZE_FUSED, 3)


def test_fused_minimize_beside_other_launches(R):
    """The one-launch Minimize waits only for blocks of smaller index, which the hardware dispatches first -- also when another
    context keeps the GPU's workgroup slots busy with frames in flight on four streams of its own.  150 Updates of config 2 beside a
    stream of 4K frames: the golden stream every time, and not one launch that gave up and was redone."""
    import torch
    gold = U.load_golden()["C2_RGB_ASCII"]
    p2, sph2, pl2 = R.config_inputs("C2")
    p3, sph3, pl3 = R.config_inputs("C3")
    W3, H3 = int(p3.x), int(p3.y)
    with R.Context(int(p2.x), int(p2.y)) as c, R.Context(W3, H3) as busy:
        c.set_scene(sph2, pl2)
        busy.set_scene(sph3, pl3)
        streams = [torch.cuda.Stream() for _ in range(4)]
        bufs = [torch.empty(20 * W3 * H3, dtype=torch.uint8, device="cuda") for _ in range(4)]
        sub = busy.make_submitter([p3] * 8, R.RGB_ASCII, [bufs[i % 4].data_ptr() for i in range(8)], [streams[i % 4].cuda_stream for i in range(8)])
        torch.cuda.synchronize()
        for k in range(150):
            if k % 3 == 0:
                sub(8, 0)                   # ~0.7 ms of 4K frames queued beside the next Updates
            got = c.update(p2, R.RGB_ASCII)
            assert len(got) == gold["minimized_bytes"], k
            if k % 25 == 0:
                assert O.fnv1a64(got) == gold["minimized_fnv1a64"], k
        torch.cuda.synchronize()
        assert O.fnv1a64(c.update(p2, R.RGB_ASCII)) == gold["minimized_fnv1a64"]
        assert c.get_option(R.STAT_MINIMIZE_FALLBACKS) == 0


# ---- the record form of Minimize as one launch (rtx_min_fused)

@pytest.mark.parametrize("res", [(1024, 1), (1024, 64), (1024, 65), (1920, 1080), (977, 331), (5, 3), (1, 7)])
def test_fused_record_minimize_is_the_two_launch_minimize(R, ctx, res):
    """rtx_minimize (records in, RayTracingManager.cu:167-319) as one launch against the two launches (RTX_OPT_MINIMIZE_FUSED = 0) and,
    for the smaller frames, the oracle: frames made of random words expanded to records (so that escapes are elided and slots are
    empty), both record sizes; then with blocks that give up (2): the same bytes, the fallback counted."""
    import torch
    w, h = res
    rng = np.random.default_rng(w * 31 + h)
    for mode, holes in ((R.RGB_ASCII, 0.0), (R.BIT_PIXEL, 0.1), (R.RGB_PIXEL, 0.6)):
        S = 20 if mode >= R.RGB_ASCII else 12
        hw = U.random_words(rng, w, h, runs=0.6, holes=holes)
        if mode < R.RGB_ASCII:
            hw = np.where((hw != 0) & (hw != 0xFFFFFFFF), hw & np.uint32(0xFF0000FF), hw).astype(np.uint32)
        frame = np.zeros(20 * w * h, dtype=np.uint8)
        frame[:S * w * h] = U.words_to_records(hw, S, ord("3") if mode in (R.RGB_ASCII, R.BIT_ASCII) else ord("4"))
        src = torch.from_numpy(frame).cuda()
        out = {}
        before = ctx.get_option(R.STAT_MINIMIZE_FALLBACKS)
        for fused in (0, 1, 2):
            ctx.set_option(R.OPT_MINIMIZE_FUSED, fused)
            dst = torch.full((S * w * h + 16,), 0xEE, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            n = ctx.minimize(mode, w, h, src.data_ptr(), dst.data_ptr())
            got = dst.cpu().numpy()
            assert (got[n:] == 0xEE).all(), "bytes written past the stream"
            out[fused] = got[:n].copy()
        ctx.set_option(R.OPT_MINIMIZE_FUSED, -1)
        assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2]), (res, R.MODE_NAMES[mode])
        assert ctx.get_option(R.STAT_MINIMIZE_FALLBACKS) - before == (1 if w * h > 1024 else 0)
        if w * h <= 1024 * 65:
            want = O.minimize(mode, frame, w, h)
            assert out[1].size == want.size and np.array_equal(out[1], want), (res, R.MODE_NAMES[mode])


# ---- the blocking Update whose Minimize launch writes the caller's pinned buffer itself (RTX_OPT_UPDATE_HOST_WRITE)

def test_update_host_write_hands_over_the_same_stream(R):
    """Small frames take it by default (one host wait per Update): the oracle's stream in every mode, with physics steps in between,
    equal to the copy form (option 0); a pageable buffer quietly goes the usual way; forced on at 1080p it is still the golden stream;
    a Minimize whose blocks give up is redone into the same host buffer."""
    import ctypes as C
    sc = O.Scene.reference_default()
    W, H = 400, 150
    p = R.camera_params(W, H)
    with R.Context(W, H) as c:
        c.set_reference_default_scene()
        assert c.get_option(R.OPT_UPDATE_HOST_WRITE) == -1
        for k, mode in enumerate([R.BIT_ASCII, R.RGB_ASCII, R.RGB_PIXEL, R.BIT_PIXEL, R.RGB_NORMALS, R.SDL]):
            O.lib().orc_update_objects(sc.ptrs(), sc.count, 0.03)
            want = O.minimize(mode, O.render(U.oracle_params(p), sc, mode), W, H)
            before = c.get_option(R.STAT_UPDATE_HOST_WRITES)
            got = c.update(p, mode, dt=0.03, run_physics=True).copy()
            assert got.size == want.size and np.array_equal(got, want), R.MODE_NAMES[mode]
            assert c.get_option(R.STAT_UPDATE_HOST_WRITES) - before == (0 if mode == R.SDL else 1)
            c.set_option(R.OPT_UPDATE_HOST_WRITE, 0)
            again = c.update(p, mode).copy()                      # no physics step: the same scene, the copy form
            assert np.array_equal(again, want)
            c.set_option(R.OPT_UPDATE_HOST_WRITE, -1)
        # blocks that give up: the three launches write the same host buffer
        c.set_option(R.OPT_MINIMIZE_FUSED, 2)
        want = O.minimize(R.RGB_ASCII, O.render(U.oracle_params(p), sc, R.RGB_ASCII), W, H)
        assert np.array_equal(c.update(p, R.RGB_ASCII), want) and c.get_option(R.STAT_MINIMIZE_FALLBACKS) == 1
        c.set_option(R.OPT_MINIMIZE_FUSED, -1)
        # the pipelined form the same way: nothing is waited for in rtx_update_begin; blocks that give up are redone in rtx_update_end
        bufs = [c.host_alloc(20 * W * H) for _ in range(2)]
        try:
            for fused in (-1, 2):
                c.set_option(R.OPT_MINIMIZE_FUSED, fused)
                before = c.get_option(R.STAT_UPDATE_HOST_WRITES)
                t0 = c.update_begin(p, R.RGB_ASCII, bufs[0][0])
                t1 = c.update_begin(p, R.BIT_ASCII, bufs[1][0])
                n0, n1 = c.update_end(t0), c.update_end(t1)
                assert np.array_equal(bufs[0][1][:n0], want)
                assert np.array_equal(bufs[1][1][:n1], O.minimize(R.BIT_ASCII, O.render(U.oracle_params(p), sc, R.BIT_ASCII), W, H))
                assert c.get_option(R.STAT_UPDATE_HOST_WRITES) - before == 2
            c.set_option(R.OPT_MINIMIZE_FUSED, -1)
        finally:
            for ptr, _ in bufs:
                c.host_free(ptr)
        # a pageable destination: not device-addressable, so the stream is copied as before
        before = c.get_option(R.STAT_UPDATE_HOST_WRITES)
        buf = np.zeros(20 * W * H, dtype=np.uint8)
        n = C.c_size_t()
        rc = R.lib().rtx_update(c._h, C.byref(p), R.RGB_ASCII, 0.0, 0, buf.ctypes.data_as(C.c_void_p), C.byref(n))
        assert rc == 0 and n.value == want.size and np.array_equal(buf[:n.value], want)
        assert c.get_option(R.STAT_UPDATE_HOST_WRITES) == before
        with pytest.raises(R.RtxError):
            c.set_option(R.OPT_UPDATE_HOST_WRITE, 2)
    gold = U.load_golden()["C2_RGB_ASCII"]
    p2, sph2, pl2 = R.config_inputs("C2")
    with R.Context(int(p2.x), int(p2.y)) as c:
        c.set_scene(sph2, pl2)
        got = c.update(p2, R.RGB_ASCII)
        assert c.get_option(R.STAT_UPDATE_HOST_WRITES) == 1        # the blocking form: at any size
        assert len(got) == gold["minimized_bytes"] and O.fnv1a64(got) == gold["minimized_fnv1a64"]
        buf = c.host_alloc(20 * int(p2.x) * int(p2.y))
        try:
            n = c.update_end(c.update_begin(p2, R.RGB_ASCII, buf[0]))
            assert c.get_option(R.STAT_UPDATE_HOST_WRITES) == 1    # the pipelined form at 1080p: the copy form
            assert n == gold["minimized_bytes"] and O.fnv1a64(buf[1][:n]) == gold["minimized_fnv1a64"]
            c.set_option(R.OPT_UPDATE_HOST_WRITE, 1)
            n = c.update_end(c.update_begin(p2, R.RGB_ASCII, buf[0]))
            assert c.get_option(R.STAT_UPDATE_HOST_WRITES) == 2
            assert n == gold["minimized_bytes"] and O.fnv1a64(buf[1][:n]) == gold["minimized_fnv1a64"]
            c.set_option(R.OPT_UPDATE_HOST_WRITE, 0)
            got = c.update(p2, R.RGB_ASCII)
            assert c.get_option(R.STAT_UPDATE_HOST_WRITES) == 2
            assert len(got) == gold["minimized_bytes"] and O.fnv1a64(got) == gold["minimized_fnv1a64"]
        finally:
            c.host_free(buf[0])

"""Device groups (rtx_group_create, include/rtx.h): the row-sharded frame BEHIND the C ABI -- SURVEY.md 8(b) `rtx_create(ndev, ...)`,
8(e) "single process, one stream per device, ncclCommInitAll".  The reference's one consumer is RayTracingManager::Update
(RayTracingManager.cu:76-154, hand-off at :150) and rows are contiguous byte ranges of the frame (RayTracing.cu:238,457); a group
context must deliver on its root, byte for byte, what one device delivers.

This box has ONE GPU: groups of 4 and 8 logical ranks are built from a device list that repeats device 0 (every rank its own
member context, scene replica, stream and slab; the gather is hipMemcpyPeerAsync between buffers of one device), and the RCCL
form is walked at ndev = 1 with RTX_EXCHANGE_RCCL_ALL (the root's own slab is sent to itself through ncclSend / ncclRecv).  What a
real 8-GPU node adds is distinct devices under the same code; that is the driver's SCALE run."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle as O
import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    return U.pkg()


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("ranks,wire", [(4, "compact"), (8, "compact"), (8, "records"), (3, "records")])
def test_group_frame_of_config_2_is_the_golden_frame(R, ranks, wire):
    """rtx_render on a group: rank g traces rows [g H / N, (g+1) H / N) with the global row index, the slabs are gathered on the
    root; the root's frame buffer then holds the golden C2 frame (oracle-generated), whole 20*W*H buffer, both wire formats."""
    gold = U.load_golden()
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    with R.Context(W, H, devices=[0] * ranks) as c:
        assert c.group_size == ranks and c.get_option(R.STAT_GROUP_SIZE) == ranks
        assert [c.group_rows(H, r) for r in range(ranks)] == [(H * r // ranks, H * (r + 1) // ranks - H * r // ranks) for r in range(ranks)]
        c.set_option(R.OPT_GROUP_WIRE, R.WIRE_COMPACT if wire == "compact" else R.WIRE_RECORDS)
        c.set_scene(sph, pl)
        for mode in (R.RGB_ASCII, R.BIT_ASCII, R.RGB_ASCII):      # (8-bit after RGB: the tail must be NUL again)
            got = c.render_to_host(p, mode)
            assert _sha(got) == gold["C2_%s" % R.MODE_NAMES[mode]]["frame_sha256"], (R.MODE_NAMES[mode], wire)
        assert c.get_option(R.STAT_GROUP_GATHERS) == 3
        S = 4 if wire == "compact" else 20
        assert c.get_option(R.STAT_GROUP_BYTES) == (H - H // ranks) * W * S     # every slab but the root's crossed
        assert c.get_option(R.STAT_GROUP_EXCHANGE) == R.EXCHANGE_PEER_COPY and "repeats a device" in c.exchange_note
        for r in range(ranks):
            assert "rtx_trace" in c.member_kernel(r)               # every rank launched a trace kernel of its own


def test_group_of_eight_renders_config_4_as_8_slabs_of_540_rows(R):
    """SURVEY 8(e)'s case: 7680x4320 over 8 ranks, 540 rows = 82 944 000 bytes of records each."""
    gold = U.load_golden()["C4_RGB_ASCII"]
    p, sph, pl = R.config_inputs("C4")
    W, H = int(p.x), int(p.y)
    with R.Context(W, H, devices=[0] * 8) as c:
        c.set_scene(sph, pl)
        assert all(c.group_rows(H, r) == (540 * r, 540) for r in range(8))
        for wire in (R.WIRE_COMPACT, R.WIRE_RECORDS):
            c.set_option(R.OPT_GROUP_WIRE, wire)
            assert O.fnv1a64(c.render_to_host(p, R.RGB_ASCII)) == gold["frame_fnv1a64"]
        assert c.get_option(R.STAT_GROUP_BYTES) == 7 * 540 * W * 20


def test_submit_frames_on_a_group_at_8k(R):
    """Two 8K frames (config 4) per call over 8 ranks: the chunk's slabs -- 540 rows x 7680 words, 16.6 MB each -- travel as one
    strided copy per rank with a 132 MB pitch."""
    import torch
    gold = U.load_golden()["C4_RGB_ASCII"]
    p, sph, pl = R.config_inputs("C4")
    W, H = int(p.x), int(p.y)
    bufs = [torch.empty(20 * W * H, dtype=torch.uint8, device="cuda") for _ in range(2)]
    with R.Context(W, H, devices=[0] * 8) as g:
        g.set_scene(sph, pl)
        for b in bufs:
            b.fill_(0xEE)
        torch.cuda.synchronize()
        g.submit_frames([p, p], R.RGB_ASCII, [b.data_ptr() for b in bufs], [None, None])
        g.synchronize()
        for b in bufs:
            assert O.fnv1a64(b.cpu().numpy()) == gold["frame_fnv1a64"]
        assert g.get_option(R.STAT_GROUP_BYTES) == 7 * 540 * W * 4


@pytest.mark.parametrize("ranks,W,H", [(7, 400, 150), (5, 97, 41), (8, 64, 3), (2, 333, 1), (16, 320, 180)])
def test_ragged_and_empty_slabs_against_the_oracle(R, ranks, W, H):
    """H not a multiple of N (ragged slabs), more ranks than rows (empty slabs), every mode, both wires: the frame and the
    minimised stream of rtx_update equal the oracle's."""
    sc = O.Scene.reference_default()
    p = R.camera_params(W, H)
    with R.Context(W, H, devices=[0] * ranks) as c:
        c.set_reference_default_scene()
        for wire in (R.WIRE_COMPACT, R.WIRE_RECORDS):
            c.set_option(R.OPT_GROUP_WIRE, wire)
            for mode in range(6):
                want = O.render(U.oracle_params(p), sc, mode)
                got = c.render_to_host(p, mode)
                assert np.array_equal(got, want), (R.MODE_NAMES[mode], wire, U.first_diff(got, want, 20 if mode >= 2 else 12, W))
                assert np.array_equal(c.update(p, mode), O.minimize(mode, want, W, H)), (R.MODE_NAMES[mode], wire)
        # the record form of Update over the group (RTX_OPT_UPDATE_WORDS = 0): the gathered frame is minimised, not the words
        c.set_option(R.OPT_UPDATE_WORDS, 0)
        for mode in (R.RGB_ASCII, R.BIT_PIXEL):
            want = O.render(U.oracle_params(p), sc, mode)
            assert np.array_equal(c.update(p, mode), O.minimize(mode, want, W, H)), R.MODE_NAMES[mode]
            assert np.array_equal(c.read_frame(20 * W * H), want)


def test_update_through_the_group_is_the_single_device_stream(R):
    """RayTracingManager::Update's sequence (physics, trace, minimise, hand-off) through a group of 4: scene edits, sphere motion
    and every physics step reach all replicas, so the stream stays byte-equal to a single-device context's (and the oracle's)
    frame after frame; the pipelined form (rtx_update_begin / _end) as well."""
    W, H, n = 640, 360, 1500
    rng = np.random.default_rng(11)
    p = R.camera_params(W, H)
    sph, pl = R.synth_scene(9, n, 1, p.element1, p.element2)
    speeds = (rng.integers(50, 400, n) / 100.0).astype(np.float32)
    sc = O.Scene.from_arrays(sph, pl)
    with R.Context(W, H, devices=[0, 0, 0, 0]) as g, R.Context(W, H) as one:
        for c in (g, one):
            c.set_scene(sph, pl)
            for i in range(0, n, 3):
                c.set_sphere_motion(i, 1 if i % 2 else -1, float(speeds[i]))
        for i in range(0, n, 3):
            sc.objects()[i].speed = float(speeds[i])
            sc.objects()[i].mover = 1 if i % 2 else -1
        assert g.object_count == one.object_count == n + 1
        for f in range(6):
            mode = (R.RGB_ASCII, R.BIT_ASCII, R.RGB_PIXEL)[f % 3]
            a = g.update(p, mode, dt=0.05, run_physics=True).copy()
            b = one.update(p, mode, dt=0.05, run_physics=True).copy()
            O.lib().orc_update_objects(sc.ptrs(), sc.count, 0.05)
            assert np.array_equal(a, b), f
            if f in (0, 5):
                assert np.array_equal(a, O.minimize(mode, O.render(U.oracle_params(p), sc, mode, threads=8), W, H)), f
            if f == 2:   # an edit in the middle: the new sphere lands in every replica, at the same creation index
                for c in (g, one):
                    assert c.add_sphere(3.0, (0.0, 0.0, 30.0), (200.0, 100.0, 50.0)) == n + 1
                sc.add_sphere(3.0, (0.0, 0.0, 30.0), (200.0, 100.0, 50.0))
        # every replica holds the same sphere positions as the single device
        for i in (0, 3, 999, n + 1):
            t0, v0 = one.get_object(i)
            assert t0 == g.get_object(i)[0] and np.array_equal(v0, g.get_object(i)[1])
        # pipelined Update
        hb = [g.host_alloc(20 * W * H) for _ in range(2)]
        tick = []
        outs = []
        for f in range(4):
            if len(tick) == 2:
                t, k = tick.pop(0)
                outs.append(hb[k][1][:g.update_end(t)].copy())
            tick.append((g.update_begin(p, R.RGB_ASCII, hb[f % 2][0], dt=0.05, run_physics=True), f % 2))
        while tick:
            t, k = tick.pop(0)
            outs.append(hb[k][1][:g.update_end(t)].copy())
        for f in range(4):
            assert np.array_equal(outs[f], one.update(p, R.RGB_ASCII, dt=0.05, run_physics=True)), f
        for ptr, _ in hb:
            g.host_free(ptr)


def test_rccl_exchange_at_one_device_and_the_fallbacks(R):
    """The RCCL form of the gather, walked on the one GPU this box has: RTX_EXCHANGE_RCCL_ALL sends the root's own slab to
    itself (ncclCommInitAll over [0], grouped ncclSend / ncclRecv on the root's stream) -- communicator set-up, the group call
    and the stream ordering are the code an 8-GPU node runs.  A device list that repeats a device cannot have a communicator
    (one rank per GPU): asking for RCCL there is refused, AUTO takes peer copies."""
    gold = U.load_golden()
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    with R.Context(W, H, devices=[0]) as c:
        c.set_scene(sph, pl)
        c.set_option(R.OPT_GROUP_EXCHANGE, R.EXCHANGE_RCCL_ALL)
        for wire in (R.WIRE_COMPACT, R.WIRE_RECORDS):
            c.set_option(R.OPT_GROUP_WIRE, wire)
            for mode in (R.RGB_ASCII, R.BIT_ASCII):
                assert _sha(c.render_to_host(p, mode)) == gold["C2_%s" % R.MODE_NAMES[mode]]["frame_sha256"], (wire, mode)
                assert c.get_option(R.STAT_GROUP_EXCHANGE) == R.EXCHANGE_RCCL, c.exchange_note
        assert c.get_option(R.STAT_GROUP_BYTES) == H * W * 12
        got = c.update(p, R.RGB_ASCII)
        assert len(got) == gold["C2_RGB_ASCII"]["minimized_bytes"] and O.fnv1a64(got) == gold["C2_RGB_ASCII"]["minimized_fnv1a64"]
    with R.Context(64, 32, devices=[0, 0]) as c:
        with pytest.raises(R.RtxError):
            c.set_option(R.OPT_GROUP_EXCHANGE, R.EXCHANGE_RCCL)
        c.set_option(R.OPT_GROUP_EXCHANGE, R.EXCHANGE_PEER_COPY)
    with R.Context(64, 32) as c:      # a plain context is a group of one, without the group options
        assert c.group_size == 1 and c.get_option(R.STAT_GROUP_SIZE) == 1
        with pytest.raises(R.RtxError):
            c.set_option(R.OPT_GROUP_WIRE, R.WIRE_RECORDS)


def test_submission_threads_on_and_off_render_the_same_frames(R):
    """RTX_OPT_GROUP_THREADS: a rank's launch and copy are queued by a thread of its own (1; the default where the list names
    distinct devices) or by the caller's (0; the default on this one-GPU box); the frames, the minimised stream and the error
    reporting are the same."""
    gold = U.load_golden()
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    with R.Context(W, H, devices=[0] * 6) as c:
        c.set_scene(sph, pl)
        assert c.get_option(R.OPT_GROUP_THREADS) == -1
        for threads in (-1, 0, 1, 0, -1):
            c.set_option(R.OPT_GROUP_THREADS, threads)
            for wire in (R.WIRE_COMPACT, R.WIRE_RECORDS):
                c.set_option(R.OPT_GROUP_WIRE, wire)
                for _ in range(3):
                    got = c.render_to_host(p, R.RGB_ASCII)
                assert _sha(got) == gold["C2_RGB_ASCII"]["frame_sha256"], (threads, wire)
            st = c.update(p, R.RGB_ASCII)
            assert len(st) == gold["C2_RGB_ASCII"]["minimized_bytes"] and O.fnv1a64(st) == gold["C2_RGB_ASCII"]["minimized_fnv1a64"], threads
        with pytest.raises(R.RtxError) as e:
            c.render(R.camera_params(W, H + 1), R.RGB_ASCII)
        assert e.value.status == R.ERR_TOO_LARGE
        assert _sha(c.render_to_host(p, R.RGB_ASCII)) == gold["C2_RGB_ASCII"]["frame_sha256"]   # and the group goes on working


def test_options_reach_every_rank_and_errors_name_the_rank(R):
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    with R.Context(W, H, devices=[0, 0, 0]) as c:
        c.set_scene(sph, pl)
        c.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
        assert all(c.member_option(r, R.OPT_KERNEL) == R.KERNEL_BRUTE for r in range(3))
        brute = c.render_to_host(p, R.RGB_ASCII)
        assert all("false" in c.member_kernel(r) for r in range(3)), [c.member_kernel(r) for r in range(3)]
        c.set_option(R.OPT_KERNEL, R.KERNEL_AUTO)
        assert np.array_equal(brute, c.render_to_host(p, R.RGB_ASCII))
        big = R.camera_params(W + 1, H)
        with pytest.raises(R.RtxError) as e:
            c.render(big, R.RGB_ASCII)
        assert e.value.status == R.ERR_TOO_LARGE


def test_reference_api_surface_over_a_group(R, tmp_path):
    """examples/headless_engine.cpp -- Engine3D::Start / Render through RayTracingManager, Scene3D, Camera3D, PrintMachine of
    include/rtx_compat.hpp, unchanged -- with RTX_DEVICES=0,0,0,0 in the environment: the facade creates a device group, and the
    back buffer is what the same binary produces on one device (and what the oracle's Update sequence produces)."""
    exe = os.path.join(R.PKG_DIR, "headless_engine")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    for mode, frames, dt in ((R.BIT_ASCII, 3, 0.05), (R.RGB_ASCII, 2, 0.25)):
        outs = []
        for devs in (None, "0,0,0,0", "0"):
            out = tmp_path / ("frame_%s.bin" % (devs or "single").replace(",", ""))
            env = dict(os.environ)
            env.pop("RTX_DEVICES", None)
            if devs:
                env["RTX_DEVICES"] = devs
            subprocess.check_call([exe, "400", "150", str(frames), str(mode), str(dt), str(out)], env=env)
            outs.append(np.fromfile(out, dtype=np.uint8))
        sc = O.Scene.reference_default()
        p = O.camera_params(400, 150)
        for _ in range(frames):
            O.lib().orc_update_objects(sc.ptrs(), sc.count, dt)
            frame = O.render(p, sc, mode)
        want = O.minimize(mode, frame, 400, 150)
        for got in outs:
            assert np.array_equal(got, want)


@pytest.mark.parametrize("ranks,wire", [(8, "compact"), (4, "records"), (3, "compact")])
def test_submit_frames_on_a_group_shards_a_chunk_of_frames_per_call(R, ranks, wire):
    """rtx_submit_frames on a group: n whole frames with n cameras, every rank tracing its rows of all of them with ONE batched
    launch (RTX_OPT_BATCH) and the chunk's slabs travelling together; each frame equals the frame a single device renders for
    that camera.  18 frames: more than one chunk (16 at most per launch)."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    n = 18
    cams = [R.camera_params(W, H, pos=(0.3 * i, 0.1 * i, -0.2 * i), rot=(0.004 * i, float(np.float32(np.pi)) + 0.01 * i, 0.0)) for i in range(n)]
    bufs = [torch.empty(20 * W * H, dtype=torch.uint8, device="cuda") for _ in range(n)]
    side = torch.cuda.Stream()
    with R.Context(W, H) as one:
        one.set_scene(sph, pl)
        want = [O.fnv1a64(one.render_to_host(c, R.RGB_ASCII)) for c in cams]
    with R.Context(W, H, devices=[0] * ranks) as g:
        g.set_option(R.OPT_GROUP_WIRE, R.WIRE_COMPACT if wire == "compact" else R.WIRE_RECORDS)
        g.set_scene(sph, pl)
        for rep in range(2):
            for b in bufs:
                b.fill_(0xEE)
            torch.cuda.synchronize()
            g.submit_frames(cams, R.RGB_ASCII, [b.data_ptr() for b in bufs], [side.cuda_stream] * n)
            side.synchronize()          # the caller's stream was made to wait for the frames
            for i in range(n):
                assert O.fnv1a64(bufs[i].cpu().numpy()) == want[i], (rep, i)
        # every rank used the batched kernel, a launch per chunk: 16 frames at most, and what a rank's slab buffer (its 20*W*H
        # frame buffer) holds of 4-byte words or 20-byte records
        S = 4 if wire == "compact" else 20
        chunk = min([16] + [(20 * H) // (g.group_rows(H, r)[1] * S) for r in range(1, ranks)])
        per_call = sum(1 for c0 in range(0, n, chunk) if min(chunk, n - c0) >= 2)
        for r in range(ranks):
            assert g.member_option(r, R.STAT_BATCHED_LAUNCHES) == 2 * per_call, (r, chunk, g.member_kernel(r))
        assert g.get_option(R.STAT_GROUP_GATHERS) == 2 * n


# ---- rtx_update without a gather: every rank minimises and copies its own rows (RTX_OPT_GROUP_UPDATE)

@pytest.mark.parametrize("ranks,W,H", [(7, 400, 150), (5, 97, 41), (8, 64, 3), (2, 333, 1), (16, 320, 180), (3, 1, 9), (4, 2, 8)])
def test_direct_update_is_the_gathered_update(R, ranks, W, H):
    """RTX_OPT_GROUP_UPDATE = 1: each rank traces its rows and the row above them, minimises its own rows with the colour carried over
    from the last pixel above, and copies its part of the stream to its place in the host buffer.  Ragged and empty slabs, widths of
    one and two slots, every mode (SDL has no words: it takes the other path), both forms of the minimise pass: the bytes of the
    oracle's Minimize of the oracle's frame, and of the same group's gathered Update."""
    sc = O.Scene.reference_default()
    p = R.camera_params(W, H)
    with R.Context(W, H, devices=[0] * ranks) as c:
        c.set_reference_default_scene()
        assert c.get_option(R.OPT_GROUP_UPDATE) == -1          # auto: one device here, so the root gathers
        for mode in range(6):
            want = O.minimize(mode, O.render(U.oracle_params(p), sc, mode), W, H)
            c.set_option(R.OPT_GROUP_UPDATE, 0)
            gathered = c.update(p, mode).copy()
            before = c.get_option(R.STAT_GROUP_DIRECT_UPDATES)
            c.set_option(R.OPT_GROUP_UPDATE, 1)
            for fused in (1, 0):
                c.set_option(R.OPT_MINIMIZE_FUSED, fused)
                got = c.update(p, mode).copy()
                assert got.size == want.size and np.array_equal(got, want), (R.MODE_NAMES[mode], fused)
                assert np.array_equal(got, gathered)
            c.set_option(R.OPT_MINIMIZE_FUSED, -1)
            assert c.get_option(R.STAT_GROUP_DIRECT_UPDATES) - before == (0 if mode == R.SDL else 2)
        with pytest.raises(R.RtxError):
            c.set_option(R.OPT_GROUP_UPDATE, 2)
    with R.Context(W, H) as one:
        with pytest.raises(R.RtxError):
            one.set_option(R.OPT_GROUP_UPDATE, 1)            # not a group


def test_direct_update_c2_golden_with_physics_and_pipelined(R):
    """Config 2 through 8 logical ranks with RTX_OPT_GROUP_UPDATE = 1: the committed golden stream (both record sizes); then a moving
    scene, frame after frame against a single-device context, blocking and through rtx_update_begin / _end; and a fused minimise
    whose blocks give up (RTX_OPT_MINIMIZE_FUSED = 2) on every rank."""
    gold = U.load_golden()
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    with R.Context(W, H, devices=[0] * 8) as g:
        g.set_scene(sph, pl)
        g.set_option(R.OPT_GROUP_UPDATE, 1)
        for threads in (0, 1):      # everything queued, then awaited, by the caller's thread; a rank's thread queues and waits in one go
            g.set_option(R.OPT_GROUP_THREADS, threads)
            for mode in (R.RGB_ASCII, R.BIT_ASCII):
                ref = gold["C2_%s" % R.MODE_NAMES[mode]]
                got = g.update(p, mode)
                assert len(got) == ref["minimized_bytes"] and O.fnv1a64(got) == ref["minimized_fnv1a64"], (threads, R.MODE_NAMES[mode])
        g.set_option(R.OPT_MINIMIZE_FUSED, 2)
        got = g.update(p, R.RGB_ASCII)
        ref = gold["C2_RGB_ASCII"]
        assert len(got) == ref["minimized_bytes"] and O.fnv1a64(got) == ref["minimized_fnv1a64"]
        assert sum(g.member_option(r, R.STAT_MINIMIZE_FALLBACKS) for r in range(8)) == 8
    W, H, n = 640, 360, 900
    rng = np.random.default_rng(5)
    p = R.camera_params(W, H)
    sph, pl = R.synth_scene(4, n, 1, p.element1, p.element2)
    with R.Context(W, H, devices=[0, 0, 0]) as g, R.Context(W, H) as one:
        speeds = (rng.integers(50, 300, n) / 100.0).astype(np.float32)
        for c in (g, one):
            c.set_scene(sph, pl)
            for i in range(0, n, 2):
                c.set_sphere_motion(i, 1 if i % 4 else -1, float(speeds[i]))
        g.set_option(R.OPT_GROUP_UPDATE, 1)
        for f in range(4):
            mode = (R.RGB_ASCII, R.BIT_PIXEL)[f % 2]
            a = g.update(p, mode, dt=0.05, run_physics=True).copy()
            b = one.update(p, mode, dt=0.05, run_physics=True).copy()
            assert a.size == b.size and np.array_equal(a, b), f
        nbytes = 20 * W * H
        bufs = [g.host_alloc(nbytes) for _ in range(2)] + [one.host_alloc(nbytes)]
        try:
            tickets = []
            for f in range(4):
                if len(tickets) == 2:
                    t, idx = tickets.pop(0)
                    na = g.update_end(t)
                    tb = one.update_begin(p, R.RGB_ASCII, bufs[2][0], dt=0.05, run_physics=True)
                    nb = one.update_end(tb)
                    assert na == nb and np.array_equal(bufs[idx][1][:na], bufs[2][1][:nb]), f
                tickets.append((g.update_begin(p, R.RGB_ASCII, bufs[f % 2][0], dt=0.05, run_physics=True), f % 2))
            for t, idx in tickets:
                na = g.update_end(t)
                tb = one.update_begin(p, R.RGB_ASCII, bufs[2][0], dt=0.05, run_physics=True)
                nb = one.update_end(tb)
                assert na == nb and np.array_equal(bufs[idx][1][:na], bufs[2][1][:nb])
        finally:
            for c, (ptr, _) in ((g, bufs[0]), (g, bufs[1]), (one, bufs[2])):
                c.host_free(ptr)
        assert g.get_option(R.STAT_GROUP_DIRECT_UPDATES) == 8

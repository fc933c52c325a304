"""Round 3 on the GPU: coarse-cell lists that outlive a frame (RTX_OPT_CELL_REUSE), the XCD-aware dispatch order of two-level
grids (RTX_OPT_XCD_ORDER), HIP graphs that refuse a scene they were not recorded on, and dispatch orders frozen by a recorded
launch.  The reference has none of these (its Culling kernel is an empty stub, RayTracingManager.cu:46-51, and it launches
one kernel per frame on the default stream, :127-134); what they must preserve is the frame, byte for byte -- checked against
the brute kernel (every pixel tests every object, RayTracing.cu:100-136) and, where it is quick enough, the oracle."""
import math

import numpy as np
import pytest

import oracle as O
import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    return U.pkg()


def _stats(R, c):
    return {k: c.get_option(v) for k, v in (("builds", R.STAT_CELL_BUILDS), ("prefetches", R.STAT_CELL_PREFETCHES),
                                            ("hits", R.STAT_CELL_HITS), ("per_frame", R.STAT_CELL_PER_FRAME))}


def _dense_scene(R, W, H, n, seed):
    p = R.camera_params(W, H)
    sph, _ = R.synth_scene(seed, n, 0, p.element1, p.element2)
    return sph, np.zeros((0, 11), dtype=np.float32)


def _pair(R, W, H, sph, pl):
    """(context under test, brute-kernel context of the same scene)."""
    a, b = R.Context(W, H), R.Context(W, H)
    for c in (a, b):
        c.set_scene(sph, pl)
    b.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
    return a, b


def _frame(c, p, mode, buf, stream=None):
    import torch
    buf.fill_(0xEE)
    torch.cuda.synchronize()
    c.render_rows(p, mode, 0, int(p.y), d_out=buf.data_ptr(), out_row_base=0, stream=stream,
                  flags=0 if mode >= O.RGB_ASCII else 1)


@pytest.mark.parametrize("seed,n,W,H", [(5, 8192, 640, 360), (11, 3000, 480, 270), (13, 2600, 1280, 720)])
def test_cell_lists_reused_across_a_moving_camera_equal_the_brute_kernel(R, seed, n, W, H):
    """A camera that rests, creeps (turning and translating), races and rests again: every frame -- served by cached lists,
    by lists built ahead of time on the side stream, by an in-line rebuild or by the per-frame pre-pass -- equals the brute
    kernel's frame of the same camera, and each of those four ways is actually taken."""
    import torch
    sph, pl = _dense_scene(R, W, H, n, seed)
    a, b = _pair(R, W, H, sph, pl)
    try:
        assert a.get_option(R.OPT_CELL_REUSE) == -1            # on by default
        got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        want = torch.empty_like(got)
        yaw, pos = math.pi, [0.0, 0.0, 0.0]
        phases = [("rest", 12, 0.0, 0.0), ("creep", 60, 4.0e-4, 0.01), ("race", 8, 0.03, 0.5), ("rest", 10, 0.0, 0.0),
                  ("creep back", 40, -7.0e-4, -0.02)]
        seen = {}
        for name, frames, dyaw, dpos in phases:
            s0 = _stats(R, a)
            for f in range(frames):
                yaw += dyaw
                pos[0] += dpos
                pos[2] += 0.5 * dpos
                p = R.camera_params(W, H, pos=tuple(pos), rot=(0.0, yaw, 0.0))
                for mode in ((O.RGB_ASCII,) if f % 7 else (O.RGB_ASCII, O.BIT_ASCII)):
                    _frame(a, p, mode, got)
                    _frame(b, p, mode, want)
                    torch.cuda.synchronize()
                    assert torch.equal(got, want), "%s frame %d mode %d: %s" % (name, f, mode, U.first_diff(got.cpu().numpy(), want.cpu().numpy(), 20 if mode >= 2 else 12, W))
            s1 = _stats(R, a)
            seen[name] = {k: s1[k] - s0[k] for k in s1}
        assert "binned" in a.last_kernel or "true" in a.last_kernel
        assert seen["rest"]["builds"] <= 1 and seen["rest"]["hits"] >= 9 and seen["rest"]["per_frame"] == 0
        assert seen["creep"]["prefetches"] >= 3 and seen["creep"]["per_frame"] == 0 and seen["creep"]["hits"] >= 50
        assert seen["race"]["per_frame"] >= 6                  # too fast for reuse: the per-frame pre-pass
        assert seen["creep back"]["hits"] >= 30
    finally:
        a.close()
        b.close()


def test_cell_lists_shared_by_frames_in_flight_on_several_streams(R):
    """Four render streams, frames queued without waiting, the camera creeping: lists built on one stream (or the side stream)
    serve launches on the others; every frame of the ring equals the brute kernel's."""
    import torch
    W, H, n = 640, 360, 6000
    sph, pl = _dense_scene(R, W, H, n, 21)
    a, b = _pair(R, W, H, sph, pl)
    try:
        streams = [torch.cuda.Stream() for _ in range(4)]
        ring = [torch.empty(20 * W * H, dtype=torch.uint8, device="cuda") for _ in range(24)]
        want = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        frame_no = 0
        for rnd in range(6):
            for r in ring:
                r.fill_(0xEE)
            torch.cuda.synchronize()
            cams = []
            for i, r in enumerate(ring):
                step = 3.0e-4 if rnd % 2 == 0 else 0.0
                frame_no += 1
                p = R.camera_params(W, H, pos=(0.002 * frame_no, 0.0, 0.0), rot=(0.0, math.pi + step * frame_no, 0.0))
                cams.append(p)
                a.render_rows(p, O.RGB_ASCII, 0, H, d_out=r.data_ptr(), out_row_base=0, stream=streams[i % 4].cuda_stream)
            torch.cuda.synchronize()
            for i, r in enumerate(ring):
                _frame(b, cams[i], O.RGB_ASCII, want)
                torch.cuda.synchronize()
                assert torch.equal(r, want), "round %d frame %d" % (rnd, i)
        s = _stats(R, a)
        assert s["hits"] > 100 and s["prefetches"] >= 2
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("mover", [-1, 3, -7])
def test_cell_lists_with_bouncing_spheres(R, mover):
    """rtx_update_objects between frames (Sphere::Update, Sphere.cu:15-23): the first step after an edit may move a sphere
    from anywhere onto +-10 and counts as an edit; later steps move a sphere by at most |speed mover dt|, which the lists'
    position budget covers -- also for |mover| > 1, which the reference never holds (Sphere.cu:9,21) but the ABI accepts: the
    bound counts |speed * mover|, not |speed|.  Every frame equals the brute kernel's on a twin context stepped alike; the
    last one the oracle's."""
    import torch
    W, H, n = 320, 180, 2600
    rng = np.random.default_rng(3)
    sph = np.concatenate([rng.uniform(-60, 60, (n, 1)), rng.uniform(-25, 25, (n, 1)), rng.uniform(30, 160, (n, 1)),
                          rng.uniform(0.3, 2.0, (n, 1)), np.floor(rng.uniform(1, 256, (n, 3)))], axis=1).astype(np.float32)
    pl = np.zeros((0, 11), dtype=np.float32)
    a, b = _pair(R, W, H, sph, pl)
    sc = O.Scene.from_arrays(sph, pl)
    try:
        speeds = (rng.integers(100, 400, n) / 100.0).astype(np.float32)
        for i in range(n):
            for c in (a, b):
                c.set_sphere_motion(i, mover, float(speeds[i]))
            sc.objects()[i].speed = float(speeds[i])
            sc.objects()[i].mover = mover
        got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        want = torch.empty_like(got)
        p = R.camera_params(W, H)
        for f in range(40):
            dt = (0.016, 0.033, 0.004)[f % 3]
            for c in (a, b):
                c.update_objects(dt)
            O.lib().orc_update_objects(sc.ptrs(), sc.count, dt)
            _frame(a, p, O.RGB_ASCII, got)
            _frame(b, p, O.RGB_ASCII, want)
            torch.cuda.synchronize()
            assert torch.equal(got, want), "frame %d" % f
        s = _stats(R, a)
        if mover == -1:
            assert s["hits"] >= 25 and s["per_frame"] == 0
        else:
            assert s["hits"] + s["per_frame"] + s["builds"] >= 35   # (faster spheres: shorter-lived lists, or none)
        assert np.array_equal(got.cpu().numpy(), O.render(U.oracle_params(p), sc, O.RGB_ASCII, threads=8))
    finally:
        a.close()
        b.close()


def test_reuse_off_and_xcd_order_off_render_the_same_frames(R):
    import torch
    W, H, n = 640, 360, 8192
    sph, pl = _dense_scene(R, W, H, n, 9)
    a, b = _pair(R, W, H, sph, pl)
    try:
        got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        want = torch.empty_like(got)
        p = R.camera_params(W, H, pos=(1.0, 0.5, 0.0), rot=(0.02, math.pi + 0.1, 0.0))
        _frame(b, p, O.RGB_ASCII, want)
        for reuse, xcd, srt in ((0, -1, -1), (-1, 0, 0), (0, 0, 0), (1, 1, 1), (-1, -1, 0), (-1, -1, -1)):
            a.set_option(R.OPT_CELL_REUSE, reuse)
            a.set_option(R.OPT_XCD_ORDER, xcd)
            a.set_option(R.OPT_SORTED_STORE, srt)   # staging from the direction-sorted copy of the sphere array, or from the array itself
            s0 = _stats(R, a)
            for _ in range(3):
                _frame(a, p, O.RGB_ASCII, got)
                torch.cuda.synchronize()
                assert torch.equal(got, want), (reuse, xcd, srt)
            s1 = _stats(R, a)
            if reuse == 0:
                assert s1["per_frame"] - s0["per_frame"] == 3 and s1["hits"] == s0["hits"]
            else:
                assert s1["per_frame"] == s0["per_frame"] and s1["hits"] - s0["hits"] >= 2
    finally:
        a.close()
        b.close()


def test_config5_with_cached_lists_is_the_golden_frame(R):
    """BASELINE config 5 at full size: the first frame bins, the following ones reuse; all equal the committed golden hash."""
    import torch
    p, sph, pl = R.config_inputs("C5")
    W, H = int(p.x), int(p.y)
    gold = U.load_golden()["C5_RGB_ASCII"]["frame_fnv1a64"]
    c = R.Context(W, H)
    try:
        c.set_scene(sph, pl)
        buf = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        for i in range(4):
            _frame(c, p, O.RGB_ASCII, buf)
            torch.cuda.synchronize()
            assert O.fnv1a64(buf.cpu().numpy()) == gold, "frame %d" % i
        s = _stats(R, c)
        assert s["builds"] == 1 and s["hits"] == 3 and s["per_frame"] == 0
    finally:
        c.close()


def test_recorded_graph_is_refused_after_a_scene_edit(R):
    """A recorded launch keeps raw pointers into the scene arrays and the object counts of the moment it was recorded:
    after rtx_scene_add_* (here: enough spheres to make the store reallocate) rtx_graph_launch must refuse the graph
    instead of replaying kernels that read freed memory; rtx_update_objects moves spheres in place and is fine."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    c = R.Context(W, H)
    try:
        c.set_scene(sph, pl)
        st = torch.cuda.Stream()
        buf = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        c.render_rows(p, O.RGB_ASCII, 0, H, d_out=buf.data_ptr(), out_row_base=0, stream=st.cuda_stream)   # uploads the scene
        torch.cuda.synchronize()
        want = buf.clone()
        c.graph_begin(st.cuda_stream)
        c.render_rows(p, O.RGB_ASCII, 0, H, d_out=buf.data_ptr(), out_row_base=0, stream=st.cuda_stream)
        g = c.graph_end(st.cuda_stream)
        buf.fill_(0xEE)
        torch.cuda.synchronize()
        c.graph_launch(g, st.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(buf, want)
        # recorded launches need caller buffers: the context's own frame is refused, and the stream leaves capture usable
        c.graph_begin(st.cuda_stream)
        with pytest.raises(R.RtxError) as e:
            c.render_rows(p, O.RGB_ASCII, 0, H, stream=st.cuda_stream)
        assert e.value.status == R.ERR_INVALID_ARGUMENT
        try:
            c.graph_destroy(c.graph_end(st.cuda_stream))
        except R.RtxError:
            pass
        # more than 1024 new spheres: the device arrays are reallocated and the old ones freed
        extra = np.concatenate([np.random.default_rng(1).uniform(-50, 50, (1500, 3)), np.full((1500, 1), 0.5), np.full((1500, 3), 200.0)],
                               axis=1).astype(np.float32)
        c.add_spheres(extra)
        c.set_option(R.OPT_TWO_LEVEL, 0)   # (2525 spheres would switch the pre-pass on, which cannot be recorded at all)
        with pytest.raises(R.RtxError) as e:
            c.graph_launch(g, st.cuda_stream)
        assert e.value.status == R.ERR_INVALID_ARGUMENT and "re-capture" in str(e.value)
        c.graph_destroy(g)
        # a fresh recording on the edited scene works, and physics steps do not invalidate it
        c.render_rows(p, O.RGB_ASCII, 0, H, d_out=buf.data_ptr(), out_row_base=0, stream=st.cuda_stream)
        torch.cuda.synchronize()
        c.graph_begin(st.cuda_stream)
        c.render_rows(p, O.RGB_ASCII, 0, H, d_out=buf.data_ptr(), out_row_base=0, stream=st.cuda_stream)
        g2 = c.graph_end(st.cuda_stream)
        c.update_objects(0.01)
        c.synchronize()
        c.graph_launch(g2, st.cuda_stream)
        torch.cuda.synchronize()
        c.graph_destroy(g2)
    finally:
        c.close()


def test_recorded_launch_runs_under_the_converged_order_which_is_then_frozen(R):
    """A grid's dispatch order converges over its first launches on a stream (rtx_balance_tiles); a launch recorded after that
    keeps the order's address, so the library stops deriving orders for that (stream, grid): replays interleaved with a
    hundred direct launches on the same stream all render the golden frame, and no further pass is queued."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    gold = U.load_golden()["C2_RGB_ASCII"]["frame_fnv1a64"]
    c = R.Context(W, H)
    try:
        c.set_scene(sph, pl)
        st = torch.cuda.Stream()
        buf = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        buf2 = torch.empty_like(buf)
        for _ in range(90):
            c.render_rows(p, O.RGB_ASCII, 0, H, d_out=buf.data_ptr(), out_row_base=0, stream=st.cuda_stream)
        torch.cuda.synchronize()
        passes = c.get_option(R.STAT_ORDER_PASSES)
        assert passes >= 10 and c.get_option(R.STAT_ORDERS_FROZEN) == 0
        c.graph_begin(st.cuda_stream)
        c.render_rows(p, O.RGB_ASCII, 0, H, d_out=buf.data_ptr(), out_row_base=0, stream=st.cuda_stream)
        g = c.graph_end(st.cuda_stream)
        assert c.get_option(R.STAT_ORDERS_FROZEN) == 1
        for i in range(6):
            buf.fill_(0xEE)
            torch.cuda.synchronize()
            c.graph_launch(g, st.cuda_stream)
            for _ in range(20):   # direct launches of the same grid on the same stream, in between and in flight with the replays
                c.render_rows(p, O.RGB_ASCII, 0, H, d_out=buf2.data_ptr(), out_row_base=0, stream=st.cuda_stream)
            c.graph_launch(g, st.cuda_stream)
            torch.cuda.synchronize()
            assert O.fnv1a64(buf.cpu().numpy()) == gold and torch.equal(buf, buf2), "replay %d" % i
        assert c.get_option(R.STAT_ORDER_PASSES) == passes      # frozen: nothing was derived any more
        # two graphs may read the same order; it is released with the last of them, and balanced again from then on
        c.graph_begin(st.cuda_stream)
        c.render_rows(p, O.RGB_ASCII, 0, H, d_out=buf2.data_ptr(), out_row_base=0, stream=st.cuda_stream)
        g2 = c.graph_end(st.cuda_stream)
        assert c.get_option(R.STAT_ORDERS_FROZEN) == 1
        c.graph_destroy(g)
        assert c.get_option(R.STAT_ORDERS_FROZEN) == 1
        torch.cuda.synchronize()
        c.graph_destroy(g2)
        assert c.get_option(R.STAT_ORDERS_FROZEN) == 0
        for _ in range(130):
            c.render_rows(p, O.RGB_ASCII, 0, H, d_out=buf.data_ptr(), out_row_base=0, stream=st.cuda_stream)
        torch.cuda.synchronize()
        assert c.get_option(R.STAT_ORDER_PASSES) > passes and O.fnv1a64(buf.cpu().numpy()) == gold
    finally:
        c.close()


def test_two_orders_alternating_on_one_stream_each_converge(R):
    """Order buffers are kept per (stream, tile grid): two slabs of a frame alternating on one stream each keep their own set
    (before round 3 every launch reset the stream's only set, queued a pass and made the next launch wait for it)."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    c = R.Context(W, H)
    b = R.Context(W, H)
    try:
        for x in (c, b):
            x.set_scene(sph, pl)
        b.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
        st = torch.cuda.Stream()
        buf = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        want = torch.empty_like(buf)
        b.render_rows(p, O.RGB_ASCII, 0, H, d_out=want.data_ptr(), out_row_base=0)
        b.synchronize()
        for i in range(120):
            c.render_rows(p, O.RGB_ASCII, 0, 540, d_out=buf.data_ptr(), out_row_base=0, stream=st.cuda_stream)
            c.render_rows(p, O.RGB_ASCII, 540, 540, d_out=buf.data_ptr(), out_row_base=0, stream=st.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(buf, want)
        passes = c.get_option(R.STAT_ORDER_PASSES)
        assert 20 <= passes <= 120     # both grids went through their settling passes; not one pass per launch
    finally:
        c.close()
        b.close()


def test_cell_capacity_grows_with_the_longest_list_instead_of_staging_the_whole_scene(R):
    """A scene packed into a corner of the view: its cells need far more than the default capacity (4 ns / cells + 1024).
    The binning pass reports the longest list it needed, the next launches plan with 1.5 x that, and from then on no
    workgroup falls back to the whole scene; every frame on the way equals the brute kernel's."""
    import torch
    W, H, n = 640, 360, 20000
    p = R.camera_params(W, H)
    rng = np.random.default_rng(17)
    # all spheres within a few degrees of one direction, 60..160 units away
    d = rng.uniform(60, 160, n)
    tx = 0.6 * p.element1 + rng.normal(0, 0.15, n)
    ty = -0.3 * p.element2 + rng.normal(0, 0.05, n)
    c = 1.0 / np.sqrt(1.0 + tx * tx + ty * ty)
    sph = np.stack([d * c * tx, d * c * ty, d * c, rng.uniform(0.05, 0.3, n), np.floor(rng.uniform(1, 256, n)), np.floor(rng.uniform(1, 256, n)),
                    np.floor(rng.uniform(1, 256, n))], axis=1).astype(np.float32)
    pl = np.zeros((0, 11), dtype=np.float32)
    a, b = _pair(R, W, H, sph, pl)
    try:
        got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        want = torch.empty_like(got)
        _frame(b, p, O.RGB_ASCII, want)
        assert a.get_option(R.STAT_CELL_CAPACITY_FLOOR) == 0
        for f in range(8):
            _frame(a, p, O.RGB_ASCII, got)
            torch.cuda.synchronize()
            assert torch.equal(got, want), "frame %d" % f
        floor = a.get_option(R.STAT_CELL_CAPACITY_FLOOR)
        assert floor > 4 * n // 256 + 1024, floor      # grew past the default
        # timing: with the grown lists a frame costs a fraction of what the whole-scene fallback cost
        a.synchronize()
        a.timer_start()
        for _ in range(20):
            a.render_rows(p, O.RGB_ASCII, 0, H, d_out=got.data_ptr(), out_row_base=0)
        grown = a.timer_stop() / 20
        a.set_option(R.OPT_CELL_CAPACITY, 256)          # an explicit small capacity is taken as is: the fallback path
        for _ in range(3):
            _frame(a, p, O.RGB_ASCII, got)
            torch.cuda.synchronize()
            assert torch.equal(got, want)
        a.timer_start()
        for _ in range(20):
            a.render_rows(p, O.RGB_ASCII, 0, H, d_out=got.data_ptr(), out_row_base=0)
        small = a.timer_stop() / 20
        assert grown < small, (grown, small)
    finally:
        a.close()
        b.close()


def test_plan_follows_a_view_that_turns_locally_dense(R):
    """Config 2's scene is sparse by its numbers, but seen along its long axis one macro tile holds 95 candidates instead of 9.
    The trace workgroups report their longest candidate list; the library then plans the following launches as for a dense
    scene (2 sub-tiles per workgroup, two-level culling, per-wave refinement) and returns to the sparse plan when the view
    does.  Whatever the plan: every frame equals the brute kernel's."""
    import torch
    p0, sph, pl = R.config_inputs("C2")
    W, H = int(p0.x), int(p0.y)
    a, b = _pair(R, W, H, sph, pl)
    try:
        assert a.get_option(R.OPT_VIEW_ADAPT) == -1
        got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        want = torch.empty_like(got)
        yaws = [0.0] * 30 + [0.05 * k for k in range(1, 29)] + [1.4] * 50 + [1.4 - 0.05 * k for k in range(1, 29)] + [0.0] * 70
        dense_at, kernels = [], set()
        for f, dy in enumerate(yaws):
            p = R.camera_params(W, H, pos=(0.0, 0.0, 0.0), rot=(0.0, math.pi + dy, 0.0))
            _frame(a, p, O.RGB_ASCII, got)
            torch.cuda.synchronize()
            kernels.add(a.last_kernel)
            dense_at.append(a.get_option(R.STAT_VIEW_DENSE))
            if f % 3 == 0 or dense_at[-1] != (dense_at[-2] if f else 0):
                _frame(b, p, O.RGB_ASCII, want)
                torch.cuda.synchronize()
                assert torch.equal(got, want), "frame %d (yaw +%.2f, dense plan %d)" % (f, dy, dense_at[-1])
        assert not any(dense_at[:30])                       # the default view: sparse plan
        assert all(dense_at[85:105])                        # looking along the scene: dense plan ...
        assert any("refine" in k for k in kernels)          # ... with the per-wave refinement kernels
        assert not any(dense_at[-20:])                      # and back
        assert a.get_option(R.STAT_DENSITY_SWITCHES) >= 2
        # switched off: the sparse plan throughout
        a.set_option(R.OPT_VIEW_ADAPT, 0)
        p = R.camera_params(W, H, pos=(0.0, 0.0, 0.0), rot=(0.0, math.pi + 1.4, 0.0))
        for _ in range(40):
            _frame(a, p, O.RGB_ASCII, got)
        torch.cuda.synchronize()
        assert a.get_option(R.STAT_VIEW_DENSE) == 0 and "refine" not in a.last_kernel
        _frame(b, p, O.RGB_ASCII, want)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    finally:
        a.close()
        b.close()


def test_exact_ties_are_broken_by_creation_order_with_sorted_arrays(R):
    """The kernels know a sphere by its position in the direction-sorted copies (RTX_OPT_SORTED_STORE); the reference keeps the FIRST of
    two objects at the same distance (strict '<' in creation order, RayTracing.cu:123).  Coincident spheres of different colours --
    created in both orders relative to their sort position -- and a plane through sphere surfaces: the frame must equal the oracle's,
    with and without the sorted copies."""
    import torch
    W, H = 320, 180
    p = R.camera_params(W, H)
    rng = np.random.default_rng(23)
    base, _ = R.synth_scene(31, 300, 0, p.element1, p.element2)
    twins = base[rng.choice(300, 60, replace=False)].copy()
    twins[:, 4:7] = np.floor(rng.uniform(1, 256, (60, 3)))          # same centre and radius, another colour: every hit is an exact tie
    more, _ = R.synth_scene(37, 200, 0, p.element1, p.element2)
    sph = np.concatenate([twins[:30], base, twins[30:], more, base[:10]]).astype(np.float32)   # twins created before and after their originals
    pl = np.zeros((0, 11), dtype=np.float32)
    sc = O.Scene.from_arrays(sph, pl)
    want = O.render(U.oracle_params(p), sc, O.RGB_ASCII, threads=8)
    want_bit = O.render(U.oracle_params(p), sc, O.BIT_ASCII, threads=8)
    c = R.Context(W, H)
    try:
        c.set_scene(sph, pl)
        for srt in (-1, 0, 1):
            c.set_option(R.OPT_SORTED_STORE, srt)
            for kernel in (R.KERNEL_AUTO, R.KERNEL_BRUTE):
                c.set_option(R.OPT_KERNEL, kernel)
                for two in (-1, 1):
                    c.set_option(R.OPT_TWO_LEVEL, two)
                    got = c.render_to_host(p, O.RGB_ASCII)
                    assert np.array_equal(got, want), (srt, kernel, two, U.first_diff(got, want, 20, W))
            got = c.render_to_host(p, O.BIT_ASCII)
            assert np.array_equal(got, want_bit), srt
    finally:
        c.close()


def test_grids_of_several_rounds_go_heaviest_first_only_on_a_lone_stream(R):
    """RTX_OPT_TILE_ORDER auto on a tile grid of several dispatch rounds (a dense 1080p scene: 4050 workgroups for 1792 slots):
    while every launch comes on one stream the library derives heaviest-first orders (rtx_order_tiles, every 16th frame) --
    nothing overlaps a launch's tail there -- and stops when frames arrive on several streams.  The frame is the brute kernel's
    under every order."""
    import torch
    W, H, n = 1920, 1080, 16384
    sph, pl = _dense_scene(R, W, H, n, 21)
    a, b = _pair(R, W, H, sph, pl)
    try:
        p = R.camera_params(W, H)
        got = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        want = torch.empty_like(got)
        _frame(b, p, O.RGB_ASCII, want)
        b.synchronize()
        passes0 = a.get_option(R.STAT_ORDER_PASSES)
        for f in range(40):
            _frame(a, p, O.RGB_ASCII, got)
            a.synchronize()
            if f in (0, 1, 2, 17, 39):
                assert torch.equal(got, want), "frame %d on a lone stream" % f
        assert a.last_kernel.endswith(",refine>")
        lone = a.get_option(R.STAT_ORDER_PASSES) - passes0
        assert lone >= 3, "no dispatch order derived on a lone stream: %d passes" % lone
        # two streams, alternating: the order is no longer derived (the static XCD order serves)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for f in range(20):
            st = (s1, s2)[f % 2]
            _frame(a, p, O.RGB_ASCII, got, stream=st.cuda_stream)
            st.synchronize()
            if f in (0, 19):
                assert torch.equal(got, want), "frame %d on two streams" % f
        before = a.get_option(R.STAT_ORDER_PASSES)
        for f in range(20):
            st = (s1, s2)[f % 2]
            _frame(a, p, O.RGB_ASCII, got, stream=st.cuda_stream)
            st.synchronize()
        assert torch.equal(got, want)
        assert a.get_option(R.STAT_ORDER_PASSES) == before, "orders still derived with frames on two streams"
    finally:
        a.close()
        b.close()

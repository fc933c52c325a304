"""Batched slab launches (RTX_OPT_BATCH, rtx_trace_batch): rtx_submit_slabs traces consecutive slabs of one stream -- a rank's
rows of every frame of a round in the row-sharded loop -- with ONE launch, the frames' cameras and output
buffers travelling in the kernel arguments.  The reference renders one whole frame per launch on one device
(RayTracingManager.cu:122-135); what must hold is that every frame of a batch is, byte for byte, the slab a launch of its own
renders -- and through it the oracle's."""
import numpy as np
import pytest

import oracle as O
import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    return U.pkg()


def _cams(R, W, H, n):
    return [R.camera_params(W, H, pos=(0.4 * i, -0.2 * i, 0.1 * i), rot=(0.01 * i, float(np.float32(np.pi)) + 0.02 * i, 0.003 * i)) for i in range(n)]


@pytest.mark.parametrize("n,rank,ranks", [(8, 3, 8), (2, 0, 2), (16, 7, 8), (5, 1, 3)])
@pytest.mark.parametrize("compact", [False, True])
def test_batched_slabs_equal_slabs_launched_one_by_one(R, n, rank, ranks, compact):
    import torch
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    row0, rows = H * rank // ranks, H * (rank + 1) // ranks - H * rank // ranks
    cams = _cams(R, W, H, n)
    S = 4 if compact else 20
    flags = R.RENDER_COMPACT if compact else 0
    with R.Context(W, H) as c:
        c.set_scene(sph, pl)
        st = torch.cuda.Stream()
        got = [torch.empty(rows * W * S, dtype=torch.uint8, device="cuda") for _ in range(n)]
        want = [torch.empty(rows * W * S, dtype=torch.uint8, device="cuda") for _ in range(n)]
        for mode in (R.RGB_ASCII, R.BIT_ASCII, R.RGB_NORMALS):
            for b in got + want:
                b.fill_(0xEE)
            torch.cuda.synchronize()
            before = c.get_option(R.STAT_BATCHED_LAUNCHES)
            c.submit_slabs(cams, mode, row0, rows, [b.data_ptr() for b in got], row0, [st.cuda_stream] * n, flags=flags)
            assert c.get_option(R.STAT_BATCHED_LAUNCHES) == before + 1 and "rtx_trace_batch" in c.last_kernel
            c.set_option(R.OPT_BATCH, 0)
            c.submit_slabs(cams, mode, row0, rows, [b.data_ptr() for b in want], row0, [st.cuda_stream] * n, flags=flags)
            c.set_option(R.OPT_BATCH, -1)
            assert c.get_option(R.STAT_BATCHED_LAUNCHES) == before + 1 and "rtx_trace<" in c.last_kernel
            torch.cuda.synchronize()
            Sm = S if compact else (20 if mode >= R.RGB_ASCII else 12)
            for i in range(n):
                assert torch.equal(got[i][:rows * W * Sm], want[i][:rows * W * Sm]), (R.MODE_NAMES[mode], i)


def test_batched_slabs_against_the_oracle_and_over_many_launches(R):
    """Small frame, the reference's default scene + synthetic spheres so that the culling kernel is in use: each frame of a batch
    against the CPU oracle's frame of that camera; repeated, so that the batch's dispatch order (heaviest first, from the
    estimates one frame of the batch leaves) comes into use and changes nothing."""
    import torch
    W, H = 400, 150
    p0 = R.camera_params(W, H)
    sph, pl = R.synth_scene(3, 150, 1, p0.element1, p0.element2)   # (sparse by its numbers: 0.0025 spheres per pixel)
    sc = O.Scene.from_arrays(sph, pl)
    cams = _cams(R, W, H, 6)
    with R.Context(W, H) as c:
        c.set_scene(sph, pl)
        c.set_option(R.OPT_KERNEL, R.KERNEL_BINNED)
        st = torch.cuda.Stream()
        bufs = [torch.empty(20 * W * H, dtype=torch.uint8, device="cuda") for _ in cams]
        want = [O.render(U.oracle_params(q), sc, O.RGB_ASCII) for q in cams]
        for it in range(40):
            if it in (0, 39):
                for b in bufs:
                    b.fill_(0xEE)
                torch.cuda.synchronize()
            c.submit_slabs(cams, R.RGB_ASCII, 0, H, [b.data_ptr() for b in bufs], 0, [st.cuda_stream] * len(cams))
            if it in (0, 39):
                torch.cuda.synchronize()
                for i, b in enumerate(bufs):
                    got = b.cpu().numpy()
                    assert np.array_equal(got, want[i]), (it, i, U.first_diff(got, want[i], 20, W))
        assert c.get_option(R.STAT_BATCHED_LAUNCHES) == 40


def test_mixed_streams_split_into_runs_and_a_graph_replays_the_batch(R):
    """Slabs of one call on different streams: each run of equal streams is its own launch (a run of one: a plain launch); 17
    slabs on one stream: 16 + 1.  A batched launch is recordable as it is (nothing is uploaded for it): three replays of a
    recorded round equal the direct call."""
    import torch
    p, sph, pl = R.config_inputs("C2")
    W, H = int(p.x), int(p.y)
    row0, rows = 405, 135
    with R.Context(W, H) as c:
        c.set_scene(sph, pl)
        s0, s1, main = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        cams = _cams(R, W, H, 17)
        bufs = [torch.empty(rows * W * 4, dtype=torch.uint8, device="cuda") for _ in cams]
        ref = [torch.empty(rows * W * 4, dtype=torch.uint8, device="cuda") for _ in cams]
        c.set_option(R.OPT_BATCH, 0)
        c.submit_slabs(cams, R.RGB_ASCII, row0, rows, [b.data_ptr() for b in ref], row0, [s0.cuda_stream] * 17, flags=R.RENDER_COMPACT)
        c.set_option(R.OPT_BATCH, -1)
        torch.cuda.synchronize()
        k0 = c.get_option(R.STAT_BATCHED_LAUNCHES)
        # streams: s0 s0 s0 s1 s0 s1 s1 ... -> runs of 3, 1, 1, 2, then the rest on s0
        streams = [s0, s0, s0, s1, s0, s1, s1] + [s0] * 10
        c.submit_slabs(cams, R.RGB_ASCII, row0, rows, [b.data_ptr() for b in bufs], row0, [s.cuda_stream for s in streams],
                       after=main.cuda_stream, flags=R.RENDER_COMPACT)
        main.synchronize()     # the join: everything the call queued is done when `after` is
        assert c.get_option(R.STAT_BATCHED_LAUNCHES) == k0 + 3          # runs of 3, 2 and 10
        for i in range(17):
            assert torch.equal(bufs[i], ref[i]), i
        c.submit_slabs(cams, R.RGB_ASCII, row0, rows, [b.data_ptr() for b in bufs], row0, [s0.cuda_stream] * 17, flags=R.RENDER_COMPACT)
        assert c.get_option(R.STAT_BATCHED_LAUNCHES) == k0 + 4          # 16 in one batched launch + 1 plain launch
        torch.cuda.synchronize()
        # a recorded round
        c.graph_begin(main.cuda_stream)
        c.submit_slabs(cams[:8], R.RGB_ASCII, row0, rows, [b.data_ptr() for b in bufs[:8]], row0, [s0.cuda_stream] * 8,
                       after=main.cuda_stream, flags=R.RENDER_COMPACT)
        g = c.graph_end(main.cuda_stream)
        for rep in range(3):
            for b in bufs[:8]:
                b.fill_(0xEE)
            torch.cuda.synchronize()
            c.graph_launch(g, main.cuda_stream)
            main.synchronize()
            for i in range(8):
                assert torch.equal(bufs[i], ref[i]), (rep, i)
        c.graph_destroy(g)


def test_plans_the_batched_kernel_does_not_take_fall_back_to_one_launch_per_slab(R):
    """Dense scenes (two-level culling, per-wave refinement) and the brute kernel keep a launch per slab; the frames are the same."""
    import torch
    W, H = 640, 360
    p = R.camera_params(W, H)
    sph, _ = R.synth_scene(5, 8192, 0, p.element1, p.element2)
    pl = np.zeros((0, 11), dtype=np.float32)
    cams = _cams(R, W, H, 4)
    with R.Context(W, H) as c:
        c.set_scene(sph, pl)
        st = torch.cuda.Stream()
        bufs = [torch.empty(20 * W * H, dtype=torch.uint8, device="cuda") for _ in cams]
        c.submit_slabs(cams, R.RGB_ASCII, 0, H, [b.data_ptr() for b in bufs], 0, [st.cuda_stream] * 4)
        torch.cuda.synchronize()
        assert c.get_option(R.STAT_BATCHED_LAUNCHES) == 0
        dense = [b.clone() for b in bufs]
        c.set_option(R.OPT_KERNEL, R.KERNEL_BRUTE)
        c.submit_slabs(cams, R.RGB_ASCII, 0, H, [b.data_ptr() for b in bufs], 0, [st.cuda_stream] * 4)
        torch.cuda.synchronize()
        assert c.get_option(R.STAT_BATCHED_LAUNCHES) == 0
        for i in range(4):
            assert torch.equal(bufs[i], dense[i]), i

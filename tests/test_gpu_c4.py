"""BASELINE config 4 (7680x4320, 1024 spheres) on ONE GPU, and the device xterm-256 mapper over all 2^24 inputs.

C4 is the row-sharded 8-GPU configuration (8 slabs of 540 rows, 82 944 000 B each).  One GPU holds
the whole 663 552 000-byte frame, so the test renders it (a) as one launch and (b) as the eight slabs each rank
would render -- global row index in ray generation, slab-local destination -- both as records and as compact
pixel words expanded on the "root", and requires the committed golden hashes (RayTracingManager.cu:122-125 is
the launch shape this replaces; tests/golden/golden.json is oracle-generated).
"""
import hashlib

import numpy as np
import pytest

import oracle as O
import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    return U.pkg()


@pytest.fixture(scope="module")
def c4(R):
    import torch
    W, H, ns, npl, seed = R.CONFIGS["C4"]
    ctx = R.Context(W, H)
    p, sph, pl = R.config_inputs("C4")
    ctx.set_scene(sph, pl)
    yield ctx, p, W, H
    ctx.close()
    torch.cuda.empty_cache()


def _check(frame, what):
    gold = U.load_golden()["C4_RGB_ASCII"]
    assert frame.size == 20 * 7680 * 4320
    assert O.fnv1a64(frame) == gold["frame_fnv1a64"], what
    assert hashlib.sha256(frame.tobytes()).hexdigest() == gold["frame_sha256"], what


def test_c4_whole_frame_one_launch(R, c4):
    ctx, p, W, H = c4
    got = ctx.render_to_host(p, R.RGB_ASCII)
    assert ctx.last_kernel.startswith("rtx_trace<")
    _check(got, "C4 whole frame")
    rec = got.reshape(H, W, 20)
    assert int((rec[:, :W - 1, 2] == ord("3")).sum()) == U.load_golden()["C4_RGB_ASCII"]["foreground_pixels"]
    assert not rec[:, W - 1, :].any()          # column W-1 stays NUL (RayTracing.cu:187)


@pytest.mark.parametrize("form", ["records", "compact"])
def test_c4_as_eight_slabs_of_540_rows(R, c4, form):
    """What the 8 ranks of SURVEY 8(e) do, one after the other on this GPU: rank g traces rows [540g, 540g+540)
    into a slab-local buffer; the slabs in rank order are the frame (records), or become it through rtx_expand."""
    import torch
    ctx, p, W, H = c4
    parts = 8
    bounds = [H * g // parts for g in range(parts + 1)]
    assert [b - a for a, b in zip(bounds[:-1], bounds[1:])] == [540] * 8
    if form == "records":
        pieces = []
        for g in range(parts):
            rows = bounds[g + 1] - bounds[g]
            slab = torch.empty(20 * W * rows, dtype=torch.uint8, device="cuda")
            assert slab.numel() == 82_944_000
            slab.fill_(0xEE)                   # the kernel must write every byte of the slab (NUL column included)
            torch.cuda.synchronize()
            ctx.render_rows(p, R.RGB_ASCII, bounds[g], rows, d_out=slab.data_ptr(), out_row_base=bounds[g])
            ctx.synchronize()
            pieces.append(slab.cpu().numpy())
            del slab
        _check(np.concatenate(pieces), "C4 as 8 record slabs")
    else:
        words = torch.empty(W * H, dtype=torch.int32, device="cuda")
        frame = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")
        frame.fill_(0xEE)
        torch.cuda.synchronize()
        segs = []
        for g in range(parts):
            rows = bounds[g + 1] - bounds[g]
            ctx.render_rows(p, R.RGB_ASCII, bounds[g], rows, d_out=words.data_ptr() + 4 * W * bounds[g], out_row_base=bounds[g],
                            flags=R.RENDER_COMPACT)
            segs.append((W * bounds[g], W * bounds[g], W * rows))
        ctx.expand(R.RGB_ASCII, words.data_ptr(), frame.data_ptr(), segs)
        ctx.synchronize()
        got = frame.cpu().numpy()
        del words, frame
        _check(got, "C4 as 8 compact slabs expanded on the root")


# ---------------------------------------------------------------- ansi256_from_rgb, all 2^24 inputs

SURVEY_ANSI_EXHAUSTIVE = "0c9c9ba3eba54d0e"   # SURVEY.md section 4 item 1 (FNV-1a-64, the survey's offset basis)


def test_device_ansi256_mapper_on_all_inputs(R):
    """The kernel-side mapper (rtx_device.hpp: palette computed on the fly + the context's rule-generated grey
    lookup) is a different implementation from the oracle's table form (ANSIRGB.h:141-189 restated); frames only
    visit the colours that happen to occur.  rtx_ansi256_map runs the device function over every 0xRRGGBB."""
    import torch
    n = 1 << 24
    with R.Context(64, 64) as ctx:
        out = torch.full((n + 8,), 0xEE, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.ansi256_map(0, n, out.data_ptr())
        # an unaligned destination and a ragged range take the byte-store path
        part = torch.full((1003 + 8,), 0xEE, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.ansi256_map(0x7f7e81, 1003, part.data_ptr() + 1)
        ctx.synchronize()
        got = out.cpu().numpy()
        gpart = part.cpu().numpy()
        with pytest.raises(R.RtxError):
            ctx.ansi256_map(n - 2, 3, out.data_ptr())
    assert (got[n:] == 0xEE).all()
    got = got[:n]
    want = O.ansi256_table()
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, "device mapper differs on %d colours, first 0x%06x: got %d want %d" % (
        bad.size, bad[0], got[bad[0]], want[bad[0]])
    assert int(got.min()) == 16 and int(got.max()) == 255
    assert O.fnv1a64(got, O.FNV_OFFSET_SURVEY) == SURVEY_ANSI_EXHAUSTIVE
    assert gpart[0] == 0xEE and (gpart[1004:] == 0xEE).all()
    assert np.array_equal(gpart[1:1004], want[0x7f7e81:0x7f7e81 + 1003])

"""Pins the CPU oracle to the known answers SURVEY.md 8(c) recorded from the reference's own sources.

The reference ships no tests or golden vectors and cannot be built in this image (pch.h needs
<windows.h> and <cuda_runtime.h>), so these eleven known answers are the pin: ten hashes of the
full zero-initialised 20*W*H buffer (reference default scene Scene3D.cpp:28-33, default camera,
five live modes, 400x150 and 1920x1080) and the exhaustive 2^24-input hash of ansi256_from_rgb.
They were recorded with host libm powf and x86 float->uint8 conversion; the oracle reproduces
them in that configuration AND with the pinned pow32 routine the HIP kernel uses, so the pin
carries over to the configuration the GPU parity tests compare against.
"""
import numpy as np
import pytest

import oracle as O

SURVEY_8C = {
    (400, 150): {
        O.BIT_ASCII: "566f369b48c48349",
        O.BIT_PIXEL: "2600c441a058a41f",
        O.RGB_ASCII: "dd3497ccdb38ff6e",
        O.RGB_PIXEL: "08bda1486917bf70",
        O.RGB_NORMALS: "998ca661d8b23494",
    },
    (1920, 1080): {
        O.BIT_ASCII: "b366f64565c06fa1",
        O.BIT_PIXEL: "454b2ee3b30179c4",
        O.RGB_ASCII: "71e4385fd8fe0844",
        O.RGB_PIXEL: "0cef41476e6725c5",
        O.RGB_NORMALS: "874179fbed44ad04",
    },
}
SURVEY_ANSI_EXHAUSTIVE = "0c9c9ba3eba54d0e"


def test_ansi256_exhaustive_hash_matches_survey():
    assert "%016x" % O.lib().orc_ansi256_exhaustive_hash(O.FNV_OFFSET_SURVEY) == SURVEY_ANSI_EXHAUSTIVE


def test_ansi256_range_and_greys():
    got = [O.lib().orc_ansi256_from_rgb((v << 16) | (v << 8) | v) for v in range(256)]
    assert min(got) == 16 and max(got) == 255          # "min 16, max 255"
    assert got[0] == 16 and got[255] == 231 and got[95] == 59 and got[8] == 232 and got[238] == 255


def test_default_camera_params_match_survey():
    # element1 = 0.866025388 (400x150) / 6.2354 (1080p), element2 = 0.577350259, far 250,
    # invV = [[-1,0,8.74228e-08,0],[0,1,0,0],[8.74228e-08,0,1,0],[0,0,0,1]]
    p = O.camera_params(400, 150)
    assert np.float32(p.element1) == np.float32(0.866025388)
    assert np.float32(p.element2) == np.float32(0.577350259)
    assert p.cam_far == 250.0
    m = np.array([[p.inv_v[i][j] for j in range(4)] for i in range(4)], dtype=np.float32)
    want = np.array([[-1, 0, 8.74228e-08, 0], [0, 1, 0, 0], [8.74228e-08, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    assert np.allclose(m, want, rtol=1e-6, atol=0)
    assert abs(O.camera_params(1920, 1080).element1 - 6.2354) < 1e-4


@pytest.mark.parametrize("res", sorted(SURVEY_8C))
@pytest.mark.parametrize("mode", [O.BIT_ASCII, O.BIT_PIXEL, O.RGB_ASCII, O.RGB_PIXEL, O.RGB_NORMALS])
def test_default_scene_buffer_hash_matches_survey(res, mode):
    w, h = res
    p = O.camera_params(w, h)
    sc = O.Scene.reference_default()
    want = SURVEY_8C[res][mode]
    # as recorded: libm powf, x86 conversion of negative normals
    buf = O.render(p, sc, mode, flags=O.POW_LIBM | O.NORMALS_WRAP, threads=4)
    assert O.fnv1a64(buf, O.FNV_OFFSET_SURVEY) == want
    # the pinned pow32 (what the GPU path uses) gives the same bytes on these frames
    buf2 = O.render(p, sc, mode, flags=O.NORMALS_WRAP, threads=4)
    assert O.fnv1a64(buf2, O.FNV_OFFSET_SURVEY) == want
    if mode != O.RGB_NORMALS:
        # the saturating conversion only matters for negative normals
        buf3 = O.render(p, sc, mode, flags=0, threads=4)
        assert np.array_equal(buf3, buf2)


def test_normals_saturate_differs_from_wrap_only_in_negative_components():
    w, h = 400, 150
    p = O.camera_params(w, h)
    sc = O.Scene.reference_default()
    sat, px = O.render(p, sc, O.RGB_NORMALS, flags=0, want_pixels=True)
    wrap = O.render(p, sc, O.RGB_NORMALS, flags=O.NORMALS_WRAP)
    sat = sat.reshape(h, w, 20)
    wrap = wrap.reshape(h, w, 20)
    differs = (sat != wrap).any(axis=2)
    neg = (px["normal"] * np.float32(255) <= -1).any(axis=2) & (px["distance"] <= 250.0)
    neg[:, -1] = False
    assert differs.any()
    assert np.array_equal(differs, neg)


def test_digit_encoder_is_plain_decimal_with_nul_padding():
    # "digit encoder over 0..255 equals plain decimal digits with NUL for absent leading digits"
    # exercised through RGB_PIXEL records of a flat-lit scene is indirect; check the records in a real frame instead
    p = O.camera_params(400, 150)
    buf, px = O.render(p, O.Scene.reference_default(), O.RGB_PIXEL, want_pixels=True)
    rec = buf.reshape(150, 400, 20)
    vis = px["distance"] <= 250.0
    vis[:, -1] = False  # column W-1 is never traced (RayTracing.cu:187)
    for ch, off in ((0, 7), (1, 11), (2, 15)):
        val = np.clip(np.trunc(px["color"][..., ch]), 0, 255).astype(int)[vis]
        d = rec[vis][:, off:off + 3]
        want = np.stack([np.where(val >= 100, val // 100 + 48, 0), np.where(val >= 10, (val // 10) % 10 + 48, 0),
                         val % 10 + 48], axis=1)
        assert np.array_equal(d, want)


def test_threaded_and_single_row_forms_equal_the_plain_loop():
    """orc_render_mt (rows handed out in 4-row blocks from a shared counter: what bench.py's cpu_baseline times) and
    orc_render_row (one row into a buffer of its own: what the GPU fuzz tests compare sampled rows with) are the same
    per-pixel loop as orc_render_rows, so the bytes are the same whatever the schedule."""
    sc = O.Scene.reference_default()
    p = O.camera_params(400, 150)
    for mode in (O.BIT_ASCII, O.RGB_ASCII, O.RGB_NORMALS):
        S = 20 if mode >= O.RGB_ASCII else 12
        full = O.render(p, sc, mode)
        for threads in (2, 7, 64):
            assert np.array_equal(O.render(p, sc, mode, threads=threads), full)
        for r in (0, 1, 77, 149):
            assert np.array_equal(O.render_row(p, sc, mode, r), full[r * 400 * S:(r + 1) * 400 * S])
    with pytest.raises(ValueError):
        O.render(p, sc, 9, threads=4)

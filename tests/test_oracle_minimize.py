"""The oracle's Minimize (RayTracingManager.cu:167-319) against a second, array-level statement of
the same rule, on real frames and on edge cases."""
import numpy as np
import pytest

import oracle as O


def minimize_by_rule(mode, buf, W, H):
    """Per pixel: whole record if its colour digits differ from the previous pixel's (scan order,
    across rows), else just the glyph; '\\n' at every row end; empty slots emit nothing."""
    S = 20 if mode >= O.RGB_ASCII else 12
    rec = np.asarray(buf, dtype=np.uint8)[:S * W * H].reshape(H, W, S)
    cols = [7, 8, 9] if S == 12 else [7, 8, 9, 11, 12, 13, 15, 16, 17]
    out = bytearray()
    last = None
    for r in range(H):
        for c in range(W):
            slot = rec[r, c]
            if slot[0] == 0x1b:
                colour = bytes(slot[cols])
                if last is None or colour != last:
                    out += bytes(slot)
                    last = colour
                else:
                    out.append(int(slot[S - 1]))
        out.append(0x0a)
    return np.frombuffer(bytes(out), dtype=np.uint8)


@pytest.mark.parametrize("mode", range(5))
def test_minimize_default_scene(mode):
    W, H = 200, 75
    p = O.camera_params(W, H)
    buf = O.render(p, O.Scene.reference_default(), mode)
    got = O.minimize(mode, buf, W, H)
    want = minimize_by_rule(mode, buf, W, H)
    assert np.array_equal(got, want)
    assert got[-1] == 0x0a and (got == 0x0a).sum() == H


def test_minimize_sdl_frame_is_only_newlines():
    W, H = 40, 10
    buf = np.zeros(20 * W * H, dtype=np.uint8)
    got = O.minimize(O.SDL, buf, W, H)
    assert bytes(got) == b"\n" * H


def test_minimize_width_one():
    W, H = 1, 7
    p = O.camera_params(W, H)
    buf = O.render(p, O.Scene.reference_default(), O.RGB_ASCII)
    assert not buf.any()
    assert bytes(O.minimize(O.RGB_ASCII, buf, W, H)) == b"\n" * H


def test_minimize_colour_persists_across_rows_and_ignores_fg_bg_kind():
    # all-miss frame: first pixel whole, every other pixel only its glyph
    W, H = 16, 4
    p = O.camera_params(W, H)
    sc = O.Scene()
    buf = O.render(p, sc, O.RGB_PIXEL)
    got = O.minimize(O.RGB_PIXEL, buf, W, H)
    assert got.size == 20 + ((W - 1) * H - 1) + H

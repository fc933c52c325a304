"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/rtx.h declares, host builders agree with the oracle bit for bit, and the library fails
loudly (never falls back to a CPU path) when no GPU is present.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as O
import util as U


def header_functions():
    src = open(os.path.join(U.ROOT, "include", "rtx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    R = U.pkg()
    R.build()
    lib = C.CDLL(R.LIB_PATH)
    names = header_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), "librtx_hip.so does not export %s" % name
    # and the ctypes table covers exactly the header
    assert sorted(R.EXPORTED_SYMBOLS) == names


def test_params_struct_layout():
    R = U.pkg()
    assert C.sizeof(R.Params) == 16 * 4 + 3 * 4 + 3 * 4 + 2 * 8
    assert R.Params.x.offset == 88 and R.Params.y.offset == 96


def test_version_and_error_strings():
    R = U.pkg()
    assert b"gfx950" in R.lib().rtx_version()


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    R = U.pkg()
    with pytest.raises(R.RtxError) as e:
        R.Context(64, 64)
    assert e.value.status == R.ERR_NO_DEVICE


def test_device_group_without_a_gpu_fails_the_same_way_and_validates_its_arguments():
    """rtx_group_create (the row-sharded frame behind the ABI): no device -> RTX_ERR_NO_DEVICE, never a CPU path; a device
    count outside [1, 64] is refused before anything is created; librccl is not mapped by loading the library (it is
    opened on the first exchange that wants it)."""
    import ctypes as C
    import torch
    R = U.pkg()
    h = C.c_void_p()
    assert R.lib().rtx_group_create(0, None, 64, 64, C.byref(h)) == R.ERR_INVALID_ARGUMENT and not h.value
    assert R.lib().rtx_group_create(65, None, 64, 64, C.byref(h)) == R.ERR_INVALID_ARGUMENT and not h.value
    assert b"ndev" in R.lib().rtx_last_error(None)
    assert R.lib().rtx_group_size(None) == 0
    import subprocess
    assert "rccl" not in subprocess.check_output(["ldd", R.LIB_PATH]).decode()
    if not torch.cuda.is_available():
        with pytest.raises(R.RtxError) as e:
            R.Context(64, 64, devices=[0, 0])
        assert e.value.status == R.ERR_NO_DEVICE


def test_product_never_links_the_oracle():
    R = U.pkg()
    import subprocess
    out = subprocess.check_output(["ldd", R.LIB_PATH]).decode()
    assert "oracle" not in out
    syms = subprocess.check_output(["nm", "-D", "--defined-only", R.LIB_PATH]).decode()
    assert "orc_" not in syms
    for root, _, files in os.walk(R.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "rtx_oracle" not in text and "librtx_oracle" not in text, os.path.join(root, f)


@pytest.mark.parametrize("pose", [
    (None, None),
    ((1.5, -2.0, 3.25), (0.3, 2.1, 0.0)),
    ((10.0, 5.0, -7.0), (-1.2, 0.4, 0.0)),
    ((0.0, 0.0, 0.0), (1.57, 3.0, 0.0)),
])
@pytest.mark.parametrize("res", [(400, 150), (320, 180), (1920, 1080), (7680, 4320)])
def test_camera_params_match_oracle_bitwise(pose, res):
    R = U.pkg()
    pos, rot = pose
    w, h = res
    p = R.camera_params(w, h, pos, rot)
    q = O.camera_params(w, h, pos if pos else (0, 0, 0), rot if rot else O.DEFAULT_ROT)
    a = np.array(p.inv_v[:], dtype=np.float32)
    b = np.array([[q.inv_v[i][j] for j in range(4)] for i in range(4)], dtype=np.float32).reshape(16)
    assert a.tobytes() == b.tobytes()
    assert (p.element1, p.element2, p.cam_far, p.x, p.y) == (q.element1, q.element2, q.cam_far, q.x, q.y)


def test_camera_inverse_is_an_inverse():
    R = U.pkg()
    pos, rot = (3.0, -4.0, 5.0), (0.4, 1.1, 0.0)
    p = R.camera_params(400, 150, pos, rot)
    inv = np.array(p.inv_v[:], dtype=np.float64).reshape(4, 4)
    sp, cp, sy, cy = np.sin(rot[0]), np.cos(rot[0]), np.sin(rot[1]), np.cos(rot[1])
    m = np.array([[cy, 0, -sy, pos[0]], [-sp * sy, cp, -sp * cy, pos[1]], [-cp * sy, -sp, -cp * cy, pos[2]], [0, 0, 0, 1]])
    assert np.allclose(inv @ m, np.eye(4), atol=1e-5)


@pytest.mark.parametrize("seed", range(8))
def test_camera_params_against_an_independent_numpy_inverse(seed):
    """rtx_camera_params and orc_camera_params carry the same sixteen cofactor lines (the reference's own term
    order, Camera3D.cpp:207-376), so comparing them is twin against twin.  Here the matrix of Camera3D::Update
    (Camera3D.cpp:51-98: rows (right.i, up.i, forward.i, pos.i)) is rebuilt in float64 from the fp32 sines and
    cosines, inverted by numpy.linalg.inv (LU, no shared code), and rtx_camera_params must agree to fp32
    rounding over random poses; the projection scalars of Camera3D::Init (:8-48) likewise."""
    R = U.pkg()
    rng = np.random.default_rng(4242 + seed)
    for _ in range(40):
        pos = [float(np.float32(v)) for v in rng.uniform(-500, 500, 3) * rng.choice([0.01, 1.0, 10.0])]
        rot = [float(np.float32(rng.uniform(-1.55, 1.55))), float(np.float32(rng.uniform(-7, 7))), float(np.float32(rng.uniform(-1, 1)))]
        w, h = int(rng.integers(1, 8000)), int(rng.integers(1, 4500))
        p = R.camera_params(w, h, pos, rot)
        f = np.float32
        sp, cp, sy, cy = (np.float64(np.sin(f(rot[0]))), np.float64(np.cos(f(rot[0]))),
                          np.float64(np.sin(f(rot[1]))), np.float64(np.cos(f(rot[1]))))
        # fp32 products as the reference forms them, then everything else in float64
        right = [cy, np.float64(f(-sp) * f(sy)), np.float64(f(-cp) * f(sy))]
        up = [0.0, cp, -sp]
        fwd = [-sy, np.float64(f(-sp) * f(cy)), np.float64(f(-cp) * f(cy))]
        m = np.eye(4)
        for i in range(3):
            m[i] = [right[i], up[i], fwd[i], pos[i]]
        want = np.linalg.inv(m)
        got = np.array(p.inv_v[:], dtype=np.float64).reshape(4, 4)
        # rotation block: entries are O(1); translation column: O(|pos|)
        scale = np.ones((4, 4))
        scale[:3, 3] = max(1.0, float(np.abs(pos).max()))
        assert np.all(np.abs(got - want) <= 4e-6 * scale), (pos, rot, got - want)
        assert np.allclose(got @ m, np.eye(4), atol=4e-6 * scale.max())
        assert np.allclose(m @ got, np.eye(4), atol=4e-6 * scale.max())
        assert list(p.cam_pos[:]) == pos and (p.x, p.y) == (w, h) and p.cam_far == 250.0
        # Camera3D::Init: e = 1/tan(FOV/2), FOV = pi/1.5; aspect = w / (0.01 w h); element1 = e / aspect
        e = 1.0 / np.tan((np.pi / 1.5) / 2.0)
        assert abs(p.element2 - e) <= 2e-7 * e
        assert abs(p.element1 - e * 0.01 * h) <= 1e-6 * e * 0.01 * h


def test_camera_params_default_pose_is_the_surveys():
    """default camera (pos 0, rot (0, pi, 0)): element1 = 0.866025388 at 400x150, element2 =
    0.577350259, far 250, invV = [[-1,0,8.74228e-08,0],[0,1,0,0],[8.74228e-08,0,1,0],[0,0,0,1]] up to zero signs."""
    R = U.pkg()
    p = R.camera_params(400, 150)
    assert np.float32(p.element1) == np.float32(0.866025388) and np.float32(p.element2) == np.float32(0.577350259)
    inv = np.array(p.inv_v[:], dtype=np.float32).reshape(4, 4)
    want = np.array([[-1, 0, 8.74228e-08, 0], [0, 1, 0, 0], [8.74228e-08, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    assert np.allclose(inv, want, rtol=1e-6, atol=1e-12)
    assert np.float32(R.camera_params(1920, 1080).element1) == np.float32(6.2354) or abs(R.camera_params(1920, 1080).element1 - 6.2354) < 1e-4


@pytest.mark.parametrize("name", ["C1", "C2", "C3"])
def test_synth_scene_matches_independent_numpy_generator(name):
    R = U.pkg()
    w, h, ns, npl, seed = R.CONFIGS[name]
    p = R.camera_params(w, h)
    s1, p1 = R.synth_scene(seed, ns, npl, p.element1, p.element2)
    s2, p2 = U.numpy_synth_scene(seed, ns, npl, p.element1, p.element2)
    assert s1.tobytes() == s2.tobytes()
    assert p1.tobytes() == p2.tobytes()


def test_c1_scene_fixture_pins_the_generator():
    R = U.pkg()
    fx = np.load(os.path.join(U.GOLDEN_DIR, "c1_scene.npz"))
    p, sph, pl = R.config_inputs("C1")
    assert sph.tobytes() == fx["spheres"].tobytes()
    assert pl.tobytes() == fx["planes"].tobytes()
    assert np.array(p.inv_v[:], dtype=np.float32).tobytes() == fx["inv_v"].tobytes()
    assert [p.element1, p.element2, p.cam_far] == [float(v) for v in fx["scalars"]]


def test_synth_scene_coverage_matches_survey():
    # "coverage ... validated: C1 66 %, C2 56 %" foreground pixels (reference kernels)
    R = U.pkg()
    gold = U.load_golden()
    for name, want in (("C1", 66), ("C2", 56)):
        w, h = R.CONFIGS[name][:2]
        frac = 100.0 * gold["%s_RGB_ASCII" % name]["foreground_pixels"] / ((w - 1) * h)
        assert round(frac) == want


def test_oracle_reproduces_committed_goldens_small():
    R = U.pkg()
    gold = U.load_golden()
    p, sph, pl = R.config_inputs("C1")
    sc = O.Scene.from_arrays(sph, pl)
    op = U.oracle_params(p)
    for mode in range(5):
        buf = O.render(op, sc, mode)
        g = gold["C1_%s" % O.MODE_NAMES[mode]]
        assert O.fnv1a64(buf) == g["frame_fnv1a64"]
        mini = O.minimize(mode, buf, int(p.x), int(p.y))
        assert (O.fnv1a64(mini), mini.size) == (g["minimized_fnv1a64"], g["minimized_bytes"])
    frame = np.load(os.path.join(U.GOLDEN_DIR, "c1_rgb_ascii_frame.npz"))["frame"]
    assert np.array_equal(frame, O.render(op, sc, O.RGB_ASCII))


def test_header_is_valid_c99_and_cxx():
    """include/rtx.h is a C ABI: it must compile as C99 (examples/c_abi_demo.c uses it from plain C) and as
    C++; rtx_compat.hpp as C++17."""
    import subprocess
    inc = os.path.join(U.ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", inc,
                           os.path.join(U.ROOT, "examples", "c_abi_demo.c")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc,
                           os.path.join(U.ROOT, "examples", "headless_engine.cpp")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc,
                           os.path.join(U.ROOT, "examples", "console_engine.cpp")])

"""ctypes binding of the CPU oracle (oracle/librtx_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ORACLE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
_SO = os.path.join(ORACLE_DIR, "librtx_oracle.so")

BIT_ASCII, BIT_PIXEL, RGB_ASCII, RGB_PIXEL, RGB_NORMALS, SDL = range(6)
MODE_NAMES = ["BIT_ASCII", "BIT_PIXEL", "RGB_ASCII", "RGB_PIXEL", "RGB_NORMALS", "SDL"]
NONE, PLANE, SPHERE = 0, 1, 2
POW_LIBM, NORMALS_WRAP = 1, 2


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Object(C.Structure):
    _fields_ = [("type", C.c_int), ("center", Vec3), ("color", Vec3), ("radius", C.c_float),
                ("mover", C.c_int), ("speed", C.c_float), ("normal", Vec3),
                ("width", C.c_float), ("height", C.c_float)]


class Params(C.Structure):
    _fields_ = [("inv_v", (C.c_float * 4) * 4), ("cam_pos", C.c_float * 3),
                ("x", C.c_uint64), ("y", C.c_uint64),
                ("element1", C.c_float), ("element2", C.c_float), ("cam_far", C.c_float)]


class Pixel(C.Structure):
    _fields_ = [("distance", C.c_float), ("shading_value", C.c_float), ("normal", Vec3), ("color", Vec3),
                ("hit", C.c_int), ("ramp_index", C.c_int), ("ansi_index", C.c_int)]


PIXEL_DTYPE = np.dtype([("distance", "<f4"), ("shading_value", "<f4"), ("normal", "<f4", 3), ("color", "<f4", 3),
                        ("hit", "<i4"), ("ramp_index", "<i4"), ("ansi_index", "<i4")])
assert PIXEL_DTYPE.itemsize == C.sizeof(Pixel)


def build(force=False):
    """Compile the oracle with its Makefile (gcc only; no GPU, no reference needed)."""
    src = [os.path.join(ORACLE_DIR, f) for f in ("rtx_oracle.c", "rtx_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        PP = C.POINTER(C.POINTER(Object))
        L.orc_trace_pixel.argtypes = [C.POINTER(Params), PP, C.c_uint, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(Pixel)]
        L.orc_trace_pixel.restype = C.c_int
        L.orc_render_rows.argtypes = [C.POINTER(Params), PP, C.c_uint, C.c_int, C.c_size_t, C.c_size_t, C.c_int,
                                      C.c_void_p, C.c_void_p]
        L.orc_render_rows.restype = C.c_int
        L.orc_render_row.argtypes = [C.POINTER(Params), PP, C.c_uint, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
        L.orc_render_row.restype = C.c_int
        L.orc_render_mt.argtypes = [C.POINTER(Params), PP, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_render_mt.restype = C.c_int
        L.orc_minimize.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_minimize.restype = C.c_size_t
        L.orc_ansi256_from_rgb.argtypes = [C.c_uint32]
        L.orc_ansi256_from_rgb.restype = C.c_uint8
        L.orc_camera_params.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(Params)]
        L.orc_camera_params.restype = None
        L.orc_update_objects.argtypes = [PP, C.c_uint, C.c_double]
        L.orc_update_objects.restype = None
        L.orc_plane_normal.argtypes = [Vec3]
        L.orc_plane_normal.restype = Vec3
        L.orc_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_fnv1a64.restype = C.c_uint64
        L.orc_fnv1a64_from.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.orc_fnv1a64_from.restype = C.c_uint64
        L.orc_ansi256_exhaustive_hash.argtypes = [C.c_uint64]
        L.orc_ansi256_exhaustive_hash.restype = C.c_uint64
        L.orc_ansi256_fill.argtypes = [C.c_uint32, C.c_size_t, C.c_void_p]
        L.orc_ansi256_fill.restype = None
        _lib = L
    return _lib


class Scene:
    """AoS objects behind a pointer array, in creation order (Scene3D.cpp:36-86)."""

    def __init__(self):
        self._objs = []
        self._ptrs = None

    def add_sphere(self, radius, pos, color, mover=-1, speed=1.0):
        o = Object()
        o.type = SPHERE
        o.center = Vec3(*[float(v) for v in pos])
        o.color = Vec3(*[float(v) for v in color])
        o.radius = float(radius)
        o.mover = int(mover)
        o.speed = float(speed)
        self._objs.append(o)
        self._ptrs = None

    def add_plane(self, pos, normal, color, width, height):
        o = Object()
        o.type = PLANE
        o.center = Vec3(*[float(v) for v in pos])
        o.color = Vec3(*[float(v) for v in color])
        o.normal = lib().orc_plane_normal(Vec3(*[float(v) for v in normal]))
        o.width = float(width)
        o.height = float(height)
        self._objs.append(o)
        self._ptrs = None

    @property
    def count(self):
        return len(self._objs)

    def ptrs(self):
        if self._ptrs is None:
            arr = (C.POINTER(Object) * max(1, len(self._objs)))()
            for i, o in enumerate(self._objs):
                arr[i] = C.pointer(o)
            self._ptrs = arr
        return C.cast(self._ptrs, C.POINTER(C.POINTER(Object)))

    def objects(self):
        return self._objs

    @staticmethod
    def reference_default():
        """Scene3D.cpp:28-33."""
        s = Scene()
        s.add_sphere(7.0, (0.0, 10.0, 20.0), (255.0, 1.0, 1.0))
        s.add_sphere(6.0, (5.0, 10.0, 20.0), (1.0, 255.0, 1.0))
        s.add_sphere(10.0, (10.0, 10.0, 40.0), (1.0, 1.0, 255.0))
        s.add_sphere(3.0, (5.0, 10.0, 20.0), (225.0, 210.0, 20.0))
        s.add_sphere(4.0, (-5.0, 10.0, 40.0), (225.0, 10.0, 220.0))
        s.add_plane((0.0, -3.0, 30.0), (0.0, 1.0, 0.0), (100.0, 100.0, 100.0), 10, 20)
        return s

    @staticmethod
    def from_arrays(spheres, planes):
        """spheres: (N,7) cx cy cz r R G B; planes: (M,11) px py pz nx ny nz R G B w h.  Spheres first."""
        s = Scene()
        for row in np.asarray(spheres, dtype=np.float32).reshape(-1, 7):
            s.add_sphere(row[3], row[0:3], row[4:7])
        for row in np.asarray(planes, dtype=np.float32).reshape(-1, 11):
            s.add_plane(row[0:3], row[3:6], row[6:9], row[9], row[10])
        return s


DEFAULT_ROT = (0.0, float(np.float32(np.pi)), 0.0)  # Camera3D.h:62


def camera_params(w, h, pos=(0.0, 0.0, 0.0), rot=DEFAULT_ROT):
    p = Params()
    lib().orc_camera_params(w, h, (C.c_float * 3)(*pos), (C.c_float * 3)(*rot), C.byref(p))
    return p


def params_from_arrays(inv_v, cam, w, h, e1, e2, far):
    p = Params()
    m = np.asarray(inv_v, dtype=np.float32).reshape(4, 4)
    for i in range(4):
        for j in range(4):
            p.inv_v[i][j] = float(m[i, j])
    for i in range(3):
        p.cam_pos[i] = float(cam[i])
    p.x, p.y = int(w), int(h)
    p.element1, p.element2, p.cam_far = float(e1), float(e2), float(far)
    return p


def render(params, scene, mode, flags=0, threads=1, want_pixels=False, row0=0, rows=None):
    """Returns the zero-initialised 20*W*H buffer after the trace (uint8 array) [, per-pixel records]."""
    W, H = int(params.x), int(params.y)
    buf = np.zeros(20 * W * H, dtype=np.uint8)
    px = np.zeros(W * H, dtype=PIXEL_DTYPE) if want_pixels else None
    if threads > 1 and not want_pixels and row0 == 0 and rows is None:
        rc = lib().orc_render_mt(C.byref(params), scene.ptrs(), scene.count, mode, flags, threads, buf.ctypes.data)
    else:
        rc = lib().orc_render_rows(C.byref(params), scene.ptrs(), scene.count, mode, row0,
                                   H if rows is None else rows, flags, buf.ctypes.data,
                                   px.ctypes.data if want_pixels else None)
    if rc != 0:
        raise ValueError("oracle: invalid rendering mode %r" % (mode,))
    return (buf, px.reshape(H, W)) if want_pixels else buf


def render_row(params, scene, mode, row, flags=0):
    """Row `row` of the frame alone: W*S bytes (S = 12 or 20 by mode), without a frame-sized buffer."""
    W = int(params.x)
    S = 20 if mode >= RGB_ASCII else 12
    buf = np.zeros(W * S, dtype=np.uint8)
    rc = lib().orc_render_row(C.byref(params), scene.ptrs(), scene.count, mode, row, flags, buf.ctypes.data)
    if rc != 0:
        raise ValueError("oracle: invalid rendering mode %r" % (mode,))
    return buf


def minimize(mode, buf, w, h):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    out = np.zeros(buf.size + 1, dtype=np.uint8)
    n = lib().orc_minimize(mode, buf.ctypes.data, buf.size, w, h, out.ctypes.data)
    return out[:n].copy()


def ansi256_table(first=0, count=1 << 24):
    """orc_ansi256_from_rgb over packed 0xRRGGBB values first .. first+count-1, one byte each."""
    out = np.zeros(count, dtype=np.uint8)
    lib().orc_ansi256_fill(first, count, out.ctypes.data)
    return out


FNV_OFFSET_STANDARD = 14695981039346656037
FNV_OFFSET_SURVEY = 1469598103934665603  # what SURVEY.md 8(c)'s known answers were hashed from (see rtx_oracle.h)


def fnv1a64(buf, offset=FNV_OFFSET_STANDARD):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    return "%016x" % lib().orc_fnv1a64_from(buf.ctypes.data, buf.size, offset)

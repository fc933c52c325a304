"""MI355X (gfx950) ray-trace hot path of Raytracing-in-Windows-Console: Python plumbing.

The product is the C-ABI shared library ``librtx_hip.so`` (include/rtx.h) built from csrc/ by the
Makefile next to this file; this module only loads it through ctypes and wraps the entry points
for tests and bench.py.  There is no CPU rendering path: if the library is missing, or no gfx950
device is visible, everything here raises.

The directory name carries hyphens, so import it with
``importlib.import_module("raytracing-in-windows-console_amd")`` (see __graft_entry__.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, os.environ.get("RTX_LIB", "librtx_hip.so"))  # RTX_LIB: experiment builds only

# enum rtx_mode == enum RenderingMode (RayTracingManager.h:21)
BIT_ASCII, BIT_PIXEL, RGB_ASCII, RGB_PIXEL, RGB_NORMALS, SDL = range(6)
MODE_NAMES = ["BIT_ASCII", "BIT_PIXEL", "RGB_ASCII", "RGB_PIXEL", "RGB_NORMALS", "SDL"]
SIZE_8BIT, SIZE_RGB = 12, 20

OK, ERR_INVALID_ARGUMENT, ERR_INVALID_MODE, ERR_HIP, ERR_OUT_OF_MEMORY, ERR_NO_DEVICE, ERR_TOO_LARGE = range(7)
KERNEL_AUTO, KERNEL_BRUTE, KERNEL_BINNED = 0, 1, 2
OPT_KERNEL, OPT_TILE_LOG2_W, OPT_SUBTILES, OPT_TWO_LEVEL, OPT_REFINE, OPT_TILE_ORDER, OPT_CELL_CAPACITY, OPT_CELL_REUSE, OPT_XCD_ORDER, OPT_VIEW_ADAPT, OPT_SORTED_STORE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
STAT_CELL_BUILDS, STAT_CELL_PREFETCHES, STAT_CELL_HITS, STAT_CELL_PER_FRAME, STAT_ORDER_PASSES, STAT_ORDERS_FROZEN, STAT_CELL_CAPACITY_FLOOR, STAT_VIEW_DENSE, STAT_DENSITY_SWITCHES = 101, 102, 103, 104, 105, 106, 107, 108, 109
OPT_GROUP_EXCHANGE, OPT_GROUP_WIRE, OPT_BATCH, OPT_UPDATE_WORDS, OPT_GROUP_THREADS = 12, 13, 14, 15, 16
STAT_BATCHED_LAUNCHES = 114
OPT_MINIMIZE_FUSED = 17
OPT_GROUP_UPDATE = 18
OPT_UPDATE_HOST_WRITE = 19
STAT_UPDATE_HOST_WRITES = 117
STAT_GROUP_DIRECT_UPDATES = 116
STAT_MINIMIZE_FALLBACKS = 115
STAT_GROUP_SIZE, STAT_GROUP_EXCHANGE, STAT_GROUP_GATHERS, STAT_GROUP_BYTES = 110, 111, 112, 113
EXCHANGE_AUTO, EXCHANGE_PEER_COPY, EXCHANGE_RCCL, EXCHANGE_RCCL_ALL = 0, 1, 2, 3
WIRE_AUTO, WIRE_RECORDS, WIRE_COMPACT = 0, 1, 2
RENDER_ZERO_TAIL = 1
RENDER_COMPACT = 2
RENDER_VALUES = 4


class RtxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rtx status %d: %s" % (status, message))
        self.status = status


class Segment(C.Structure):
    """struct rtx_segment: one run of pixels for rtx_expand."""
    _fields_ = [("src_pixel", C.c_uint64), ("dst_pixel", C.c_uint64), ("n_pixels", C.c_uint64)]


class Params(C.Structure):
    """struct rtx_params (RayTracingCPUToGPUData, RayTracingManager.h:9-19)."""
    _fields_ = [("inv_v", C.c_float * 16), ("cam_pos", C.c_float * 3),
                ("element1", C.c_float), ("element2", C.c_float), ("cam_far", C.c_float),
                ("x", C.c_uint64), ("y", C.c_uint64)]


# every symbol include/rtx.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SIGNATURES = [
    ("rtx_create", C.c_int, [C.c_int, C.c_size_t, C.c_size_t, C.POINTER(_P)]),
    ("rtx_group_create", C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_size_t, C.c_size_t, C.POINTER(_P)]),
    ("rtx_group_size", C.c_int, [_P]),
    ("rtx_group_member", _P, [_P, C.c_int]),
    ("rtx_group_rows", C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("rtx_group_exchange_note", C.c_char_p, [_P]),
    ("rtx_destroy", None, [_P]),
    ("rtx_last_error", C.c_char_p, [_P]),
    ("rtx_version", C.c_char_p, []),
    ("rtx_set_option", C.c_int, [_P, C.c_int, C.c_int64]),
    ("rtx_get_option", C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    ("rtx_scene_clear", C.c_int, [_P]),
    ("rtx_scene_add_sphere", C.c_int, [_P, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float)]),
    ("rtx_scene_add_plane", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float]),
    ("rtx_scene_add_spheres", C.c_int, [_P, C.c_size_t, _P]),
    ("rtx_scene_count", C.c_uint, [_P]),
    ("rtx_scene_set_sphere_motion", C.c_int, [_P, C.c_uint, C.c_int, C.c_float]),
    ("rtx_scene_get_object", C.c_int, [_P, C.c_uint, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    ("rtx_render", C.c_int, [_P, C.POINTER(Params), C.c_int]),
    ("rtx_render_rows", C.c_int, [_P, C.POINTER(Params), C.c_int, C.c_size_t, C.c_size_t, _P, C.c_size_t, _P, C.c_uint]),
    ("rtx_submit_frames", C.c_int, [_P, C.c_size_t, C.POINTER(Params), C.c_int, C.POINTER(_P), C.POINTER(_P)]),
    ("rtx_submit_slabs", C.c_int, [_P, C.c_size_t, C.POINTER(Params), C.c_int, C.c_size_t, C.c_size_t, C.POINTER(_P), C.c_size_t,
                                   C.POINTER(_P), _P, C.c_uint]),
    ("rtx_expand", C.c_int, [_P, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    ("rtx_graph_begin", C.c_int, [_P, _P]),
    ("rtx_graph_end", C.c_int, [_P, _P, C.POINTER(_P)]),
    ("rtx_graph_launch", C.c_int, [_P, _P, _P]),
    ("rtx_graph_destroy", None, [_P, _P]),
    ("rtx_synchronize", C.c_int, [_P]),
    ("rtx_frame_device_ptr", _P, [_P]),
    ("rtx_frame_capacity", C.c_size_t, [_P]),
    ("rtx_read_frame", C.c_int, [_P, _P, C.c_size_t]),
    ("rtx_minimize", C.c_int, [_P, C.c_int, C.c_size_t, C.c_size_t, _P, _P, C.POINTER(C.c_size_t)]),
    ("rtx_minimized_device_ptr", _P, [_P]),
    ("rtx_minimize_words", C.c_int, [_P, C.c_int, C.c_size_t, C.c_size_t, _P, _P, C.POINTER(C.c_size_t)]),
    ("rtx_update_objects", C.c_int, [_P, C.c_double]),
    ("rtx_update", C.c_int, [_P, C.POINTER(Params), C.c_int, C.c_double, C.c_int, _P, C.POINTER(C.c_size_t)]),
    ("rtx_update_begin", C.c_int, [_P, C.POINTER(Params), C.c_int, C.c_double, C.c_int, _P, C.POINTER(C.c_int)]),
    ("rtx_update_end", C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t)]),
    ("rtx_ansi256_map", C.c_int, [_P, C.c_uint32, C.c_size_t, _P, _P]),
    ("rtx_host_alloc", _P, [_P, C.c_size_t]),
    ("rtx_host_free", None, [_P, _P]),
    ("rtx_timer_start", C.c_int, [_P]),
    ("rtx_timer_stop", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("rtx_last_kernel_name", C.c_char_p, [_P]),
    ("rtx_camera_params", C.c_int, [C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(Params)]),
    ("rtx_synth_scene", C.c_int, [C.c_uint32, C.c_size_t, C.c_size_t, C.c_float, C.c_float, _P, _P]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]


def build(verbose=False):
    """hipcc --offload-arch=gfx950 over csrc/ (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", PKG_DIR], stdout=out)
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                               "This package has no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64 (SONAME libamdhip64.so.7).
        # Loading it first makes this library's DT_NEEDED libamdhip64.so.7 resolve to that same copy;
        # the other order would put two runtimes in the process and the second one sees no GPU.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, restype, argtypes in _SIGNATURES:
            fn = getattr(L, name)  # AttributeError if the .so lacks a symbol the header declares
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def camera_params(w, h, pos=None, rot=None):
    """rtx_camera_params: what Engine3D::Render fills (Engine3D.cpp:90-97); defaults = the start pose."""
    p = Params()
    fp = (C.c_float * 3)(*pos) if pos is not None else None
    fr = (C.c_float * 3)(*rot) if rot is not None else None
    rc = lib().rtx_camera_params(w, h, fp, fr, C.byref(p))
    if rc != OK:
        raise RtxError(rc, "rtx_camera_params")
    return p


def synth_scene(seed, n_spheres, n_planes, e1, e2):
    """SURVEY.md Appendix D scene: (spheres (N,7) float32, planes (M,11) float32)."""
    sph = np.zeros((n_spheres, 7), dtype=np.float32)
    pl = np.zeros((n_planes, 11), dtype=np.float32)
    rc = lib().rtx_synth_scene(seed, n_spheres, n_planes, e1, e2, sph.ctypes.data, pl.ctypes.data)
    if rc != OK:
        raise RtxError(rc, "rtx_synth_scene")
    return sph, pl


# BASELINE.json configs: (W, H, spheres, planes, LCG seed)  
CONFIGS = {
    "C1": (320, 180, 8, 1, 1),
    "C2": (1920, 1080, 1024, 1, 2),
    "C3": (3840, 2160, 4096, 6, 3),
    "C4": (7680, 4320, 1024, 0, 4),
    "C5": (1920, 1080, 65536, 0, 5),
}


def config_inputs(name):
    w, h, ns, npl, seed = CONFIGS[name]
    p = camera_params(w, h)
    sph, pl = synth_scene(seed, ns, npl, p.element1, p.element2)
    return p, sph, pl


class Context:
    """One rtx_ctx: what RayTracingManager + Scene3D own on the device (RayTracingManager.cu:53-74)."""

    def __init__(self, max_w, max_h, device=0, devices=None):
        """devices: a list of HIP device ordinals -> a device group (rtx_group_create): the frame shards by rows over them,
        one logical rank per entry (an ordinal may repeat), and is assembled on devices[0]."""
        self._h = _P()
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = lib().rtx_group_create(len(devices), devs, max_w, max_h, C.byref(self._h))
        else:
            rc = lib().rtx_create(device, max_w, max_h, C.byref(self._h))
        if rc != OK:
            raise RtxError(rc, (lib().rtx_last_error(None) or b"").decode())
        self.max_w, self.max_h = max_w, max_h

    # -- device groups
    @property
    def group_size(self):
        return lib().rtx_group_size(self._h)

    def group_rows(self, h, rank):
        r0, n = C.c_size_t(), C.c_size_t()
        self._check(lib().rtx_group_rows(self._h, h, rank, C.byref(r0), C.byref(n)))
        return r0.value, n.value

    @property
    def exchange_note(self):
        return (lib().rtx_group_exchange_note(self._h) or b"").decode()

    def member_kernel(self, rank):
        """Name of the kernel logical rank `rank` launched last."""
        m = lib().rtx_group_member(self._h, rank)
        return (lib().rtx_last_kernel_name(m) or b"").decode() if m else ""

    def member_option(self, rank, opt):
        m = lib().rtx_group_member(self._h, rank)
        v = C.c_int64(0)
        self._check(lib().rtx_get_option(m, opt, C.byref(v)))
        return v.value

    def close(self):
        if self._h:
            if getattr(self, "_pinned", None):
                lib().rtx_host_free(self._h, self._pinned[0])
                self._pinned = None
            lib().rtx_destroy(self._h)
            self._h = _P()

    def _pinned_buffer(self, nbytes):
        """A pinned host buffer of at least nbytes (what m_minimizedResultArray is in the facade)."""
        cur = getattr(self, "_pinned", None)
        if cur is None or cur[1] < nbytes:
            if cur is not None:
                lib().rtx_host_free(self._h, cur[0])
            p = lib().rtx_host_alloc(self._h, nbytes)
            if not p:
                raise RtxError(ERR_OUT_OF_MEMORY, "rtx_host_alloc")
            self._pinned = (p, nbytes, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)))
        return self._pinned

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != OK:
            raise RtxError(rc, (lib().rtx_last_error(self._h) or b"").decode())

    # -- options
    def set_option(self, opt, value):
        self._check(lib().rtx_set_option(self._h, opt, value))

    def get_option(self, opt):
        v = C.c_int64(0)
        self._check(lib().rtx_get_option(self._h, opt, C.byref(v)))
        return v.value

    # -- scene (Scene3D::CreateSphere / CreatePlane, Scene3D.cpp:36-86)
    def scene_clear(self):
        self._check(lib().rtx_scene_clear(self._h))

    def add_sphere(self, radius, pos, color):
        idx = lib().rtx_scene_add_sphere(self._h, (C.c_float * 3)(*pos), radius, (C.c_float * 3)(*color))
        if idx < 0:
            self._check(-idx)
        return idx

    def add_plane(self, pos, normal, color, width, height):
        idx = lib().rtx_scene_add_plane(self._h, (C.c_float * 3)(*pos), (C.c_float * 3)(*normal),
                                        (C.c_float * 3)(*color), width, height)
        if idx < 0:
            self._check(-idx)
        return idx

    def add_spheres(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, 7)
        idx = lib().rtx_scene_add_spheres(self._h, arr.shape[0], arr.ctypes.data)
        if idx < 0 and arr.shape[0]:
            self._check(-idx)
        return idx

    def set_scene(self, spheres, planes):
        """Spheres first, then planes (the order SURVEY App. D defines)."""
        self.scene_clear()
        self.add_spheres(spheres)
        for row in np.asarray(planes, dtype=np.float32).reshape(-1, 11):
            self.add_plane(row[0:3], row[3:6], row[6:9], float(row[9]), float(row[10]))

    def set_reference_default_scene(self):
        """Scene3D::Init, Scene3D.cpp:28-33."""
        self.scene_clear()
        self.add_sphere(7.0, (0.0, 10.0, 20.0), (255.0, 1.0, 1.0))
        self.add_sphere(6.0, (5.0, 10.0, 20.0), (1.0, 255.0, 1.0))
        self.add_sphere(10.0, (10.0, 10.0, 40.0), (1.0, 1.0, 255.0))
        self.add_sphere(3.0, (5.0, 10.0, 20.0), (225.0, 210.0, 20.0))
        self.add_sphere(4.0, (-5.0, 10.0, 40.0), (225.0, 10.0, 220.0))
        self.add_plane((0.0, -3.0, 30.0), (0.0, 1.0, 0.0), (100.0, 100.0, 100.0), 10.0, 20.0)

    @property
    def object_count(self):
        return lib().rtx_scene_count(self._h)

    def set_sphere_motion(self, index, mover, speed):
        self._check(lib().rtx_scene_set_sphere_motion(self._h, index, mover, speed))

    def get_object(self, index):
        t = C.c_int()
        out = (C.c_float * 11)()
        self._check(lib().rtx_scene_get_object(self._h, index, C.byref(t), out))
        return t.value, np.array(out[:], dtype=np.float32)

    # -- render
    def render(self, params, mode):
        self._check(lib().rtx_render(self._h, C.byref(params), mode))

    def render_rows(self, params, mode, row0, rows, d_out=None, out_row_base=0, stream=None, flags=0):
        self._check(lib().rtx_render_rows(self._h, C.byref(params), mode, row0, rows, d_out, out_row_base, stream, flags))

    def submit_frames(self, params_list, mode, d_outs, streams):
        """rtx_submit_frames: queue len(params_list) whole frames with one call."""
        n = len(params_list)
        pa = (Params * n)(*params_list)
        oa = (_P * n)(*d_outs)
        sa = (_P * n)(*streams)
        self._check(lib().rtx_submit_frames(self._h, n, pa, mode, oa, sa))

    def make_submitter(self, params, mode, d_outs, streams):
        """Pre-built argument arrays for repeatedly queueing the same batch (bench.py): returns a callable.
        `params`: one rtx_params for every ring position, or a list with one per position (a camera per frame)."""
        n = len(d_outs)
        plist = list(params) if isinstance(params, (list, tuple)) else [params] * n
        if len(plist) != n:
            raise ValueError("make_submitter: %d params for %d ring positions" % (len(plist), n))
        pa = (Params * n)(*plist)
        oa = (_P * n)(*d_outs)
        sa = (_P * n)(*streams)
        fn, h = lib().rtx_submit_frames, self._h
        # pointer triples for every ring position, built once: the per-frame host cost is one foreign call
        slots = [(C.cast(C.byref(pa, i * C.sizeof(Params)), C.POINTER(Params)),
                  C.cast(C.byref(oa, i * C.sizeof(_P)), C.POINTER(_P)),
                  C.cast(C.byref(sa, i * C.sizeof(_P)), C.POINTER(_P))) for i in range(n)]
        keep = (pa, oa, sa)

        def submit(count=n, first=0):
            # frames first .. first+count-1 of the ring (count <= n - first)
            p_, o_, s_ = slots[first]
            rc = fn(h, count, p_, mode, o_, s_)
            if rc != OK:
                self._check(rc)
        submit._keep = keep
        return submit

    def submit_slabs(self, params_list, mode, row0, rows, d_outs, out_row_base, streams, after=None, flags=0):
        """rtx_submit_slabs: rows [row0, row0+rows) of len(params_list) frames with one call."""
        n = len(params_list)
        pa = (Params * n)(*params_list)
        oa = (_P * n)(*d_outs)
        sa = (_P * n)(*streams)
        self._check(lib().rtx_submit_slabs(self._h, n, pa, mode, row0, rows, oa, out_row_base, sa, after, flags))

    def make_slab_submitter(self, params, mode, row0, rows, out_row_base, d_outs, streams, after, flags=0):
        """rtx_submit_slabs with pre-built argument arrays: rows [row0, row0+rows) of len(d_outs) frames, frame i
        into d_outs[i] on streams[i], forked from / joined into the stream `after`.  Returns submit(count)."""
        n = len(d_outs)
        pa = (Params * n)(*([params] * n))
        oa = (_P * n)(*d_outs)
        sa = (_P * n)(*streams)
        fn, h = lib().rtx_submit_slabs, self._h

        def submit(count=n):
            rc = fn(h, count, pa, mode, row0, rows, oa, out_row_base, sa, after, flags)
            if rc != OK:
                self._check(rc)
        submit._keep = (pa, oa, sa)
        return submit

    def expand(self, mode, d_compact, d_out, segments, stream=None):
        """rtx_expand: compact pixel words -> records; segments = [(src_pixel, dst_pixel, n_pixels), ...]."""
        n = len(segments)
        sa = (Segment * n)(*[Segment(*g) for g in segments])
        self._check(lib().rtx_expand(self._h, mode, d_compact, d_out, sa, n, stream))

    def make_expander(self, mode, d_compact, d_out, segments, stream=None):
        """rtx_expand with a pre-built segment array: returns a callable."""
        n = len(segments)
        sa = (Segment * n)(*[Segment(*g) for g in segments])
        fn, h = lib().rtx_expand, self._h

        def expand():
            rc = fn(h, mode, d_compact, d_out, sa, n, stream)
            if rc != OK:
                self._check(rc)
        expand._keep = sa
        return expand

    # -- HIP graphs: record a launch sequence once, replay it with one host call
    def graph_begin(self, stream=None):
        self._check(lib().rtx_graph_begin(self._h, stream))

    def graph_end(self, stream=None):
        g = _P()
        self._check(lib().rtx_graph_end(self._h, stream, C.byref(g)))
        return g

    def graph_launch(self, graph, stream=None):
        self._check(lib().rtx_graph_launch(self._h, graph, stream))

    def graph_launcher(self, graph, stream=None):
        """A callable that replays `graph` on `stream` (pre-bound: one foreign call per replay)."""
        fn, h = lib().rtx_graph_launch, self._h

        def launch():
            rc = fn(h, graph, stream)
            if rc != OK:
                self._check(rc)
        launch._keep = graph
        return launch

    def graph_destroy(self, graph):
        lib().rtx_graph_destroy(self._h, graph)

    def synchronize(self):
        self._check(lib().rtx_synchronize(self._h))

    def read_frame(self, nbytes):
        buf = np.empty(nbytes, dtype=np.uint8)
        self._check(lib().rtx_read_frame(self._h, buf.ctypes.data, nbytes))
        return buf

    def render_to_host(self, params, mode):
        """Trace + copy back the 20*W*H frame (what RayTracingManager.cu:127-143 leaves in m_hostResultArray)."""
        self.render(params, mode)
        return self.read_frame(20 * int(params.x) * int(params.y))

    @property
    def frame_ptr(self):
        return lib().rtx_frame_device_ptr(self._h)

    @property
    def last_kernel(self):
        return (lib().rtx_last_kernel_name(self._h) or b"").decode()

    # -- minimise / physics / whole Update
    def minimize(self, mode, w, h, d_in=None, d_out=None):
        n = C.c_size_t()
        self._check(lib().rtx_minimize(self._h, mode, w, h, d_in, d_out, C.byref(n)))
        return n.value

    def minimize_words(self, mode, w, h, d_words, d_out=None):
        n = C.c_size_t()
        self._check(lib().rtx_minimize_words(self._h, mode, w, h, d_words, d_out, C.byref(n)))
        return n.value

    @property
    def minimized_ptr(self):
        return lib().rtx_minimized_device_ptr(self._h)

    def update_objects(self, dt):
        self._check(lib().rtx_update_objects(self._h, dt))

    def update(self, params, mode, dt=0.0, run_physics=False):
        """RayTracingManager::Update: returns the minimised byte stream handed to PrintMachine."""
        ptr, _, arr = self._pinned_buffer(20 * int(params.x) * int(params.y))
        n = C.c_size_t()
        self._check(lib().rtx_update(self._h, C.byref(params), mode, dt, 1 if run_physics else 0, ptr, C.byref(n)))
        return arr[:n.value]  # a view of the pinned buffer: valid until the next update()

    def ansi256_map(self, first_rgb, count, d_out, stream=None):
        """rtx_ansi256_map: xterm-256 indices of packed 0xRRGGBB values first_rgb .. first_rgb+count-1 into d_out."""
        self._check(lib().rtx_ansi256_map(self._h, first_rgb, count, d_out, stream))

    def host_alloc(self, nbytes):
        """Pinned host buffer as (pointer, uint8 numpy view); freed with host_free or at close()."""
        p = lib().rtx_host_alloc(self._h, nbytes)
        if not p:
            raise RtxError(ERR_OUT_OF_MEMORY, "rtx_host_alloc")
        return p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def host_free(self, p):
        lib().rtx_host_free(self._h, p)

    def update_begin(self, params, mode, host_ptr, dt=0.0, run_physics=False):
        t = C.c_int()
        self._check(lib().rtx_update_begin(self._h, C.byref(params), mode, dt, 1 if run_physics else 0, host_ptr, C.byref(t)))
        return t.value

    def update_end(self, ticket):
        n = C.c_size_t()
        self._check(lib().rtx_update_end(self._h, ticket, C.byref(n)))
        return n.value

    # -- timing on the context's stream
    def timer_start(self):
        self._check(lib().rtx_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        self._check(lib().rtx_timer_stop(self._h, C.byref(ms)))
        return ms.value

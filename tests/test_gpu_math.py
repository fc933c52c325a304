"""Exhaustive proof, on the GPU, that the short correctly-rounded reciprocal / square root used by the
kernels (rtx_device.hpp: rcp_cr, sqrt_cr) return exactly the bits of the compiler's IEEE expansions of
1.0f/x and sqrtf(x) -- for every one of the 2^32 fp32 inputs, and for the composition 1.0f/sqrt(x)
that Normalize_GPU (MyMath.h:139-145) evaluates."""
import ctypes as C
import os

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def checker():
    import torch  # noqa: F401  (one HIP runtime per process: torch's copy first)
    path = os.path.join(HERE, "gpu_checks", "libmath_check.so")
    assert os.path.exists(path), "run __graft_entry__.build()"
    lib = C.CDLL(path)
    lib.rtx_check_math_exhaustive.restype = C.c_longlong
    lib.rtx_check_math_exhaustive.argtypes = [C.c_int, C.POINTER(C.c_uint)]
    return lib


@pytest.mark.parametrize("which,name", [(0, "rcp_cr(x) == 1.0f/x"), (1, "sqrt_cr(x) == sqrtf(x)"),
                                        (2, "rcp_cr(sqrt_cr(x)) == 1.0f/sqrtf(x)")])
def test_short_math_is_bit_identical_on_all_inputs(checker, which, name):
    first = C.c_uint(0)
    bad = checker.rtx_check_math_exhaustive(which, C.byref(first))
    assert bad == 0, "%s fails on %d inputs, first bit pattern 0x%08x" % (name, bad, first.value)

"""The directed 8K scene of tests/test_gpu_parity.py::test_side_planes_of_thin_tiles_with_a_rolled_camera_at_8k is built on the host
(tools/wide_view_directed_gpu.py emulates the fp32 cross-product planes that rounds 1-2 used): checked here without a GPU, so
that the GPU test never runs on an empty scene."""
import os
import sys

import numpy as np

import util as U

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_directed_scene_finds_tiles_the_old_planes_lose():
    from wide_view_directed_gpu import directed_scene
    R = U.pkg()
    p, sph, report = directed_scene(R, 7680, 4320, (0.1, 2.8, 0.3), (1.0, 2.0, -1.0))
    assert len(sph) >= 6 and len(report) == len(sph)
    assert sph.dtype == np.float32 and sph.shape[1] == 7 and np.isfinite(sph).all()
    # every chosen tile: its own boundary-row rays at least 2e-5 rad outside the emulated plane (half a pixel is 5.3e-6 there)
    assert all(float(line.split("rays ")[1].split(" rad")[0]) >= 2.0e-5 for line in report), report
    # the matrix went into the parameters as given: a rotation with roll (m[4] != 0), orthonormal to fp32
    m = np.array(p.inv_v, dtype=np.float64).reshape(4, 4)[:3, :3]
    assert abs(m[1, 0]) > 0.1 and np.allclose(m @ m.T, np.eye(3), atol=1e-6)
    # spheres are where a pixel ray of the first 16 columns can reach them: r = 20, centres ~200 away
    d = np.linalg.norm(sph[:, :3].astype(np.float64) - np.array([1.0, 2.0, -1.0]), axis=1)
    assert np.all(sph[:, 3] == 20.0) and np.all((d > 150.0) & (d < 250.0))

"""bench.py's side of the driver contract that can be checked without a GPU."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_metric_is_baselines_metric_verbatim():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert _bench().BASELINE_METRIC == json.load(f)["metric"]


def test_algorithmic_bytes_and_flops_follow_survey_8d():
    b = _bench()
    # C2: 1919*1080 rays, 20 B records, 1024 spheres + 1 plane, 88 B of params (41.48 MB)
    assert b.algorithmic_bytes(1920, 1080, 20, 1024, 1) == 1919 * 1080 * 20 + 28 * 1024 + 44 + 88 == 41479204
    assert b.algorithmic_bytes(1920, 1080, 20, 1024, 1, rows=135) == 1919 * 135 * 20 + 28 * 1024 + 44 + 88
    assert b.algorithmic_flops(1920, 1080, 1024, 1, 0.5) == 1919 * 1080 * (19.0 * 1024 + 7.0 + 30.0 + 75.0)


def test_golden_frames_carry_the_hash_the_bench_checks():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        g = json.load(f)
    for key in ("C1_RGB_ASCII", "C2_RGB_ASCII", "C2_BIT_ASCII", "C3_RGB_ASCII", "C4_RGB_ASCII", "C5_RGB_ASCII"):
        assert len(g[key]["frame_sha256"]) == 64


def test_committed_counters_are_dropped_when_the_sources_changed(tmp_path):
    """roofline.traffic / roofline.valu come from profiles/counters.json only while the library sources still hash to
    what was profiled: a kernel edit that keeps the kernel's name must not report stale counters."""
    b = _bench()
    root = tmp_path
    (root / b.PKG / "csrc").mkdir(parents=True)
    (root / "include").mkdir()
    (root / "profiles").mkdir()
    (root / b.PKG / "csrc" / "k.hip").write_text("__global__ void k() {}\n")
    (root / "include" / "rtx.h").write_text("/* abi */\n")
    h = b.csrc_sha256(str(root))
    entry = {"csrc_sha256": h, "SQ_INSTS_VALU": 10.0, "total_bytes": 5.0}
    (root / "profiles" / "counters.json").write_text(json.dumps({"C2_RGB_ASCII_k": entry, "C2_RGB_ASCII_old": {"SQ_INSTS_VALU": 1.0}}))
    got, why = b.committed_counters("C2", "RGB_ASCII", "k", str(root))
    assert got == entry and why is None
    got, why = b.committed_counters("C2", "RGB_ASCII", "old", str(root))      # an entry from before the hash existed
    assert got is None and "no source hash" in why
    got, why = b.committed_counters("C2", "RGB_ASCII", "absent", str(root))
    assert got is None and why
    (root / b.PKG / "csrc" / "k.hip").write_text("__global__ void k() { /* edited */ }\n")
    got, why = b.committed_counters("C2", "RGB_ASCII", "k", str(root))
    assert got is None and "other sources" in why
    r = b.roofline_object("C2", "RGB_ASCII", "k", 1920, 1080, 20, 1024, 1, 1080, 0.025, 0.5)
    assert r["bytes_per_launch"] == 41479204 and abs(r["achieved"] - 41479204 / 25e-6 / 1e9) < 0.1


def test_moving_view_ring_steps_by_a_constant_angle():
    b = _bench()

    class FakeR:
        @staticmethod
        def camera_params(w, h, pos=None, rot=None):
            return rot[1]
    import math
    ys = b.moving_cameras(FakeR, 1920, 1080)
    assert len(ys) == 1000
    for i in range(1000):
        assert abs(abs(ys[(i + 1) % 1000] - ys[i]) - b.MOVING_STEP_RAD) < 1e-9     # also across the wrap-around
    assert abs(max(ys) - math.pi - 0.25) < 1e-9 and abs(min(ys) - math.pi + 0.25) < 1e-9

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

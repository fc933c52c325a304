"""csrc/rtx_plan.hpp -- tile shapes, the coarse-cell grid, the dispatch-order state machine, the cell-list reuse policy and its
motion bounds -- compiled as host-only C++ and run under AddressSanitizer + UndefinedBehaviorSanitizer (tests/host/test_plan.cpp).
No HIP, no GPU: the planning behind rtx_render_rows is pure."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_planners_under_sanitizers(tmp_path):
    exe = str(tmp_path / "test_plan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
                           "-Werror", os.path.join(ROOT, "tests", "host", "test_plan.cpp"), "-o", exe])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "all host planning tests passed" in p.stdout, p.stdout[-4000:]


def test_resident_workgroups_per_cu_is_one_constant():
    """The trace kernels' occupancy (RTX_WAVES_PER_EU) is named in one place on the host side: rtxplan::kResidentPerCU."""
    api = open(os.path.join(ROOT, "raytracing-in-windows-console_amd", "csrc", "rtx_render.cpp")).read()
    assert "7ull *" not in api and "resident = 7u" not in api
    assert "rtxplan::resident_slots" in api and "rtxplan::kResidentPerCU" in api
    plan = open(os.path.join(ROOT, "raytracing-in-windows-console_amd", "csrc", "rtx_plan.hpp")).read()
    assert "kResidentPerCU = RTX_WAVES_PER_EU" in plan

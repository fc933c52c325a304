"""`python bench.py --gpus N` without a launcher must start its N ranks itself (as a child process, before any GPU call),
relay rank 0's one JSON line and the exit code.  Exercised here on CPU: --dry runs the N > 1 loops over gloo with the CPU
oracle as the renderer (world_size 2), through the very entry the driver uses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RTX_BENCH_FORCE_DIST")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def _one_line(proc):
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one line on stdout, got %r\nstderr:\n%s" % (lines, proc.stderr[-2000:])
    return json.loads(lines[0])


def test_self_launch_world2_over_gloo_with_cpu_baseline_and_sub_config():
    p = _run("--gpus", "2", "--dry", "--steps", "6", "--warmup", "1", "--sub-configs", "C1", "--cpu-threads", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    d = _one_line(p)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert d["metric"] == json.load(f)["metric"]
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 1 and d["dry"] is True
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["verified_against_golden"] is True          # the frame assembled over gloo has the committed golden hash
    assert d["ms_per_step"] > 0 and d["value"] > 0
    assert d["config"]["rows_per_rank"] == 90             # C1: 180 rows over 2 ranks
    # the N > 1 line carries the CPU baseline (rank 0, after the ranks are done) ...
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 2 and cpu["value"] > 0 and "320x180" in cpu["sample"]
    # ... and the sub-record of the second configuration with its own step time, verification and (null in a dry run) roofline
    sub = d["configs"]["C1"]
    assert sub["verified_against_golden"] is True and sub["ms_per_step"] > 0 and "roofline" in sub and sub["n_gpus"] == 2


@pytest.mark.parametrize("extra", [["--exchange", "p2p", "--root", "fixed"], ["--exchange", "p2p"], ["--exchange", "rounds"],
                                   ["--exchange", "compact", "--root", "fixed", "--frames-per-root", "3"]])
def test_self_launch_other_exchanges(extra):
    p = _run("--gpus", "2", "--dry", "--steps", "5", "--warmup", "0", "--no-cpu-baseline", *extra)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _one_line(p)
    assert d["verified_against_golden"] is True and d["cpu_baseline"] is None and "configs" not in d


def test_self_launch_relays_a_failing_rank_as_a_failure():
    p = _run("--gpus", "2", "--dry", "--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--sub-configs", "C9")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]


def test_under_a_launcher_it_is_a_rank_not_a_launcher():
    # WORLD_SIZE present but inconsistent with --gpus: the process must take itself for a rank (and refuse), not spawn anything
    p = _run("--gpus", "2", "--dry", "--steps", "2", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and ("WORLD_SIZE" in p.stderr or "--dry" in p.stderr)


def test_parent_never_imports_torch_before_spawning():
    """The launcher half of bench.py must not touch the GPU: it does not even import torch."""
    code = ("import sys, runpy; sys.argv=['bench.py','--gpus','2','--dry','--steps','2','--warmup','0','--no-cpu-baseline'];\n"
            "import subprocess\n"
            "orig = subprocess.Popen\n"
            "class P(orig):\n"
            "    def __init__(self, *a, **k):\n"
            "        assert 'torch' not in sys.modules, 'torch imported before the ranks were spawned'\n"
            "        print('SPAWN_OK', file=sys.stderr)\n"
            "        super().__init__(*a, **k)\n"
            "subprocess.Popen = P\n"
            "runpy.run_path(%r, run_name='__main__')\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert "SPAWN_OK" in p.stderr and p.returncode == 0, p.stderr[-2000:]

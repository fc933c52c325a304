#!/usr/bin/env python3
"""The reference's own workload: a console-sized frame (its default scene at 400 x 150 and at 200 x 60; BASELINE config 1) through the
blocking rtx_update (physics + trace + Minimize + hand-off to the host), wall clock per Update over 3000 frames.  At this size nothing
is bound by bandwidth: the figure is launches, host synchronisations and the two small copies.  GPU only."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = importlib.import_module("raytracing-in-windows-console_amd")


def run(label, make, W, H, mode, n=3000):
    with R.Context(W, H) as c:
        p = make(c)
        for _ in range(300):
            got = c.update(p, mode, dt=0.016, run_physics=True)
        t0 = time.perf_counter()
        for _ in range(n):
            got = c.update(p, mode, dt=0.016, run_physics=True)
        us = (time.perf_counter() - t0) / n * 1e6
        hb = [c.host_alloc(20 * W * H) for _ in range(2)]
        tick = []
        for i in range(n + 300):
            if i == 300:
                while tick:
                    c.update_end(tick.pop(0))
                t0 = time.perf_counter()
            if len(tick) == 2:
                c.update_end(tick.pop(0))
            tick.append(c.update_begin(p, mode, hb[i % 2][0], dt=0.016, run_physics=True))
        while tick:
            c.update_end(tick.pop(0))
        us2 = (time.perf_counter() - t0) / n * 1e6
        for q, _ in hb:
            c.host_free(q)
        print("%-34s %-10s %4d x %-4d  blocking %7.2f us per Update, pipelined %7.2f   (%d bytes handed over)" % (label, R.MODE_NAMES[mode], W, H, us, us2, len(got)), flush=True)


def default_scene(W, H):
    def make(c):
        c.set_reference_default_scene()
        return R.camera_params(W, H)
    return make


def config(name):
    def make(c):
        p, sph, pl = R.config_inputs(name)
        c.set_scene(sph, pl)
        return p
    return make


if "--soak" in sys.argv:
    # minutes of pipelined Updates of the default scene: every frame's length against the first's is not checked (the spheres move) --
    # what is watched is that no Minimize launch ever gives up (RTX_STAT_MINIMIZE_FALLBACKS) and that the rate holds
    secs = float(next((a.split("=")[1] for a in sys.argv if a.startswith("--seconds=")), "60"))
    with R.Context(400, 150) as c:
        c.set_reference_default_scene()
        p = R.camera_params(400, 150)
        hb = [c.host_alloc(20 * 400 * 150) for _ in range(2)]
        tick, n, t0 = [], 0, time.perf_counter()
        while time.perf_counter() - t0 < secs:
            for _ in range(1000):
                if len(tick) == 2:
                    c.update_end(tick.pop(0))
                tick.append(c.update_begin(p, (R.BIT_ASCII, R.RGB_ASCII)[n & 1], hb[n % 2][0], dt=0.016, run_physics=True))
                n += 1
        while tick:
            c.update_end(tick.pop(0))
        dt = time.perf_counter() - t0
        print("soak: %d pipelined Updates in %.1f s (%.2f us each), Minimize launches that gave up: %d, host-write Updates %d" % (
            n, dt, dt / n * 1e6, c.get_option(R.STAT_MINIMIZE_FALLBACKS), c.get_option(R.STAT_UPDATE_HOST_WRITES)), flush=True)
    sys.exit(0)

for mode in (R.BIT_ASCII, R.RGB_ASCII):
    run("reference default scene", default_scene(400, 150), 400, 150, mode)
    run("reference default scene", default_scene(200, 60), 200, 60, mode)
    run("config 1 (8 spheres + 1 plane)", config("C1"), 320, 180, mode)

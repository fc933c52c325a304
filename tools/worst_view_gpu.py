#!/usr/bin/env python3
"""Where does a turning camera hurt?  C2's scene seen from yaw pi .. pi + 2.4 (the sweep of tools/moving_camera_gpu.py, whose
average is 40 us per launch alone against 26 for the default view): per view, the time of one launch alone (product
library), and -- `stamps` as first argument: experiment library -- what its workgroups saw: candidates per macro tile
(list overflow sets in past 192 after the first staging step and past 704 in all) and the longest workgroup lifetimes.

  python tools/worst_view_gpu.py            # timings, product library
  python tools/worst_view_gpu.py stamps     # per-workgroup candidates and lifetimes, librtx_hip_ablate.so
  --config=C5 (or C3 ...): another BASELINE scene; --coarse: every 0.2 rad; --yaws=1.2,1.4: these offsets only
"""
import importlib
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stamps = len(sys.argv) > 1 and sys.argv[1] == "stamps"
if stamps:
    os.environ["RTX_LIB"] = os.environ.get("RTX_STAMPS_LIB", "librtx_hip_ablate.so")
    os.environ["RTX_ABLATE"] = str(0x8000)
import torch  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
CONFIG = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--config=")), "C2")
p0, sph, pl = R.config_inputs(CONFIG)
W, H = int(p0.x), int(p0.y)
yaws = [math.pi + 0.1 * k for k in range(25)]
if any(a == "--coarse" for a in sys.argv[1:]):
    yaws = [math.pi + 0.2 * k for k in range(13)]
for a in sys.argv[1:]:
    if a.startswith("--yaws="):
        yaws = [math.pi + float(v) for v in a.split("=")[1].split(",")]
extra = [a for a in sys.argv[1:] if a.startswith("--")]
for yaw in yaws:
    cam = R.camera_params(W, H, (0.0, 0.0, 0.0), (0.0, yaw, 0.0))
    ctx = R.Context(W, H)
    ctx.set_scene(sph, pl)
    if "--two-level" in extra:
        ctx.set_option(R.OPT_TWO_LEVEL, 1)
    for e in extra:
        if e.startswith("--subtiles="):
            ctx.set_option(R.OPT_SUBTILES, int(e.split("=")[1]))
        if e.startswith("--refine="):
            ctx.set_option(R.OPT_REFINE, int(e.split("=")[1]))
    ctx.set_option(R.OPT_TILE_ORDER, 0)       # frame order: the view's own cost, not what balancing makes of it
    if "--no-adapt" in extra:
        ctx.set_option(R.OPT_VIEW_ADAPT, 0)
    for _ in range(48):                        # (the plan follows what the launches see: six epochs of 8)
        ctx.render(cam, R.RGB_ASCII)
        ctx.synchronize()
    if not stamps:
        ts = []
        for _ in range(3):
            ctx.timer_start()
            for _ in range(50):
                ctx.render(cam, R.RGB_ASCII)
            ts.append(ctx.timer_stop() / 50 * 1e3)
        ctx.set_option(R.OPT_TILE_ORDER, -1)
        for _ in range(100):
            ctx.render(cam, R.RGB_ASCII)
        ctx.synchronize()
        tb = []
        for _ in range(3):
            ctx.timer_start()
            for _ in range(50):
                ctx.render(cam, R.RGB_ASCII)
            tb.append(ctx.timer_stop() / 50 * 1e3)
        print("yaw pi%+.1f: alone %.1f us in frame order, %.1f us balanced  (%s)" % (yaw - math.pi, sorted(ts)[1], sorted(tb)[1], ctx.last_kernel), flush=True)
    else:
        nwg = 16384
        buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        os.environ["RTX_STAMPS_PTR"] = str(buf.data_ptr())
        ctx.render(cam, R.RGB_ASCII)
        ctx.synchronize()
        del os.environ["RTX_STAMPS_PTR"]
        s = buf.cpu().numpy().reshape(nwg, 16)
        s = s[s[:, 15] != 0]
        total = (s[:, 12] >> 32).astype(np.int64)
        life = (s[:, 13] - s[:, 15]) / 100.0
        span = (s[:, 13].max() - s[:, 15].min()) / 100.0
        order = np.argsort(-life)[:5]
        # per-wave features of the longest-lived workgroups (slots 8..11: estimate | slow-path entries << 20 | passes with a hit << 32 |
        # passes << 36 | candidates scanned << 40 | exact tests that changed no lane's best hit << 52)
        feats = []
        for i in order[:2]:
            for w in range(4):
                v = int(s[i, 8 + w]) & 0xFFFFFFFFFFFFFFFF
                feats.append("wave%d: exact-test entries %d (useless %d), candidates scanned %d, passes %d (with a hit %d)" % (
                    w, (v >> 20) & 0xFFF, (v >> 52) & 0xFFF, (v >> 40) & 0xFFF, (v >> 36) & 0xF, (v >> 32) & 0xF))
        print("yaw pi%+.1f: %d workgroups, launch %.1f us; candidates per tile: median %d, p99 %d, max %d; tiles over 192: %d, over 704: %d; "
              "longest lifetimes (us, candidates): %s" % (yaw - math.pi, s.shape[0], span, np.median(total), np.percentile(total, 99), total.max(),
                                                           int((total > 192).sum()), int((total > 704).sum()),
                                                           [(round(float(life[i]), 1), int(total[i])) for i in order]), flush=True)
        if yaw - math.pi > 1.35 and yaw - math.pi < 1.45 or abs(yaw - math.pi) < 0.05:
            print("      " + "\n      ".join(feats), flush=True)
    ctx.close()

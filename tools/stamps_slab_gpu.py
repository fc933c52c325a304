#!/usr/bin/env python3
"""Diagnostic (librtx_hip_ablate.so): phase durations of workgroups that run ALONE on their CU -- one row of macro
tiles of C2 (120 workgroups on 256 CUs), full stamps.  Shows the latency chain of a workgroup without contention."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTX_LIB"] = "librtx_hip_ablate.so"
import torch  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
p, sph, pl = R.config_inputs("C2")
ctx = R.Context(1920, 1080)
ctx.set_scene(sph, pl)
ctx.set_option(R.OPT_TILE_ORDER, 0)
out = torch.zeros(20 * 1920 * 64, dtype=torch.uint8, device="cuda")
for row0 in (0, 512, 1016):
    for _ in range(20):
        ctx.render_rows(p, R.RGB_ASCII, row0, 64, d_out=out.data_ptr(), out_row_base=row0)
    ctx.synchronize()
    buf = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    os.environ["RTX_STAMPS_PTR"] = str(buf.data_ptr())
    ctx.render_rows(p, R.RGB_ASCII, row0, 64, d_out=out.data_ptr(), out_row_base=row0)
    ctx.synchronize()
    del os.environ["RTX_STAMPS_PTR"]
    s = buf.cpu().numpy().reshape(4096, 16)
    s = s[s[:, 0] != 0]
    q = [0, 50, 100]
    total = (s[:, 12] >> 32).astype(int)
    print("rows %d..%d: %d workgroups, candidates per tile median %d" % (row0, row0 + 64, len(s), np.median(total)))
    names = ["tables+frustum (0->1)", "staging (1->2)", "pass 0", "pass 1", "pass 2", "pass 3"]
    for i, nm in enumerate(names):
        d = s[:, i + 1] - s[:, i]
        print("  %-24s cycles min/median/max %s" % (nm, np.percentile(d, q).astype(int)))
    print("  lifetime us (realtime)   %s" % np.round(np.percentile((s[:, 13] - s[:, 15]) / 100.0, q), 2))

#!/usr/bin/env python3
"""Diagnostic (librtx_hip_ablate.so): C2 frame after frame with per-wave light stamps, from a fresh context: how the
spread of the CU groups' finish times develops while rtx_balance_tiles' corrections settle."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTX_LIB"] = "librtx_hip_ablate.so"
os.environ["RTX_ABLATE"] = str(0x8000)
import torch  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
period = int(sys.argv[2]) if len(sys.argv) > 2 else -1
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
R = importlib.import_module("raytracing-in-windows-console_amd")
p, sph, pl = R.config_inputs("C2")
ctx = R.Context(1920, 1080)
ctx.set_scene(sph, pl)
ctx.set_option(R.OPT_TILE_ORDER, period)
buf = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
os.environ["RTX_STAMPS_PTR"] = str(buf.data_ptr())
for f in range(frames):
    buf.zero_()
    torch.cuda.synchronize()
    ctx.render(p, mode)
    ctx.synchronize()
    s = buf.cpu().numpy().reshape(4096, 16)
    s = s[s[:, 15] != 0]
    n = len(s)
    xcc = ((s[:, 14] >> 32) & 0xff).astype(int)
    base = s[:, 15].min()
    st = (s[:, 15] - base) / 100.0
    off = np.array([st[xcc == x].min() if (xcc == x).any() else 0.0 for x in range(8)])
    wend = (s[:, 4:8] - base) / 100.0 - off[xcc][:, None]
    g = np.arange(n) % 256
    fin = np.array([wend[g == i].max() for i in range(256)])
    print("frame %2d: %d workgroups, group finish us min %.2f p10 %.2f median %.2f p90 %.2f max %.2f std %.2f" % (
        f + 1, n, fin.min(), np.percentile(fin, 10), np.median(fin), np.percentile(fin, 90), fin.max(), fin.std()), flush=True)

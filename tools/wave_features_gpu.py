#!/usr/bin/env python3
"""Diagnostic (librtx_hip_ablate.so): what the waves of one trace launch of a BASELINE config did -- passes,
candidates scanned, candidates taken to the exact test, exact tests that changed no lane's best hit, passes that shaded."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTX_LIB"] = "librtx_hip_ablate.so"
os.environ["RTX_ABLATE"] = str(0x8000)
import torch  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
R = importlib.import_module("raytracing-in-windows-console_amd")
p, sph, pl = R.config_inputs(cfg)
ctx = R.Context(int(p.x), int(p.y))
ctx.set_scene(sph, pl)
ctx.set_option(R.OPT_TILE_ORDER, 0)
for _ in range(3):
    ctx.render(p, R.RGB_ASCII)
ctx.synchronize()
nwg = 65536
buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
os.environ["RTX_STAMPS_PTR"] = str(buf.data_ptr())
ctx.render(p, R.RGB_ASCII)
ctx.synchronize()
del os.environ["RTX_STAMPS_PTR"]
s = buf.cpu().numpy().reshape(nwg, 16)
s = s[s[:, 15] != 0]
f = s[:, 8:12].ravel()
passes = ((f >> 36) & 0xf).astype(float)
cand = ((f >> 40) & 0xfff).astype(float)
slow = ((f >> 20) & 0xfff).astype(float)
useless = ((f >> 52) & 0xfff).astype(float)
hit = ((f >> 32) & 0xf).astype(float)
print(cfg, ctx.last_kernel_name() if hasattr(ctx, "last_kernel_name") else "", "workgroups", len(s), "waves", f.size)
tp = passes.sum()
print("per wave-pass: candidates %.2f, exact tests %.2f, of which change nothing %.2f (%.0f %%), passes that shade %.2f" % (
    cand.sum() / tp, slow.sum() / tp, useless.sum() / tp, 100.0 * useless.sum() / max(1.0, slow.sum()), hit.sum() / tp))

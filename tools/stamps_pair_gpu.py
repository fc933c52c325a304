#!/usr/bin/env python3
"""Diagnostic (librtx_hip_ablate.so): the same C2 launch twice with per-wave light stamps -- once as shipped, once with
every workgroup padded with dynamic LDS so that a CU holds only two of them (durations close to a workgroup's own,
uncontended work).  Dumps both stamp arrays (row = linear block id; same tile order in both) for offline analysis:
does the sum of the uncontended durations on a CU predict when that CU finishes?"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTX_LIB"] = "librtx_hip_ablate.so"
os.environ["RTX_ABLATE"] = str(0x8000)
import torch  # noqa: E402

sub = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 44000
R = importlib.import_module("raytracing-in-windows-console_amd")
p, sph, pl = R.config_inputs("C2")
ctx = R.Context(1920, 1080)
ctx.set_scene(sph, pl)
ctx.set_option(R.OPT_SUBTILES, sub)
ctx.set_option(R.OPT_TILE_ORDER, 1 << 20)   # order settled after two frames, never refreshed afterwards
for _ in range(5):
    ctx.render(p, R.RGB_ASCII)
ctx.synchronize()
out = {}
for name, lds in (("shipped", 0), ("padded", pad), ("shipped2", 0)):
    buf = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    os.environ["RTX_STAMPS_PTR"] = str(buf.data_ptr())
    os.environ["RTX_LDS_PAD"] = str(lds)
    ctx.render(p, R.RGB_ASCII)
    ctx.synchronize()
    del os.environ["RTX_STAMPS_PTR"]
    out[name] = buf.cpu().numpy().reshape(4096, 16)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "stamps_pair_sub%d.npz" % sub), **out)
for k, s in out.items():
    s = s[s[:, 15] != 0]
    print(k, "workgroups", len(s), "span us", (s[:, 4:8].max() - s[:, 15].min()) / 100.0)

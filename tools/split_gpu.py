#!/usr/bin/env python3
"""Experiment: one frame as P row-slab launches on P streams, frames strictly one after the other."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
R = importlib.import_module("raytracing-in-windows-console_amd")
p, sph, pl = R.config_inputs("C2")
W, H = 1920, 1080
ctx = R.Context(W, H)
ctx.set_scene(sph, pl)
frame = torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda")
K = 200
for parts in (1, 2, 3, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    bounds = [(H * i // parts) // 32 * 32 if 0 < i < parts else (0 if i == 0 else H) for i in range(parts + 1)]
    torch.cuda.synchronize()
    def one():
        for i in range(parts):
            ctx.render_rows(p, R.RGB_ASCII, bounds[i], bounds[i + 1] - bounds[i], d_out=frame.data_ptr(), out_row_base=0,
                            stream=streams[i].cuda_stream)
        torch.cuda.synchronize()
    for _ in range(10):
        one()
    t0 = time.perf_counter()
    for _ in range(K):
        one()
    dt = time.perf_counter() - t0
    print("frame as %d slab launches on %d streams, host sync per frame: %.2f us/frame" % (parts, parts, dt / K * 1e6))

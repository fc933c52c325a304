#!/usr/bin/env python3
"""Experiment: frames in flight.  K frames of C2 on 1, 2, 3 alternating streams / frame buffers."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
R = importlib.import_module("raytracing-in-windows-console_amd")
p, sph, pl = R.config_inputs("C2")
W, H = 1920, 1080
ctx = R.Context(W, H)
ctx.set_scene(sph, pl)
K = 300
for nfl in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    frames = [torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda") for _ in range(nfl)]
    torch.cuda.synchronize()
    for i in range(20):
        ctx.render_rows(p, R.RGB_ASCII, 0, H, d_out=frames[i % nfl].data_ptr(), stream=streams[i % nfl].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        ctx.render_rows(p, R.RGB_ASCII, 0, H, d_out=frames[i % nfl].data_ptr(), stream=streams[i % nfl].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("frames in flight %d: %.2f us/frame, %.1f Grays/s" % (nfl, dt / K * 1e6, (W - 1) * H * K / dt / 1e9))

# host-side enqueue cost of one rtx_render_rows call from Python (launch queue not full)
streams = [torch.cuda.Stream() for _ in range(4)]
frames = [torch.zeros(20 * W * H, dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    ctx.render_rows(p, R.RGB_ASCII, 0, H, d_out=frames[i % 4].data_ptr(), stream=streams[i % 4].cuda_stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue of 40 launches: %.2f us each; drain %.2f us" % ((t1 - t0) / 40 * 1e6, (t2 - t1) * 1e6))

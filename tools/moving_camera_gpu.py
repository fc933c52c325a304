#!/usr/bin/env python3
"""C2's scene with a camera that moves every frame (yaw and position change by a fixed step per frame): time per frame,
one launch at a time and 4 frames in flight, with the dispatch order on auto and off -- does the balancing, which
learns from earlier frames, hold up when no two frames are the same?"""
import importlib
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
p0, sph, pl = R.config_inputs("C2")
W, H = int(p0.x), int(p0.y)
N = 2400
for step in [float(v) for v in (sys.argv[1:] or ["0", "0.0001", "0.001", "0.01"])]:
    cams = [R.camera_params(W, H, (3.0 * math.sin(i * step * 3.0), 0.0, 10.0 * i * step), (0.0, math.pi + i * step, 0.0)) for i in range(N)]
    for order in (-1, 0):
        ctx = R.Context(W, H)
        ctx.set_scene(sph, pl)
        ctx.set_option(R.OPT_TILE_ORDER, order)
        for c in cams[:400]:
            ctx.render(c, R.RGB_ASCII)
        ctx.synchronize()
        ctx.timer_start()
        for c in cams[400:]:
            ctx.render(c, R.RGB_ASCII)
        alone = ctx.timer_stop() / (N - 400) * 1e3
        streams = [torch.cuda.Stream() for _ in range(4)]
        bufs = [torch.empty(20 * W * H, dtype=torch.uint8, device="cuda") for _ in range(4)]
        for i, c in enumerate(cams[:400]):
            ctx.render_rows(c, R.RGB_ASCII, 0, H, d_out=bufs[i % 4].data_ptr(), out_row_base=0, stream=streams[i % 4].cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        for i, c in enumerate(cams[400:]):
            ctx.render_rows(c, R.RGB_ASCII, 0, H, d_out=bufs[i % 4].data_ptr(), out_row_base=0, stream=streams[i % 4].cuda_stream)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        e1.record()
        e1.synchronize()
        print("step %g rad/frame, tile order %2d: alone %.2f us/frame, 4 in flight %.2f us/frame" % (
            step, order, alone, e0.elapsed_time(e1) / (N - 400) * 1e3), flush=True)
        ctx.close()

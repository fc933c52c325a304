#!/usr/bin/env python3
"""One rank's slab launches of a row-sharded 1080p frame (C2): time per launch for 1080 / 540 / 270 / 135 rows
(1 / 2 / 4 / 8 GPUs) by sub-tiles per workgroup, one launch at a time and two streams at a time, compact words."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
p, sph, pl = R.config_inputs("C2")
W, H = int(p.x), int(p.y)
ctx = R.Context(W, H)
ctx.set_scene(sph, pl)
buf = torch.empty(4 * W * H, dtype=torch.uint8, device="cuda")
streams = [torch.cuda.Stream() for _ in range(2)]
for n in [int(v) for v in os.environ.get('SLAB_N', '1,2,4,8').split(',')]:
    rows = H // n
    for sub in [int(v) for v in os.environ.get('SLAB_SUB', '0,1,2,4').split(',')]:
        ctx.set_option(R.OPT_SUBTILES, sub)
        res = []
        for rank in (0, n // 2, n - 1) if n > 1 else (0,):
            row0 = rank * rows
            def launch(stream=None):
                ctx.render_rows(p, R.RGB_ASCII, row0, rows, d_out=buf.data_ptr(), out_row_base=0, stream=stream, flags=R.RENDER_COMPACT)
            for _ in range(300):
                launch()
            ctx.synchronize()
            ctx.timer_start()
            for _ in range(500):
                launch()
            alone = ctx.timer_stop() / 500 * 1e3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(150):
                for st in streams:
                    launch(st.cuda_stream)
            torch.cuda.synchronize()
            e0.record()
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
            for _ in range(250):
                for st in streams:
                    launch(st.cuda_stream)
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
            e1.record()
            e1.synchronize()
            res.append((rank, alone, e0.elapsed_time(e1) / 500 * 1e3))
        print("N=%d rows %4d sub-tiles %d (%s): " % (n, rows, sub, ctx.last_kernel) +
              "  ".join("rank %d alone %.2f us, 2 streams %.2f us" % r for r in res), flush=True)

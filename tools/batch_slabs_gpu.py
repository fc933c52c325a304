#!/usr/bin/env python3
"""A rank's slab of every frame of a round, one launch per slab-frame on F streams (rounds 1-3) against ONE batched launch
(RTX_OPT_BATCH, rtx_trace_batch): C2 at N = 8 / 4 / 2 (135 / 270 / 540 rows), M = 8 frames, records and compact words, by
sub-tile count.  Prints microseconds per M slab-frames (HIP events, median of repeats).

  python tools/batch_slabs_gpu.py [config] [M]
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

R = importlib.import_module("raytracing-in-windows-console_amd")
config = sys.argv[1] if len(sys.argv) > 1 else "C2"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
p, sph, pl = R.config_inputs(config)
W, H = int(p.x), int(p.y)


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def timed(fn, sync, on, reps=30, inner=20):
    """microseconds per call of fn, HIP events on stream `on` (the stream the call's work ends on), median of reps"""
    for _ in range(200):
        fn()
    sync()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(on)
        for _ in range(inner):
            fn()
        e1.record(on)
        e1.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / inner)
    return med(out)


main = torch.cuda.Stream()
for ranks in (8, 4, 2, 1):
    rank = ranks // 2
    row0, rows = H * rank // ranks, H * (rank + 1) // ranks - H * rank // ranks
    for compact in (True, False):
        S = 4 if compact else 20
        bufs = [torch.empty(rows * W * S, dtype=torch.uint8, device="cuda") for _ in range(M)]
        line = "%s N=%d rows %4d M=%d %-7s:" % (config, ranks, rows, M, "compact" if compact else "records")
        for sub in (0, 1, 2, 3, 4, 5, 6, 8):
            with R.Context(W, H) as c:
                c.set_scene(sph, pl)
                c.set_option(R.OPT_SUBTILES, sub)
                flags = R.RENDER_COMPACT if compact else 0
                if sub == 0:
                    for F in (4, 2):
                        rs = [torch.cuda.Stream() for _ in range(F)]
                        c.set_option(R.OPT_BATCH, 0)
                        one = c.make_slab_submitter(p, R.RGB_ASCII, row0, rows, row0, [b.data_ptr() for b in bufs], [rs[i % F].cuda_stream for i in range(M)],
                                                    main.cuda_stream, flags=flags)
                        line += "  per-slab launches on %d streams %6.1f us |" % (F, timed(one, torch.cuda.synchronize, main))
                    c.set_option(R.OPT_BATCH, -1)
                # the batched launch alone on its stream, launch after launch (no fork / join around it: `after` = None), and -- for
                # the default plan -- forked from / joined into another stream as the sharded loop does
                st = torch.cuda.Stream()
                bat = c.make_slab_submitter(p, R.RGB_ASCII, row0, rows, row0, [b.data_ptr() for b in bufs], [st.cuda_stream] * M, None, flags=flags)
                t = timed(bat, torch.cuda.synchronize, st)
                assert c.get_option(R.STAT_BATCHED_LAUNCHES) > 0
                line += "  batch sub=%d %6.1f" % (sub, t)
                if sub == 0:
                    fj = c.make_slab_submitter(p, R.RGB_ASCII, row0, rows, row0, [b.data_ptr() for b in bufs], [st.cuda_stream] * M, main.cuda_stream, flags=flags)
                    line += " (forked/joined %6.1f" % timed(fj, torch.cuda.synchronize, main)
                    c.set_option(R.OPT_TILE_ORDER, 0)
                    line += ", frame order %6.1f)" % timed(bat, torch.cuda.synchronize, st)
        print(line, flush=True)

#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary rays) of the ray-trace hot path on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1: launched by torch.distributed.run, one rank per GPU, backend nccl (= RCCL).

Workload: BASELINE.json configs[1] -- 1920x1080, 1024 spheres + 1 plane (SURVEY.md App. D scene,
LCG seed 2), mode RGB_ASCII (20-byte records).  A step is one frame: primary-ray generation,
closest hit over the scene, shading and the ANSI record write for every pixel, complete character
buffer resident in HBM at the end.  rays per frame = (W-1)*H (RayTracing.cu:187).

N = 1: each frame is rendered by one launch; by default 4 frames are in flight on 4 HIP streams (frame i in
       frame buffer i % 4), because consecutive frames are independent and the drain of one launch overlaps
       the ramp of the next.  --frames-in-flight 1 renders strictly one after the other.
N > 1: every frame is sharded by pixel rows (rank g renders rows [g*H/N, (g+1)*H/N) with the global
       row index in ray generation) and assembled, complete, in the HBM of its root GPU.  The root rotates
       (frame i on rank i % N), so every directed xGMI link carries one slab per N frames.  Default
       (--exchange rounds): frames go in rounds of N; a rank queues its N slab launches of a round on
       several streams and ONE RCCL all-to-all per round delivers slab j of every rank to rank j in row
       order.  --exchange p2p: one point-to-point gather per frame (--root fixed pins the root to rank 0).
       Total work is fixed as N grows ("strong").  Buffers are rings, so the exchange of a round overlaps
       the trace of the next; all K frames are complete inside the timed region.

Timing (N = 1): after the run-in and the W warm-up steps, a batch of exactly K steps is timed with HIP events
(recorded on the stream(s) the kernels are launched on: every render stream stamps the start of its first frame of
the batch and the end of its last one; the batch took from the earliest start to the latest end) and bracketed by
synchronisation on both sides; the batch is repeated until at least --min-timed-ms of GPU time has been measured and the MEDIAN batch
is reported ("timing" says how many repeats, their spread, and the wall-clock figure beside it).  The last frame
of the last batch is then compared (SHA-256 of the whole 20*W*H buffer) with the committed golden value,
outside the timed region: "verified_against_golden".

Also reported: "roofline" (algorithmic HBM bytes / measured kernel time vs 8 TB/s, plus the executed-VALU
utilisation from the committed rocprofv3 counters, since this path is VALU-issue bound, not HBM bound) and
"cpu_baseline" (the CPU oracle -- a structure-faithful port of the reference's loop -- timed on this host's
cores on full frames of the same workload).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

BASELINE_METRIC = "Mrays/s (primary rays) at 1920\u00d71080, 1024 spheres; 1/2/4/8 GPU"   # BASELINE.json "metric", verbatim
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak counts FMA as 2; this path may not contract, so 78.65 Tops/s
VALU_PEAK_TLANEOPS = VALU_PEAK_TFLOPS / 2.0   # lane-operations per second when no instruction is an FMA


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def committed_counters(config, mode_name, kernel):
    """Per-launch rocprofv3 PMC averages of this kernel from profiles/counters.json (written by
    tools/make_counters.py from the committed *_summary.json of the same bench command); None when absent."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        with open(path) as f:
            return json.load(f).get("%s_%s_%s" % (config, mode_name, kernel))
    except (OSError, ValueError):
        return None


def algorithmic_bytes(W, H, S, ns, npl, rows=None):
    """records written + scene payload read once + params payload."""
    rows = H if rows is None else rows
    return (W - 1) * rows * S + 28 * ns + 44 * npl + 88


def algorithmic_flops(W, H, ns, npl, hit_frac):
    """19*Ns + 7*Np + 30 (ray-gen) + 150*[hit] per ray."""
    return (W - 1) * H * (19.0 * ns + 7.0 * npl + 30.0 + 150.0 * hit_frac)


def _frame_matches_golden(frame, config, mode_name):
    """SHA-256 of the whole frame buffer against tests/golden/golden.json (oracle-generated; no oracle needed here)."""
    import hashlib
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        g = json.load(f).get("%s_%s" % (config, mode_name), {})
    want = g.get("frame_sha256")
    if not want:
        return None   # no golden frame for this config / mode
    return hashlib.sha256(frame.tobytes()).hexdigest() == want


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="C2", help="BASELINE config name (C1..C5); the graded line uses C2")
    ap.add_argument("--mode", default="RGB_ASCII")
    ap.add_argument("--kernel", default="auto", choices=["auto", "brute", "binned"])
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--subtiles", type=int, default=0)
    ap.add_argument("--two-level", type=int, default=-1, help="-1 auto, 0 off, 1 on, 2 on with block binning")
    ap.add_argument("--refine", type=int, default=-1)
    ap.add_argument("--tile-order", type=int, default=-2, help="heaviest-first dispatch of the macro tiles: -2 library default (auto), -1 auto, 0 off, k refresh period")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--verify", action="store_true", help="(default) check the last frame against the golden hash, outside the timed region")
    ap.add_argument("--no-verify", action="store_true", help="skip the byte check of the last frame")
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="N=1: the K-step batch is repeated until this much GPU time has been measured; the median batch is reported")
    ap.add_argument("--max-repeats", type=int, default=2000)
    ap.add_argument("--root", default="rotate", choices=["rotate", "fixed"],
                    help="N>1: frame i is assembled on rank i %% N (rotate: highest throughput) or always on rank 0, in frame order "
                         "(fixed: north_star's literal gather, in-order delivery to one consumer)")
    ap.add_argument("--exchange", default="compact", choices=["compact", "rounds", "p2p"],
                    help="N>1: compact (default) = frames in rounds of M*N with rotating roots, one RCCL all-to-all per "
                         "round, slabs as 4-byte pixel words expanded into records on the root; rounds = the same with "
                         "the slabs as records (M=1); p2p = one point-to-point gather per frame (see --root)")
    ap.add_argument("--frames-per-root", type=int, default=0, help="N>1, --exchange compact: M (0 = 8/4/4 for 2/4/8 GPUs; 8 with --root fixed)")
    ap.add_argument("--graphs", type=int, default=1, help="N>1, --exchange compact: record a round's slab launches (and its expansions) as HIP graphs (0 = off)")
    ap.add_argument("--latency", action="store_true", help="N>1, --exchange compact: also stamp every frame's completion and report queue-to-complete latency")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="N=1: frames rendered concurrently on separate HIP streams into separate frame buffers (default 4; 6 from 16384 spheres; "
                         "1 = strictly one launch after the other, the form the rocprof summaries are taken in).  "
                         "N>1: render streams the slab launches of a round are spread over (default 2 for N<=2, else 4)")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed, uncounted run-in before the warm-up steps so that the GPU is at its running clocks "
                         "(N>1: the equivalent number of frames, fixed so that all ranks agree)")
    ap.add_argument("--what", default="trace", choices=["trace", "update", "update-async"],
                    help="trace: the graded step (frame resident in HBM). update: whole RayTracingManager::Update "
                         "(trace + GPU minimise + copy of the minimised stream to the host), N=1 only")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    # RTX_BENCH_FORCE_DIST=1 under torch.distributed.run with one rank walks the N>1 code (RCCL init, rings,
    # all-reduce of the time) on a one-GPU box; the numbers it prints are not a bench line
    distributed = world > 1 or os.environ.get("RTX_BENCH_FORCE_DIST") == "1"
    if distributed and world != n_gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (n_gpus, world))
    if not distributed and n_gpus != 1:
        raise SystemExit("--gpus %d needs torch.distributed.run (one rank per GPU)" % n_gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-trace path has no CPU fallback")

    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    R = importlib.import_module("raytracing-in-windows-console_amd")
    sharding = importlib.import_module("raytracing-in-windows-console_amd.sharding")
    mode = R.MODE_NAMES.index(args.mode)
    S = R.SIZE_RGB if mode >= R.RGB_ASCII else R.SIZE_8BIT
    W, H, ns, npl, seed = R.CONFIGS[args.config]
    params, sph, pl = R.config_inputs(args.config)

    ctx = R.Context(W, H, device=local_rank)
    ctx.set_scene(sph, pl)
    ctx.set_option(R.OPT_KERNEL, {"auto": R.KERNEL_AUTO, "brute": R.KERNEL_BRUTE, "binned": R.KERNEL_BINNED}[args.kernel])
    ctx.set_option(R.OPT_TILE_LOG2_W, args.tile)
    ctx.set_option(R.OPT_SUBTILES, args.subtiles)
    ctx.set_option(R.OPT_TWO_LEVEL, args.two_level)
    ctx.set_option(R.OPT_REFINE, args.refine)
    if args.tile_order >= -1:
        ctx.set_option(R.OPT_TILE_ORDER, args.tile_order)

    ctx.render_rows(params, mode, 0, 1)   # uploads the scene (a HIP graph capture later on must not have to)
    ctx.synchronize()
    if args.frames_in_flight <= 0:
        # N=1: 4 frames in flight; 6 where a frame is two dependent launches (the coarse-cell pre-pass of large scenes, then
        # the trace): config 5 44.7 -> 42.7 us per frame, the other configs the same with 4 and 6
        args.frames_in_flight = ((6 if len(sph) >= 16384 else 4) if not distributed else (4 if world > 2 else 2))
    K, Wm = args.steps, args.warmup
    bounds = sharding.row_bounds(H, world)
    row0, rows = bounds[rank], bounds[rank + 1] - bounds[rank]
    frame_bytes = 20 * W * H

    kernel_ms = None
    if not distributed:
        F = max(1, args.frames_in_flight)
        if args.what == "update":
            F = 1

            def step(i):
                ctx.update(params, mode)
        elif args.what == "update-async":
            F = 1
            hb = [ctx.host_alloc(frame_bytes) for _ in range(2)]
            inflight = []

            def step(i):
                if len(inflight) == 2:
                    ctx.update_end(inflight.pop(0))
                inflight.append(ctx.update_begin(params, mode, hb[i % 2][0]))
        elif F == 1:
            def step(i):
                ctx.render(params, mode)
        else:
            # F frames in flight: frame i goes to stream i % F and frame buffer i % F.  Consecutive frames do
            # not depend on each other, so the tail of one frame's launch (few workgroups left, SIMDs
            # under-occupied) overlaps with the start of the next ones.  Every frame is rendered in full.
            streams = [torch.cuda.Stream() for _ in range(F)]
            fbufs = [torch.zeros(frame_bytes, dtype=torch.uint8, device="cuda") for _ in range(F)]
            torch.cuda.synchronize()

            submit = ctx.make_submitter(params, mode, [b.data_ptr() for b in fbufs], [st.cuda_stream for st in streams])

            def step(i):
                # one frame per step; rtx_submit_frames queues it on stream i % F (one host call per frame)
                submit(1, i % F)

        # Clock ramp: an idle MI355X needs some tens of milliseconds of work to reach its running clocks (measured:
        # 25.3 us per frame when timing starts 5 ms after idle, 20.2 us in steady state).  So the GPU is first kept
        # busy with the same frames for --prewarm-ms; none of this is timed or counted.
        def drain():
            if args.what == "update-async":
                while inflight:
                    ctx.update_end(inflight.pop(0))
            ctx.synchronize()
            torch.cuda.synchronize()

        n_pre = 0
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            for _ in range(64):
                step(n_pre)
                n_pre += 1
            drain()
        for i in range(Wm):
            step(i)
        drain()

        # One timed batch = exactly K steps, device-timed with HIP events on the streams the kernels run on and
        # bracketed by synchronisation on both sides.
        if args.what != "trace":
            def timed_batch():       # the Update forms block on the host: wall clock is the measurement
                t0 = time.perf_counter()
                for i in range(K):
                    step(i)
                drain()
                dt = (time.perf_counter() - t0) * 1e3
                return dt, dt
        elif F == 1:
            def timed_batch():
                t0 = time.perf_counter()
                ctx.timer_start()                     # hipEventRecord on the context's stream
                for i in range(K):
                    step(i)
                ev_ms = ctx.timer_stop()              # hipEventRecord + hipEventSynchronize + elapsed
                torch.cuda.synchronize()
                return ev_ms, (time.perf_counter() - t0) * 1e3
        else:
            # Each render stream stamps the start of its first frame of the batch and the end of its last one; the batch
            # took from the earliest start to the latest end (hipEventElapsedTime between events of different streams).
            # No stream waits for another at either end, so the window holds the K frames' launches and nothing else.
            ev_s = [torch.cuda.Event(enable_timing=True) for _ in range(F)]
            ev_e = [torch.cuda.Event(enable_timing=True) for _ in range(F)]

            def timed_batch():
                t0 = time.perf_counter()
                used = min(F, K)
                for i in range(K):
                    if i < F:
                        ev_s[i].record(streams[i])
                    step(i)
                for j in range(used):
                    ev_e[j].record(streams[j])
                for j in range(used):
                    ev_e[j].synchronize()
                torch.cuda.synchronize()
                ev_ms = max(ev_s[a].elapsed_time(ev_e[b]) for a in range(used) for b in range(used))
                return ev_ms, (time.perf_counter() - t0) * 1e3

        ev_batches, wall_batches = [], []
        while True:
            e_ms, w_ms = timed_batch()
            ev_batches.append(e_ms)
            wall_batches.append(w_ms)
            if len(ev_batches) >= args.max_repeats or (sum(ev_batches) >= args.min_timed_ms and len(ev_batches) >= 3):
                break
        batch_ms = median(ev_batches)
        elapsed = batch_ms * 1e-3                     # seconds per K steps, the median batch
        timing = {"method": "HIP events on the render streams around each batch of K steps (earliest first-frame start to latest "
                            "last-frame end), sync on both sides; median over repeated batches" if args.what == "trace" else "wall clock around each batch of K blocking steps; median",
                  "repeats": len(ev_batches), "timed_ms_total": round(sum(ev_batches), 3),
                  "batch_ms": {"median": round(batch_ms, 5), "min": round(min(ev_batches), 5), "max": round(max(ev_batches), 5)},
                  "wall_ms_per_step_median": round(median(wall_batches) / K, 5)}
        if not args.no_verify and args.what == "trace":
            final = ctx.read_frame(frame_bytes) if F == 1 else fbufs[(K - 1) % F].cpu().numpy()
        else:
            final = None
        # The dominant kernel on its own: launches one after the other on the context's stream, timed with
        # HIP events on that stream (this is what a rocprofv3 kernel trace of --frames-in-flight 1 shows).
        if args.what != "trace":
            kernel_ms = elapsed / K * 1e3
        else:
            Kr = max(10, min(K, 100))
            # (the context's own stream has its own dispatch order: let it settle as the render streams' did --
            # the library balances a tile grid over its first launches on a stream, and keeps refining every 64th)
            for _ in range(5 if F == 1 else 1000):
                ctx.render(params, mode)
            ctx.synchronize()
            singles = []
            for _ in range(5):
                ctx.timer_start()
                for _ in range(Kr):
                    ctx.render(params, mode)
                singles.append(ctx.timer_stop() / Kr)
            kernel_ms = median(singles)
    else:
        import torch.distributed as dist
        # The loop runs on a stream of its own, made torch's current stream: the collectives are ordered after it, and
        # its handle is not 0.  (torch's default stream has handle 0, which the C ABI reads as "the context's own
        # stream" / "no stream to join": with it neither rtx_submit_slabs' fork/join nor a graph capture would touch
        # the stream the exchange is queued on.  Round 1's loop did exactly that; its frames only looked right because
        # every frame of a bench run is the same frame.  The check rounds below would now catch it.)
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        assert stream.cuda_stream != 0
        if args.exchange in ("compact", "rounds"):
            # Frames in rounds of M*N, frame m*N+j of a round assembled on rank j, ONE all-to-all per round.  The
            # slab launches of a round go to F streams forked from / joined into torch's current stream (inside
            # rtx_submit_slabs), which the RCCL call is ordered after.  "compact": the slabs travel as 4-byte
            # pixel words and the root expands them into records on a side stream once the exchange is done.
            compact = args.exchange == "compact"
            fixed_root = args.root == "fixed"      # in-order delivery: every frame assembled on rank 0, in frame order
            roots = [0] if fixed_root else None
            M = 1 if not compact else (args.frames_per_root or ({2: 8, 4: 8, 8: 8} if fixed_root else {2: 8, 4: 4, 8: 4}).get(world, max(1, 16 // world)))
            post = torch.cuda.Stream()
            expanders = {}
            use_graphs = [compact and args.graphs != 0]
            slab_graphs, expand_graphs = {}, {}
            full_mine = [None]
            # per-frame latency: an event when a round's slab launches are queued, one when each of its frames is
            # complete on its root (pools of timing events, reused round-robin; read after the timed region)
            POOL = 64
            ev_submit = [torch.cuda.Event(enable_timing=True) for _ in range(POOL)]
            ev_done = [[torch.cuda.Event(enable_timing=True) for _ in range(max(1, M))] for _ in range(POOL)]
            done_count = {}

            class _After:
                def __init__(self, ev):
                    self.ev = ev

                def wait(self):
                    torch.cuda.current_stream().wait_event(self.ev)

            def graph_or_none(build, on_stream, what):
                """Records what build() queues on `on_stream` as a HIP graph; None (and no more attempts) if that fails."""
                if not use_graphs[0]:
                    return None
                try:
                    ctx.graph_begin(on_stream)
                    try:
                        build()
                    finally:
                        g = ctx.graph_end(on_stream)
                    return ctx.graph_launcher(g, on_stream)
                except R.RtxError as exc:
                    sys.stderr.write("bench.py: HIP graph capture of the %s failed (%s); queueing launch by launch\n" % (what, exc))
                    use_graphs[0] = False
                    return None

            def finish(q, b, work, mine):
                with torch.cuda.stream(post):
                    work.wait()   # orders `post` (only) after the exchange
                    key = (b, len(mine))
                    for m, segs in mine:
                        ek = (b, m, len(mine))
                        if ek not in expanders:
                            expanders[ek] = ctx.make_expander(mode, pipe.recv[b].data_ptr(), pipe.frames[b][m].data_ptr(), segs, post.cuda_stream)
                    if args.latency:
                        # frame by frame, so that every frame's completion can be stamped (frames leave in frame order)
                        for m, _ in mine:
                            expanders[(b, m, len(mine))]()
                            ev_done[q % POOL][m].record(post)
                        done_count[q % POOL] = (q, len(mine))
                    else:
                        # a graph only for full rounds (the first round, untimed, is one: it fixes how many frames of a
                        # full round are this rank's); a partial round -- K is not a multiple of the round -- is queued
                        # launch by launch, so that nothing is recorded inside the timed region
                        if full_mine[0] is None:
                            full_mine[0] = len(mine)
                        if mine and len(mine) == full_mine[0] and key not in expand_graphs:
                            expand_graphs[key] = graph_or_none(lambda: [expanders[(b, m, len(mine))]() for m, _ in mine], post.cuda_stream, "expansions")
                        if mine and expand_graphs.get(key) is not None:
                            expand_graphs[key]()
                        else:
                            for m, _ in mine:
                                expanders[(b, m, len(mine))]()
                    ev = torch.cuda.Event()
                    ev.record(post)
                return _After(ev)

            pipe = sharding.RowShardedRounds(dist, torch, rank, world, W, H, S, "cuda", nbuf=2, frames_per_root=M,
                                             pixel_bytes=4 if compact else None, finish=finish if compact else None, roots=roots)
            F = max(1, args.frames_in_flight)
            rstreams = [torch.cuda.Stream() for _ in range(F)]
            torch.cuda.synchronize()
            RF = pipe.round_frames
            submitters = [ctx.make_slab_submitter(params, mode, row0, rows, row0,
                                                  [pipe.unit(b, f).data_ptr() if rows else pipe.send[b].data_ptr() for f in range(RF)],
                                                  [rstreams[f % F].cuda_stream for f in range(RF)], stream.cuda_stream,
                                                  flags=R.RENDER_COMPACT if compact else 0)
                          for b in range(pipe.nbuf)]

            def render_round(q, b, nframes):
                if args.latency:
                    ev_submit[q % POOL].record(stream)
                if not rows:
                    return
                if nframes == RF and use_graphs[0]:
                    # a full round's slab launches (forked over the render streams, joined back) as one graph replay
                    if b not in slab_graphs:
                        slab_graphs[b] = graph_or_none(lambda: submitters[b](RF), stream.cuda_stream, "slab launches")
                    if slab_graphs.get(b) is not None:
                        slab_graphs[b]()
                        return
                submitters[b](nframes)

            elapsed, q0 = sharding.timed_rounds(dist, torch, pipe, render_round, K, Wm, "cuda", torch.cuda.synchronize,
                                                prewarm=int(args.prewarm_ms * 40),   # ~25 us per frame on one GPU
                                                min_total=args.min_timed_ms * 1e-3, max_repeats=min(args.max_repeats, 200))
            n_rounds = -(-K // RF)
            last_frame = (q0 + n_rounds - 1) * RF + (K - 1 - (n_rounds - 1) * RF)
            slab0 = pipe.unit(0, 0)
            latency = None
            if args.latency and compact:
                # queued -> complete on the root, for the frames of the last rounds still in the event pools
                lat = []
                for slot, (q, n) in done_count.items():
                    if q0 <= q < q0 + n_rounds:
                        for m in range(n):
                            lat.append(ev_submit[slot].elapsed_time(ev_done[slot][m]))
                t = torch.tensor([median(lat) if lat else -1.0, max(lat) if lat else -1.0], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                latency = {"median_ms": round(float(t[0].item()), 4), "max_ms": round(float(t[1].item()), 4),
                           "of": "a frame, from the queueing of its round's slab launches to the complete frame on its root (HIP events; "
                                 "worst rank); a round is %d frames" % RF}
            exchange_note = ("frames in rounds of %d (M=%d per root), %s by one RCCL all-to-all per round; "
                             % (RF, M, "every frame assembled on rank 0 in frame order (in-order delivery)" if fixed_root
                                else "frame i of a round assembled on rank i %% N")) + \
                            ("slabs travel as 4-byte pixel words, the root expands them into records (rtx_expand)" if compact
                             else "slabs travel as records") + ("; HIP graph per round" if use_graphs[0] and slab_graphs else "")
        else:
            pipe = sharding.RowShardedFrames(dist, torch, rank, world, W, H, S, "cuda", nbuf=2, rotate_root=(args.root == "rotate"))

            def render(buf, r0, nrows, base):
                # root traces its own rows straight into the frame; peers into their slab.  Same stream as the
                # RCCL transfers are ordered after.
                ctx.render_rows(params, mode, r0, nrows, d_out=buf.data_ptr(), out_row_base=base, stream=stream.cuda_stream)

            elapsed = sharding.timed_frames(dist, torch, pipe, render, K, Wm, "cuda", torch.cuda.synchronize,
                                            prewarm=int(args.prewarm_ms * 40),
                                            min_total=args.min_timed_ms * 1e-3, max_repeats=min(args.max_repeats, 200))
            last_frame = K - 1
            slab0 = pipe.slabs[0] if pipe.slabs is not None else None
            exchange_note = "RCCL p2p gather per frame; frame i assembled on rank %s" % ("i % N" if args.root == "rotate" else "0")
        final = None
        wins = getattr(pipe, "timed_windows", [elapsed])
        timing = {"method": "wall clock around exactly K frames, barrier + synchronize on both sides, MAX over ranks; the window is repeated "
                            "(each time from a drained pipeline) and the median window reported",
                  "repeats": len(wins), "timed_ms_total": round(sum(wins) * 1e3, 3),
                  "batch_ms": {"median": round(elapsed * 1e3, 5), "min": round(min(wins) * 1e3, 5), "max": round(max(wins) * 1e3, 5)}}
        dist_verified = None
        if not args.no_verify:
            # byte check of the last assembled frame, outside the timed region: its root hashes it (SHA-256 of the
            # whole 20*W*H buffer against the committed golden value); the verdict is reduced to rank 0 as a
            # tri-state: 1 = compared and equal, 0 = compared and different (or the check itself failed),
            # 2 = no golden value for this config / mode, nothing compared
            code = 2
            try:
                if rank == pipe.root_of(last_frame):
                    m = _frame_matches_golden(pipe.frame(last_frame).cpu().numpy(), args.config, args.mode)
                    code = 2 if m is None else int(bool(m))
            except Exception as exc:   # a failed check must not lose the measurement, but it must be seen
                sys.stderr.write("bench.py: golden check of frame %d failed on rank %d: %r\n" % (last_frame, rank, exc))
                code = 0
            flag = torch.tensor([code], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            dist_verified = {0: False, 1: True}.get(int(flag.item()))   # None: no rank compared anything
        # ---- check rounds, outside the timed region: is every frame that arrives on a root the frame its camera
        # renders?  (a) the send buffers are poisoned and one more round goes through the very path that was timed
        # (graph replays included): a slab that was not rendered again, or exchanged before it was rendered, shows as
        # poison; (b) a round in which every frame has its own camera, queued launch by launch: a slab that lands in the
        # wrong frame or on the wrong root shows.  Each root compares its frames on the GPU with the whole frame it
        # renders itself for the same camera.
        rounds_ok = None
        if args.exchange in ("compact", "rounds") and not args.no_verify:
            code = 1
            try:
                qn = q0 + n_rounds
                tmp = torch.empty(20 * W * H, dtype=torch.uint8, device="cuda")

                def frame_is(i, p_i):
                    ctx.render_rows(p_i, mode, 0, H, d_out=tmp.data_ptr(), out_row_base=0, stream=stream.cuda_stream,
                                    flags=0 if mode >= R.RGB_ASCII else R.RENDER_ZERO_TAIL)
                    torch.cuda.synchronize()
                    return bool(torch.equal(pipe.frame(i)[:S * W * H], tmp[:S * W * H]))

                for b in range(pipe.nbuf):
                    pipe.send[b].fill_(0xEE)
                for r in range(pipe.nbuf):
                    pipe.round(qn + r, RF, render_round)
                pipe.drain()
                torch.cuda.synchronize()
                for r in range(pipe.nbuf):
                    for f in range(RF):
                        i = (qn + r) * RF + f
                        if pipe.root_of(i) == rank and not frame_is(i, params):
                            sys.stderr.write("bench.py: rank %d: frame %d of a poisoned round differs from the frame rendered in one piece\n" % (rank, i))
                            code = 0
                qn += pipe.nbuf
                cams = [R.camera_params(W, H, pos=(0.03 * f, 0.01 * f, 0.0), rot=(0.0, float(np.float32(np.pi)) + 0.002 * f, 0.0)) for f in range(RF)]

                def render_round_cams(q, b, nframes):
                    if rows:
                        ctx.submit_slabs(cams[:nframes], mode, row0, rows, [pipe.unit(b, f).data_ptr() for f in range(nframes)], row0,
                                         [rstreams[f % F].cuda_stream for f in range(nframes)], after=stream.cuda_stream,
                                         flags=R.RENDER_COMPACT if compact else 0)

                pipe.round(qn, RF, render_round_cams)
                pipe.drain()
                torch.cuda.synchronize()
                for f in range(RF):
                    i = qn * RF + f
                    if pipe.root_of(i) == rank and not frame_is(i, cams[f]):
                        sys.stderr.write("bench.py: rank %d: frame %d (own camera) differs from the frame rendered in one piece\n" % (rank, i))
                        code = 0
            except Exception as exc:
                sys.stderr.write("bench.py: check rounds failed on rank %d: %r\n" % (rank, exc))
                code = 0
            flag = torch.tensor([code], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            rounds_ok = bool(flag.item())
            if rounds_ok is False:
                dist_verified = False
            timing["check_rounds"] = {"passed": rounds_ok,
                                      "what": "after the timed region: %d round(s) through the timed path with poisoned send buffers, one round with a camera "
                                              "per frame; every root compares its frames with the same frames rendered in one piece" % pipe.nbuf}
        if args.exchange in ("compact", "rounds") and latency is not None:
            timing["frame_latency"] = latency
        # per-rank kernel time, measured apart from the pipeline, for the roofline object
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(min(K, 50)):
            tgt = slab0 if slab0 is not None else pipe.frames[0]
            ctx.render_rows(params, mode, row0, rows, d_out=tgt.data_ptr(), out_row_base=row0 if slab0 is not None else 0,
                            flags=R.RENDER_COMPACT if args.exchange == "compact" else 0)
        kernel_ms = ctx.timer_stop() / min(K, 50)

    rays_per_frame = (W - 1) * H
    mrays = rays_per_frame * K / elapsed / 1e6

    out = None
    if rank == 0:
        gold = {}
        try:
            with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
                gold = json.load(f)
        except OSError:
            pass
        g = gold.get("%s_%s" % (args.config, args.mode), {})
        hit_frac = (g.get("foreground_pixels") or 0) / float(rays_per_frame)
        verified = dist_verified if distributed else None
        if final is not None:
            verified = _frame_matches_golden(final, args.config, args.mode)

        my_rows = rows
        # the trace kernel of a rank writes S bytes per pixel, or 4 when the slabs travel as pixel words
        bytes_alg = algorithmic_bytes(W, H, 4 if (distributed and args.exchange == "compact") else S, ns, npl, my_rows)
        achieved_gbs = bytes_alg / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters of the committed profile of this same kernel
        # (profiles/counters.json: WRITE_SIZE + 2*FETCH_SIZE, the gfx950 correction); null when there is none.
        traffic = None
        traffic_detail = committed_counters(args.config, args.mode, ctx.last_kernel) if my_rows == H else None
        if traffic_detail and traffic_detail.get("total_bytes"):
            traffic = round(traffic_detail["total_bytes"])
        flops = algorithmic_flops(W, H, ns, npl, hit_frac) * (my_rows / float(H))
        # second view: this path is bound by VALU issue, not by HBM.  Executed utilisation from the committed
        # rocprofv3 counters of this same kernel and workload: SQ_INSTS_VALU wave-instructions x 64 lanes per
        # launch / kernel time / 78.65 T lane-ops/s (the fp32 vector peak with no FMA: -ffp-contract=off).
        ctr = committed_counters(args.config, args.mode, ctx.last_kernel) if my_rows == H else None
        valu = None
        if ctr and ctr.get("SQ_INSTS_VALU"):
            lane_ops = ctr["SQ_INSTS_VALU"] * 64.0
            valu = {"executed_wave_instructions_per_launch": round(ctr["SQ_INSTS_VALU"]), "lane_ops_per_launch": lane_ops,
                    "achieved": round(lane_ops / (kernel_ms * 1e-3) / 1e12, 3), "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s",
                    "frac": round(lane_ops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS, 5),
                    "source": ctr.get("source"),
                    "note": "executed VALU instructions (PMC) over the live kernel time; the reference's all-pairs loop would "
                            "need %.3g flops per launch (SURVEY 8(d) formula), which the culling kernel provably does not have to do" % flops}
        roofline = {
            "bound": "hbm", "kernel": ctx.last_kernel, "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
            "traffic_unit": "bytes per launch (compare with bytes_per_launch)", "traffic_source": (traffic_detail or {}).get("source"),
            "bytes_per_launch": bytes_alg, "kernel_ms": round(kernel_ms, 5),
            "kernel_ms_note": "one launch at a time on one stream (HIP events, median of 5 batches), as in the rocprofv3 summaries under profiles/",
            "valu": valu,
        }

        cpu = None
        if not args.no_cpu_baseline and not distributed:
            import oracle as O
            import util as U
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            sc = O.Scene.from_arrays(sph, pl)
            op = U.oracle_params(params)
            reps = 3 if rays_per_frame <= 4_000_000 else 1   # median of >= 3 frames (1 for the big configs)
            times = []
            for _ in range(reps):
                t1 = time.perf_counter()
                O.render(op, sc, mode, threads=threads)
                times.append(time.perf_counter() - t1)
            dt = sorted(times)[len(times) // 2]
            cpu = {"value": round(rays_per_frame / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": "%d full %dx%d frame(s) of the same scene and mode (median), row-block partition over %d threads, "
                             "gcc -O2 -ffp-contract=off; %.2f s wall per frame" % (reps, W, H, threads, dt)}
            if threads > 1 and rays_per_frame <= 4_000_000:
                # SURVEY 8(d) asks for T=1 beside T=all; the middle quarter of the rows keeps it to ~1 s
                rows1 = max(8, (H // 4) // 8 * 8)
                t1 = time.perf_counter()
                O.render(op, sc, mode, threads=1, row0=(H - rows1) // 2, rows=rows1)
                dt1 = time.perf_counter() - t1
                cpu["single_thread"] = {"value": round((W - 1) * rows1 / dt1 / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                                        "sample": "rows %d..%d of the same frame; %.2f s wall" % ((H - rows1) // 2, (H - rows1) // 2 + rows1, dt1)}

        out = {
            "metric": BASELINE_METRIC,
            "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": K, "warmup": Wm,
            "ms_per_step": round(elapsed / K * 1e3, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %d spheres + %d planes, mode %s, SURVEY App. D scene seed %d"
                                   % (args.config, W, H, ns, npl, args.mode, seed),
                       "rays_per_frame": rays_per_frame, "kernel": ctx.last_kernel, "prewarm_ms": args.prewarm_ms,
                       "parallelism": "1 GPU" if n_gpus == 1 else "rows sharded over %d GPUs; %s" % (n_gpus, exchange_note)},
            "timing": timing, "roofline": roofline, "cpu_baseline": cpu,
        }
        if args.what != "trace":
            out["metric"] = "Mrays/s through the whole Update (trace + minimise + D2H of the minimised stream); not the graded metric"
            out["roofline"] = None
        if not distributed and args.what == "trace":
            eff_ms = elapsed / K * 1e3
            out["config"]["frames_in_flight"] = max(1, args.frames_in_flight)
            roofline["pipelined"] = {
                "frames_in_flight": max(1, args.frames_in_flight), "effective_ms_per_frame": round(eff_ms, 5),
                "achieved": round(bytes_alg / (eff_ms * 1e-3) / 1e9, 2), "unit": "GB/s",
                "frac": round(bytes_alg / (eff_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "whole-job rate with overlapping launches; 'achieved'/'frac' above are for one launch alone"}
        # true / false = the last frame was compared with the committed golden SHA-256; null = nothing was compared
        out["verified_against_golden"] = verified
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(mrays / cpu["value"], 1)

    ctx.close()
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

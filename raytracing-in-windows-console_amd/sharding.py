"""Row sharding of a frame over ranks and its assembly on a root rank.

Two frame loops over the same partition: RowShardedRounds (bench.py's default: frames in rounds of N with
rotating roots, one all-to-all per round) and RowShardedFrames (one point-to-point gather per frame).

The frame is row-major with row stride W*S (RayTracing.cu:238,457), so rows [r0, r1) are the contiguous
byte range [r0*W*S, r1*W*S).  Rank g traces rows [H*g/N, H*(g+1)/N); the root posts one receive per
peer straight into that peer's byte range of the full frame and every peer posts one send of its slab:
a gather made of point-to-point transfers (each peer crosses its own xGMI link into the root), not a ring.
Works on any torch.distributed backend (nccl = RCCL on GPUs; gloo in the CPU tests).
"""


def row_bounds(height, world):
    """Row split points: rank g owns rows [b[g], b[g+1])."""
    return [height * g // world for g in range(world + 1)]


def slab_bytes(bounds, rank, width, record_size):
    return (bounds[rank + 1] - bounds[rank]) * width * record_size


def post_gather(dist, rank, world, bounds, width, record_size, root_frame=None, slab=None):
    """Posts the transfers of one frame; returns the request list (empty for world 1 or empty slabs).

    Root (rank 0): `root_frame` is the flat uint8 tensor of the whole frame; its own rows are already in
    place.  Peers: `slab` is the flat uint8 tensor holding exactly their rows."""
    ops = []
    if rank == 0:
        for g in range(1, world):
            lo, hi = bounds[g] * width * record_size, bounds[g + 1] * width * record_size
            if hi > lo:
                ops.append(dist.P2POp(dist.irecv, root_frame[lo:hi], g))
    else:
        if slab.numel() > 0:
            ops.append(dist.P2POp(dist.isend, slab, 0))
    return dist.batch_isend_irecv(ops) if ops else []


def wait_all(reqs):
    for r in reqs:
        r.wait()


class RowShardedFrames:
    """The N > 1 frame loop of bench.py: every rank renders its row slab of every frame, and the frame is
    assembled on its root rank by point-to-point transfers.

    `rotate_root=True` (default): frame i is assembled on rank i % N, so that successive frames converge on
    different GPUs: every GPU's inbound xGMI links carry one frame in N instead of rank 0's carrying all of
    them, and each finished frame sits next to its own PCIe link to the host.  `rotate_root=False`: every
    frame is assembled on rank 0.

    Buffers: a ring of `nbuf` whole frames (used when this rank is the root; it renders its own rows in
    place) and a ring of `nbuf` slabs (used otherwise), so the transfers of a frame overlap the rendering of
    the next ones.  `render(buffer, row0, rows, out_row_base)` must queue the rendering of rows
    [row0, row0+rows) into `buffer`, whose first byte is row `out_row_base`, on the stream (or thread) the
    transfers are ordered after -- torch's current stream on GPUs.  Device-agnostic: the CPU tests drive it
    over gloo with the oracle as renderer."""

    def __init__(self, dist, torch, rank, world, width, height, record_size, device, nbuf=2, rotate_root=True):
        self.dist, self.rank, self.world = dist, rank, world
        self.W, self.H, self.S, self.nbuf = width, height, record_size, nbuf
        self.rotate = rotate_root
        self.bounds = row_bounds(height, world)
        self.row0 = self.bounds[rank]
        self.rows = self.bounds[rank + 1] - self.bounds[rank]
        self.frames = None
        if rank == 0 or rotate_root:
            # 20*W*H bytes each: the reference's frame size whatever the mode (PrintMachine.cpp:140)
            self.frames = [torch.zeros(20 * width * height, dtype=torch.uint8, device=device) for _ in range(nbuf)]
        self.slabs = None
        if rank != 0 or rotate_root:
            self.slabs = [torch.zeros(max(1, record_size * width * self.rows), dtype=torch.uint8, device=device) for _ in range(nbuf)]
        self.pending_frame = [None] * nbuf
        self.pending_slab = [None] * nbuf
        self.n_root = 0   # frames this rank has been root of
        self.n_peer = 0   # frames this rank has sent a slab for
        self.where = {}   # frame number -> frame ring index (root frames only)

    def root_of(self, i):
        return i % self.world if self.rotate else 0

    def step(self, i, render):
        root = self.root_of(i)
        if root == self.rank:
            b = self.n_root % self.nbuf
            self.n_root += 1
            if self.pending_frame[b] is not None:  # the transfers that last used this frame buffer
                wait_all(self.pending_frame[b])
                self.pending_frame[b] = None
            render(self.frames[b], self.row0, self.rows, 0)
            ops = []
            for g in range(self.world):
                lo, hi = self.bounds[g] * self.W * self.S, self.bounds[g + 1] * self.W * self.S
                if g != self.rank and hi > lo:
                    ops.append(self.dist.P2POp(self.dist.irecv, self.frames[b][lo:hi], g))
            self.pending_frame[b] = self.dist.batch_isend_irecv(ops) if ops else []
            self.where[i] = b
        else:
            b = self.n_peer % self.nbuf
            self.n_peer += 1
            if self.pending_slab[b] is not None:
                wait_all(self.pending_slab[b])
                self.pending_slab[b] = None
            render(self.slabs[b], self.row0, self.rows, self.row0)
            n = self.S * self.W * self.rows
            self.pending_slab[b] = self.dist.batch_isend_irecv(
                [self.dist.P2POp(self.dist.isend, self.slabs[b][:n], root)]) if n else []

    def drain(self):
        for ring in (self.pending_frame, self.pending_slab):
            for b in range(self.nbuf):
                if ring[b] is not None:
                    wait_all(ring[b])
                    ring[b] = None

    def frame(self, i):
        """On the root of frame i: the buffer it was assembled in (valid after drain() and until nbuf more
        frames have been rooted here)."""
        return self.frames[self.where[i]]


def _median(xs):
    ys = sorted(xs)
    n = len(ys)
    return ys[n // 2] if n % 2 else 0.5 * (ys[n // 2 - 1] + ys[n // 2])


def timed_frames(dist, torch, pipe, render, steps, warmup, device, synchronize, prewarm=0, min_total=0.0, max_repeats=1):
    """bench.py's timing contract for N > 1: warm-up, then exactly `steps` frames between barrier +
    synchronize on both sides; returns the MAX over ranks of the elapsed seconds.  `prewarm` more frames run
    first, neither timed nor counted (clock ramp of an idle GPU; a fixed count so that all ranks agree).
    With max_repeats > 1 the window of `steps` frames is repeated, each time from a drained pipeline and between
    barriers, until `min_total` seconds have been measured (the decision uses the all-reduced times, so every rank
    takes it alike); the MEDIAN window is returned and all of them are left in pipe.timed_windows."""
    import time
    # communicator set-up (not a timed or counted step): one full rotation of roots, so that every pair of
    # ranks that will exchange slabs has done so once
    for i in range((pipe.world if pipe.rotate else 1) + prewarm):
        pipe.step(i, render)
    pipe.drain()
    for i in range(warmup):
        pipe.step(i, render)
    pipe.drain()
    windows = []
    while True:
        dist.barrier()
        synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.step(i, render)
        pipe.drain()
        synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        windows.append(float(t.item()))
        if len(windows) >= max_repeats or sum(windows) >= min_total:
            break
    pipe.timed_windows = windows
    return _median(windows)


class RowShardedRounds:
    """The N > 1 frame loop with one exchange per round of M*R frames, R = the number of root ranks.

    `roots` (ascending rank numbers; default: every rank) are the ranks frames are assembled on.  Frame f of a
    round (f = m*R + i, m < M) is assembled on roots[i]: consecutive frames rotate over the roots and every root
    receives M frames per round.  roots = [0] is the in-order form: every frame is assembled on rank 0, in frame
    order (what the reference's single consumer, PrintMachine::SetDataInBackBuffer, RayTracingManager.cu:150,
    expects), as one gather per M frames.  In a round every rank renders its row slab of each of the
    round's frames into one send buffer (the slab of frame f at unit (j*M + m), so that everything bound for
    rank j is contiguous) and ONE all_to_all_single moves it: per round each directed xGMI link carries M
    slabs, and the per-call cost of the collective (of the order of a whole frame's trace) is paid once per
    M*N frames -- fewer, larger collectives.  A round may hold fewer frames (the tail of a run): the absent
    frames shrink the splits.

    Two forms of the slabs:
      * records (`pixel_bytes` = S, `frames_per_root` = 1): the units of a root arrive in rank order = row
        order, i.e. as the finished frame; nothing else to do.
      * compact pixel words (`pixel_bytes` = 4, RTX_RENDER_COMPACT): a rank ships 4 instead of S bytes per
        pixel and the root expands them into records (rtx_expand).  What a root receives from rank r is that
        rank's rows of its M frames back to back, so frame m is the segment list
        [(source pixel, destination pixel, pixels) per rank] that `finish` is handed.

    Callbacks (device-specific; bench.py supplies the GPU ones, the CPU tests the oracle):
      render_round(q, b, nframes): queue the rendering of this rank's rows of the round's first `nframes`
          frames, frame f into `self.unit(b, f)`, ordered before whatever is queued next on the current stream.
      finish(q, b, work, mine) -> token: `work` is the all-to-all's request, `mine` the list of
          (m, segments) of the frames rooted here; must arrange for `work` to complete and then expand
          frame m from `self.recv[b]` into `self.frames[b][m]`; returns an object whose wait() orders the
          caller after all that (None = nothing to wait for).  Without `finish` the token is `work` itself.
    Buffers are rings of `nbuf`, so the exchange of a round overlaps the rendering of the next."""

    def __init__(self, dist, torch, rank, world, width, height, record_size, device, nbuf=2, frames_per_root=1, pixel_bytes=None,
                 finish=None, roots=None):
        self.dist, self.torch, self.rank, self.world = dist, torch, rank, world
        self.roots = list(range(world)) if roots is None else sorted(set(int(r) for r in roots))
        if not self.roots or self.roots[0] < 0 or self.roots[-1] >= world:
            raise ValueError("roots must be rank numbers")
        self.R = len(self.roots)
        self.root_index = {r: i for i, r in enumerate(self.roots)}
        self.W, self.H, self.S, self.nbuf, self.M = width, height, record_size, nbuf, frames_per_root
        self.px = record_size if pixel_bytes is None else pixel_bytes
        self.expanding = self.px != record_size
        if not self.expanding and frames_per_root != 1:
            raise ValueError("records land as the finished frame only with one frame per root and round")
        if self.expanding and finish is None:
            raise ValueError("compact slabs need a finish callback that expands them")
        self.finish = finish
        self.bounds = row_bounds(height, world)
        self.row0 = self.bounds[rank]
        self.rows = self.bounds[rank + 1] - self.bounds[rank]
        self.unit_len = self.px * width * self.rows
        self.round_frames = self.M * self.R
        self.send = [torch.zeros(self.round_frames * self.unit_len, dtype=torch.uint8, device=device) for _ in range(nbuf)]
        # 20*W*H bytes each: the reference's frame size whatever the mode (PrintMachine.cpp:140); only roots hold frames
        self.is_root = rank in self.root_index
        self.frames = [[torch.zeros(20 * width * height if self.is_root else 0, dtype=torch.uint8, device=device) for _ in range(self.M)]
                       for _ in range(nbuf)]
        self.recv = None
        if self.expanding and self.is_root:
            self.recv = [torch.zeros(self.M * self.px * width * height, dtype=torch.uint8, device=device) for _ in range(nbuf)]
        self.nothing = torch.zeros(0, dtype=torch.uint8, device=device)
        self.pending = [None] * nbuf
        self.where = {}   # frame number -> (ring index, m) for frames rooted here
        self._mine_cache = {}
        self._full_splits = None

    def root_of(self, i):
        return self.roots[i % self.R]

    def unit(self, b, f):
        """The send-buffer slot of frame f of a round (this rank's rows of it)."""
        k = (f % self.R) * self.M + f // self.R
        return self.send[b][k * self.unit_len:(k + 1) * self.unit_len]

    def frames_for_root(self, j, nframes):
        """How many of a round's first `nframes` frames are rooted on rank j."""
        i = self.root_index.get(j)
        if i is None or nframes <= i:
            return 0
        return (nframes - i + self.R - 1) // self.R

    def segments(self, m, mine):
        """Frame m of this root in recv[b] when `mine` frames arrived: (src pixel, dst pixel, pixels) per rank."""
        segs, src = [], 0
        for r in range(self.world):
            rows_r = self.bounds[r + 1] - self.bounds[r]
            if rows_r:
                segs.append(((src + m * rows_r) * self.W, self.bounds[r] * self.W, rows_r * self.W))
            src += mine * rows_r
        return segs

    def _wait(self, b):
        if self.pending[b] is not None:   # the exchange (and expansion) that last used ring slot b
            self.pending[b].wait()
            self.pending[b] = None

    def round(self, q, nframes, render_round):
        b = q % self.nbuf
        self._wait(b)
        render_round(q, b, nframes)
        N, M = self.world, self.M
        if nframes == self.round_frames and self._full_splits is not None:
            ins, mine, outs = self._full_splits
        else:
            ins = [self.frames_for_root(j, nframes) * self.unit_len for j in range(N)]
            mine = self.frames_for_root(self.rank, nframes)
            outs = [mine * self.px * self.W * (self.bounds[r + 1] - self.bounds[r]) for r in range(N)]
            if nframes == self.round_frames:
                self._full_splits = (ins, mine, outs)
        if mine == 0:
            out = self.nothing
        elif self.expanding:
            out = self.recv[b][:sum(outs)]
        else:
            out = self.frames[b][0][:sum(outs)]
        # units bound for rank j are contiguous and the present ones (m < frames_for_root) come first
        if nframes == self.round_frames:
            inp = self.send[b]
        else:
            inp = self._present_units(b, nframes)
        work = self.dist.all_to_all_single(out, inp, outs, ins, async_op=True)
        for m in range(mine):
            i = self.root_index[self.rank]
            self.where[q * self.round_frames + m * self.R + i] = (b, m)
            self.where.pop((q - self.nbuf) * self.round_frames + m * self.R + i, None)   # overwritten by now
        if self.finish is not None:
            if mine not in self._mine_cache:   # the segment lists depend on `mine` only: built once
                self._mine_cache[mine] = [(m, self.segments(m, mine)) for m in range(mine)] if self.expanding else []
            self.pending[b] = self.finish(q, b, work, self._mine_cache[mine])
        else:
            self.pending[b] = work

    def _present_units(self, b, nframes):
        """Partial round: the present units of each destination, packed (all_to_all_single wants one contiguous
        input).  Only the last round of a run takes this path."""
        parts = []
        for i, j in enumerate(self.roots):
            k = self.frames_for_root(j, nframes)
            if k:
                parts.append(self.send[b][i * self.M * self.unit_len:(i * self.M + k) * self.unit_len])
        if not parts:
            return self.nothing
        return self.torch.cat(parts)

    def run(self, first_round, count, render_round):
        """`count` frames starting at round `first_round`; returns the next round number."""
        q = first_round
        while count > 0:
            n = min(self.round_frames, count)
            self.round(q, n, render_round)
            count -= n
            q += 1
        return q

    def drain(self):
        for b in range(self.nbuf):
            self._wait(b)

    def frame(self, i):
        """On the root of frame i: the buffer it was assembled in (valid after drain() and until nbuf more
        rounds have passed)."""
        b, m = self.where[i]
        return self.frames[b][m]


def timed_rounds(dist, torch, pipe, render_round, steps, warmup, device, synchronize, prewarm=0, min_total=0.0, max_repeats=1):
    """bench.py's timing contract for RowShardedRounds: warm-up, then exactly `steps` frames between
    barrier + synchronize on both sides; returns (MAX over ranks of the elapsed seconds, first timed round).
    `prewarm` more frames run first, neither timed nor counted (clock ramp of an idle GPU; a fixed count so that
    all ranks agree).  With max_repeats > 1 the window of `steps` frames is repeated as in timed_frames: the MEDIAN
    window and the first round of the LAST window are returned, all windows are left in pipe.timed_windows."""
    import time
    # communicator set-up: one full round, neither timed nor counted; then the run-in
    q = pipe.run(0, pipe.round_frames + prewarm, render_round)
    pipe.drain()
    q = pipe.run(q, warmup, render_round)
    pipe.drain()
    windows = []
    while True:
        dist.barrier()
        synchronize()
        t0 = time.perf_counter()
        q0 = q
        q = pipe.run(q, steps, render_round)
        pipe.drain()
        synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        windows.append(float(t.item()))
        if len(windows) >= max_repeats or sum(windows) >= min_total:
            break
    pipe.timed_windows = windows
    return _median(windows), q0

"""Row sharding of a frame over ranks and its assembly on rank 0.

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
    """The N > 1 frame loop of bench.py: every rank renders its row slab, rank 0 assembles the frame.

    Buffers are a ring of `nbuf` (rank 0: whole frames, in which it renders its own rows in place; peers:
    slabs), so that the transfer of frame i overlaps the rendering of frame i+1.  `render(buffer, row0, rows,
    out_row_base)` must queue the rendering of rows [row0, row0+rows) into `buffer`, whose first byte is row
    `out_row_base`, on the stream (or thread) the transfers are ordered after -- torch's current stream on
    GPUs.  Device-agnostic: the CPU tests drive it over gloo with the oracle as renderer."""

    def __init__(self, dist, torch, rank, world, width, height, record_size, device, nbuf=2):
        self.dist, self.rank, self.world = dist, rank, world
        self.W, self.H, self.S, self.nbuf = width, height, record_size, nbuf
        self.bounds = row_bounds(height, world)
        self.row0 = self.bounds[rank]
        self.rows = self.bounds[rank + 1] - self.bounds[rank]
        if rank == 0:
            # 20*W*H bytes each: the reference's frame size whatever the mode (PrintMachine.cpp:140)
            self.bufs = [torch.zeros(20 * width * height, dtype=torch.uint8, device=device) for _ in range(nbuf)]
        else:
            self.bufs = [torch.zeros(record_size * width * self.rows, dtype=torch.uint8, device=device) for _ in range(nbuf)]
        self.pending = [None] * nbuf

    def step(self, i, render):
        b = i % self.nbuf
        if self.pending[b] is not None:  # the transfer that last used this buffer
            wait_all(self.pending[b])
            self.pending[b] = None
        if self.rank == 0:
            render(self.bufs[b], self.row0, self.rows, 0)
            self.pending[b] = post_gather(self.dist, 0, self.world, self.bounds, self.W, self.S, root_frame=self.bufs[b])
        else:
            render(self.bufs[b], self.row0, self.rows, self.row0)
            self.pending[b] = post_gather(self.dist, self.rank, self.world, self.bounds, self.W, self.S, slab=self.bufs[b])

    def drain(self):
        for b in range(self.nbuf):
            if self.pending[b] is not None:
                wait_all(self.pending[b])
                self.pending[b] = None

    def frame(self, i):
        """Rank 0: the buffer frame i was assembled in (valid after drain())."""
        return self.bufs[i % self.nbuf]


def timed_frames(dist, torch, pipe, render, steps, warmup, device, synchronize):
    """bench.py's timing contract for N > 1: warm-up, then exactly `steps` frames between barrier +
    synchronize on both sides; returns the MAX over ranks of the elapsed seconds."""
    import time
    for i in range(warmup):
        pipe.step(i, render)
    pipe.drain()
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
    return float(t.item())

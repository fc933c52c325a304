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

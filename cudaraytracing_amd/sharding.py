"""Multi-GPU sharding of the path: one process per GPU, torch.distributed for the exchange.

Pixels are independent and the scene (<= 115 KB) is replicated, so rendering needs no
collective.  Two ways to shard:

  frames  whole frames are dealt to ranks round-robin (frame f -> rank f % world); nothing is
          exchanged on the data path.
  bands   one frame is cut into `world` row bands (configs.band_rows); every rank renders its
          band (rc_opts.y_offset / full_height) and the bands are assembled on one rank with
          point-to-point transfers: the root posts one receive per peer straight into the
          band's rows of the full frame, each peer posts one send (on RCCL the batch runs as
          one grouped operation and each transfer rides that pair's own xGMI link).

The reference has no multi-GPU path (single device, default stream: kernel.cu:123).
"""
import torch
import torch.distributed as dist

from .configs import band_rows


def frames_for_rank(frames, world, rank):
    """Frame numbers rendered by `rank` when frames are dealt round-robin."""
    return list(range(rank, frames, world))


def gather_bands(band, full, height, world, rank, dst=0):
    """Assemble row bands on rank `dst`.

    band: this rank's rows [rows_r, width, 4]; full: [height, width, 4] on `dst` (ignored
    elsewhere).  Returns when this rank's part of the exchange has completed.
    """
    if world == 1:
        full.copy_(band)
        return
    ops = []
    if rank == dst:
        for r in range(world):
            b, e = band_rows(height, world, r)
            if r == dst:
                full[b:e].copy_(band)
            elif e > b:
                ops.append(dist.P2POp(dist.irecv, full[b:e], r))
    else:
        b, e = band_rows(height, world, rank)
        if e > b:
            ops.append(dist.P2POp(dist.isend, band, dst))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def max_over_ranks(value, device):
    """MAX of a python float over all ranks (used for the timed region of bench.py)."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

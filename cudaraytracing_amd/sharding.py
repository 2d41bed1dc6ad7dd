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


def sub_bands(begin, end, parts):
    """Split rows [begin, end) into `parts` nearly equal sub-bands (for render/transfer overlap)."""
    rows = end - begin
    parts = max(1, min(parts, rows)) if rows > 0 else 1
    base, rem = divmod(rows, parts)
    out, b = [], begin
    for i in range(parts):
        e = b + base + (1 if i < rem else 0)
        out.append((b, e))
        b = e
    return out


def rank_pieces(height, world, rank, parts=2, layout="contiguous"):
    """Row ranges [(begin, end), ...] that `rank` renders, in the order it renders and sends them.

    contiguous  one band per rank (configs.band_rows), cut into `parts` sub-bands.
    interleaved the frame is cut into 2 * world bands and rank r takes band r and band world + r: rows near
                the middle of the picture cost more than rows near its edges (more hits), and with a cost
                that is symmetric about the middle this pairs band r with the mirror image of band
                world - 1 - r, an outer band with an inner one.  Always two pieces.
    """
    if layout == "contiguous":
        return sub_bands(*band_rows(height, world, rank), parts)
    if layout == "interleaved":
        return [band_rows(height, 2 * world, rank), band_rows(height, 2 * world, world + rank)]
    raise ValueError(f"unknown band layout {layout!r}")


class BandGather:
    """Row-band gather with render/transfer overlap.

    A rank's rows are rendered piece by piece (rank_pieces); the transfer of piece k is posted as soon
    as its render has been enqueued, so it travels while piece k+1 renders (the collective
    library orders each transfer after the work already enqueued on the current stream).  The
    root posts all its receives first, straight into the rows of the full frame.  A rank keeps its
    pieces one after the other in one [rows, width, ...] tensor.
    """

    def __init__(self, height, world, rank, parts=2, dst=0, layout="contiguous"):
        self.height, self.world, self.rank, self.dst, self.layout = height, world, rank, dst, layout
        self.parts = 2 if layout == "interleaved" else parts
        self.begin, self.end = band_rows(height, world, rank)      # the contiguous band (layout "contiguous")
        self.pieces = rank_pieces(height, world, rank, parts, layout)
        self.offsets, o = [], 0                                    # first row of piece k inside this rank's tensor
        for b, e in self.pieces:
            self.offsets.append(o)
            o += e - b
        self.rows = o
        self._reqs = []

    def post_receives(self, full):
        """Root only: one receive per peer sub-band, into full[rows].  One batch per piece
        index, so the root's groups mirror the peers' (each peer sends piece k as group k)."""
        if self.rank != self.dst or self.world == 1:
            return
        for k in range(self.parts):
            ops = []
            for r in range(self.world):
                if r == self.dst:
                    continue
                pieces = rank_pieces(self.height, self.world, r, self.parts, self.layout)
                if k < len(pieces) and pieces[k][1] > pieces[k][0]:
                    ops.append(dist.P2POp(dist.irecv, full[pieces[k][0]:pieces[k][1]], r))
            if ops:
                self._reqs += dist.batch_isend_irecv(ops)

    def piece_rendered(self, k, band, full=None):
        """Call right after sub-band k of this rank's band has been enqueued for rendering.
        band: this rank's [rows, width, 4] tensor."""
        sb, se = self.pieces[k]
        if se <= sb:
            return
        view = band[self.offsets[k]: self.offsets[k] + (se - sb)]
        if self.world == 1 or self.rank == self.dst:
            if full is not None:
                full[sb:se].copy_(view, non_blocking=True)
        else:
            self._reqs += dist.batch_isend_irecv([dist.P2POp(dist.isend, view, self.dst)])

    def wait(self):
        for r in self._reqs:
            r.wait()
        self._reqs = []


def max_over_ranks(value, device):
    """MAX of a python float over all ranks (used for the timed region of bench.py)."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device):
    """One python float per rank -> list over ranks, on every rank (per-GPU kernel times)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]

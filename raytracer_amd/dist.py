"""Multi-GPU partitioning of one frame (one process per GPU, torch.distributed).

The reference's only parallelism is N threads pulling RenderChunks from a queue
(src/render.rs:186-224); here the chunk list is dealt to ranks instead, and each
frame needs exactly ONE exchange step:

  * tile mode  : the image is cut into horizontal strips with
    create_render_chunks (viewport.rs:30-54), dealt round-robin so expensive and
    cheap regions interleave; each rank renders its strips into one packed device
    buffer; rank 0 gathers the packed buffers (RCCL gather = one send per peer
    straight over its own xGMI link) and scatters the rows into the frame.
    Bit-identical to a single-GPU render by construction (counter-based RNG,
    fixed per-pixel sample order).
  * spp mode   : every rank renders the full frame for a contiguous range of
    sample indices (multiples of the sample segment (rt_sample_segment)) into a float3 accumulator
    of per-sample-clamped sums (implementation.rs:635,656: the clamp is per sample,
    so sums add); rank 0 gathers the accumulators and adds them IN RANK ORDER
    (deterministic), then divides and converts once.  A plain all-reduce would do
    too but fixes no summation order.

Nothing here touches pixels; it only moves them.  Works on any backend
(`nccl` = RCCL on the GPU box, `gloo` in CPU tests).
"""
import torch
import torch.distributed as dist

from .scene import create_render_chunks, sample_segment


def _gather(tensor, gather_list, dst, rank, world):
    """dist.gather, or (backends without it) all_gather keeping only rank dst's copy."""
    try:
        dist.gather(tensor, gather_list, dst=dst)
        return gather_list
    except (NotImplementedError, RuntimeError) as e:
        if "gather" not in str(e).lower() and "support" not in str(e).lower():
            raise
        everywhere = [torch.zeros_like(tensor) for _ in range(world)]
        dist.all_gather(everywhere, tensor)
        return everywhere if rank == dst else None


def strip_chunks(width, height, world, strips_per_rank=8):
    """The frame as `world * strips_per_rank` horizontal strips (fewer if the image
    is shorter), in create_render_chunks order."""
    strips = max(1, min(height, world * strips_per_rank))
    return create_render_chunks((width, height), (1, strips))


def chunks_of_rank(chunks, rank, world):
    return [c for i, c in enumerate(chunks) if i % world == rank]


def sample_range_of_rank(samples_per_pixel, rank, world):
    """Contiguous [first, first+count) of sample indices for `rank`; boundaries fall on
    multiples of the sample segment (rt_sample_segment) so that every segment sum is formed on one rank."""
    seg = sample_segment()
    segs = (samples_per_pixel + seg - 1) // seg
    lo = (segs * rank) // world * seg
    hi = min((segs * (rank + 1)) // world * seg, samples_per_pixel)
    return lo, max(0, hi - lo)


class StripCompositor:
    """Packed per-rank strip buffers + the gather onto rank 0."""

    def __init__(self, width, height, chunks, rank, world, device, channels=3, dtype=torch.float32, dst=0):
        self.width, self.height, self.chunks, self.rank, self.world, self.dst = width, height, chunks, rank, world, dst
        self.mine = chunks_of_rank(chunks, rank, world)
        rows = [sum(c.height for c in chunks_of_rank(chunks, r, world)) for r in range(world)]
        self.max_rows = max(rows) if rows else 0
        self.packed = torch.zeros((max(self.max_rows, 1), width, channels), dtype=dtype, device=device)
        self.gathered = ([torch.zeros_like(self.packed) for _ in range(world)] if rank == dst and world > 1 else None)
        self.frame = torch.zeros((height, width, channels), dtype=dtype, device=device) if rank == dst else None
        # (row offset inside my packed buffer, chunk) pairs
        self.slots, row = [], 0
        for c in self.mine:
            self.slots.append((row, c))
            row += c.height

    def composite(self):
        """One exchange: afterwards rank `dst` holds the whole frame in .frame."""
        sources = [self.packed]
        if self.world > 1:
            sources = _gather(self.packed, self.gathered, self.dst, self.rank, self.world)
        if self.rank != self.dst:
            return None
        for r in range(self.world):
            row = 0
            for c in chunks_of_rank(self.chunks, r, self.world):
                self.frame[c.top:c.top + c.height] = sources[r][row:row + c.height]
                row += c.height
        return self.frame


class InterleavedCompositor:
    """Fine-grained tile partition: the frame is cut into strips of `strip_height`
    rows (a multiple of 8 keeps the kernel's 8x8 ray tiles inside one strip) and rank r
    owns strips r, r+world, r+2*world, ...  The rank renders ALL of them in one launch
    (rt_render_strips_device) into `.packed`; one gather later rank `dst` scatters the
    rows into `.frame` with a precomputed row index."""

    def __init__(self, width, height, rank, world, device, strip_height=8, channels=3, dtype=torch.float32, dst=0):
        from .scene import strip_rows
        self.width, self.height, self.rank, self.world, self.dst, self.strip_height = width, height, rank, world, dst, strip_height
        self.rows = [strip_rows(height, strip_height, r, world) for r in range(world)]
        assert sum(self.rows) == height
        self.my_rows = self.rows[rank]
        max_rows = max(max(self.rows), 1)
        self.packed = torch.zeros((max_rows, width, channels), dtype=dtype, device=device)
        self.gathered = self.frame = self.row_index = None
        if rank == dst:
            # One receive buffer for all ranks and ONE scatter per frame: row k of rank r goes to its image row,
            # the padding rows of ranks with fewer rows go to a spare row below the frame.
            self._all = torch.zeros((world, max_rows, width, channels), dtype=dtype, device=device)
            self.gathered = [self._all[r] for r in range(world)] if world > 1 else None
            self._frame_ext = torch.zeros((height + 1, width, channels), dtype=dtype, device=device)
            self.frame = self._frame_ext[:height]
            index = []
            for r in range(world):
                rows = self.image_rows(r)
                index.extend(rows + [height] * (max_rows - len(rows)))
            self.row_index = torch.tensor(index, dtype=torch.long, device=device)

    def image_rows(self, r):
        """Image row of every packed local row of rank r."""
        h, out = self.strip_height, []
        top = r * h
        while top < self.height:
            out.extend(range(top, min(top + h, self.height)))
            top += self.world * h
        return out

    def composite(self):
        got = None
        if self.world > 1:
            got = _gather(self.packed, self.gathered, self.dst, self.rank, self.world)
        if self.rank != self.dst:
            return None
        if got is not None and got is not self.gathered:       # (the all_gather fallback brings its own buffers)
            for r in range(self.world):
                self._all[r].copy_(got[r])
        src = self._all if self.world > 1 else self.packed.unsqueeze(0)
        self._frame_ext.index_copy_(0, self.row_index, src.reshape(-1, self.width, src.shape[-1]))
        return self.frame


def columns_of_rank(width, rank, world):
    """Image column of every packed column of `rank` under the dealt tile-column partition (8-pixel tile columns rank, rank + world, ...)."""
    out, tiles_x = [], (width + 7) // 8
    for tx in range(rank, tiles_x, world):
        out.extend(range(tx * 8, min(tx * 8 + 8, width)))
    return out


class ColumnCompositor:
    """The finer tile partition: rank r owns the 8-pixel-wide tile columns r, r + world, r + 2 * world, ... over the whole height
    (rt_render_tile_columns_device renders them in one launch into `.packed`, `height` rows of `.my_width` pixels); one gather later
    rank `dst` scatters the columns into `.frame` with a precomputed column index.  Every rank holds a share of every region of the
    image, which 8-row strips do not give (per-rank times of an 8-rank split: DESIGN.md section 6)."""

    def __init__(self, width, height, rank, world, device, channels=3, dtype=torch.float32, dst=0):
        from .scene import tile_columns_width
        self.width, self.height, self.rank, self.world, self.dst = width, height, rank, world, dst
        self.widths = [tile_columns_width(width, r, world) for r in range(world)]
        assert sum(self.widths) == width
        self.my_width = self.widths[rank]
        max_w = max(max(self.widths), 1)
        self._packed_full = torch.zeros((height * max_w, channels), dtype=dtype, device=device)
        # (a rank's launch writes rows of ITS width: the leading height * my_width pixels of the buffer)
        self.packed = self._packed_full[:height * self.my_width].view(height, self.my_width, channels) if self.my_width else self._packed_full[:0]
        self.gathered = self.frame = self.col_index = None
        if rank == dst:
            self._all = torch.zeros((world, height * max_w, channels), dtype=dtype, device=device)
            self.gathered = [self._all[r] for r in range(world)] if world > 1 else None
            self.frame = torch.zeros((height, width, channels), dtype=dtype, device=device)
            self.col_index = [torch.tensor(self.image_columns(r), dtype=torch.long, device=device) for r in range(world)]

    def image_columns(self, r):
        """Image column of every packed column of rank r."""
        return columns_of_rank(self.width, r, self.world)

    def composite(self):
        got = None
        if self.world > 1:
            got = _gather(self._packed_full, self.gathered, self.dst, self.rank, self.world)
        if self.rank != self.dst:
            return None
        if got is not None and got is not self.gathered:
            for r in range(self.world):
                self._all[r].copy_(got[r])
        return self.assemble([self._all[r] for r in range(self.world)] if self.world > 1 else [self._packed_full])

    def assemble(self, sources):
        """Rank dst: the frame from every rank's packed buffer (sources[r] = rank r's `_packed_full`), one scatter per rank."""
        for r in range(self.world):
            w = self.widths[r]
            if w == 0:
                continue
            src = sources[r][:self.height * w].view(self.height, w, -1)
            self.frame.index_copy_(1, self.col_index[r], src)      # its packed columns to their image columns
        return self.frame


def reduce_accumulators(acc, rank, world, dst=0, how="ordered"):
    """Sum the per-rank accumulators onto rank `dst`.

    how="ordered"  : gather + add in rank order on `dst`: deterministic, equal bit for bit to a one-GPU frame that forms
                     its segment sums in the same ranges (the default).
    how="allreduce": one RCCL all-reduce (BASELINE configs[4] names it): every rank ends with the frame sum; the order of the
                     additions is the collective's, so the result may differ from the ordered sum in the last ulp."""
    if world == 1:
        return acc
    if how == "allreduce":
        total = acc.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        return total if rank == dst else None
    gathered = [torch.zeros_like(acc) for _ in range(world)] if rank == dst else None
    gathered = _gather(acc, gathered, dst, rank, world)
    if rank != dst:
        return None
    total = gathered[0].clone()
    for r in range(1, world):
        total += gathered[r]
    return total

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
    sample indices (multiples of the 32-sample segment) into a float3 accumulator
    of per-sample-clamped sums (implementation.rs:635,656: the clamp is per sample,
    so sums add); rank 0 gathers the accumulators and adds them IN RANK ORDER
    (deterministic), then divides and converts once.  A plain all-reduce would do
    too but fixes no summation order.

Nothing here touches pixels; it only moves them.  Works on any backend
(`nccl` = RCCL on the GPU box, `gloo` in CPU tests).
"""
import torch
import torch.distributed as dist

from .scene import create_render_chunks

SAMPLE_SEGMENT = 32   # RT_SAMPLE_SEGMENT of csrc/render.hip


def strip_chunks(width, height, world, strips_per_rank=8):
    """The frame as `world * strips_per_rank` horizontal strips (fewer if the image
    is shorter), in create_render_chunks order."""
    strips = max(1, min(height, world * strips_per_rank))
    return create_render_chunks((width, height), (1, strips))


def chunks_of_rank(chunks, rank, world):
    return [c for i, c in enumerate(chunks) if i % world == rank]


def sample_range_of_rank(samples_per_pixel, rank, world):
    """Contiguous [first, first+count) of sample indices for `rank`; boundaries fall on
    multiples of the 32-sample segment so that every segment sum is formed on one rank."""
    segs = (samples_per_pixel + SAMPLE_SEGMENT - 1) // SAMPLE_SEGMENT
    lo = (segs * rank) // world * SAMPLE_SEGMENT
    hi = min((segs * (rank + 1)) // world * SAMPLE_SEGMENT, samples_per_pixel)
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
        if self.world > 1:
            dist.gather(self.packed, self.gathered, dst=self.dst)
        if self.rank != self.dst:
            return None
        sources = self.gathered if self.world > 1 else [self.packed]
        for r in range(self.world):
            row = 0
            for c in chunks_of_rank(self.chunks, r, self.world):
                self.frame[c.top:c.top + c.height] = sources[r][row:row + c.height]
                row += c.height
        return self.frame


def reduce_accumulators(acc, rank, world, dst=0):
    """Sum the per-rank accumulators onto rank `dst` in rank order (deterministic)."""
    if world == 1:
        return acc
    gathered = [torch.zeros_like(acc) for _ in range(world)] if rank == dst else None
    dist.gather(acc, gathered, dst=dst)
    if rank != dst:
        return None
    total = gathered[0].clone()
    for r in range(1, world):
        total += gathered[r]
    return total

"""Pixel-row sharding of one frame across the GPUs of a node.

Pixels are independent in every stage of the path, so rank r renders the contiguous row band
`row_band(H, r, world)` with the SAME kernels (the band is just a shorter ray tensor).  The
only exchanges are: one all-gather of the image rows in the forward, and one all-reduce(sum)
of the per-Gaussian gradients in the backward.  One process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).  The reference has no
counterpart (its only multi-device code, DataParallelBatchifier, Utils.py:179-333, is unused).
"""
import torch
import torch.distributed as dist


def row_band(H, rank, world):
    """Contiguous, balanced row band [r0, r1) of rank `rank`; bands differ by at most one row."""
    base, extra = divmod(int(H), int(world))
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


class _GatherRows(torch.autograd.Function):
    """all_gather of row bands [B,h_r,W,C] -> [B,H,W,C]; backward hands each rank the slice of the
    upstream gradient that belongs to its own band (every rank holds the same full-image loss)."""

    @staticmethod
    def forward(ctx, band, H, group):
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        B, h, W, C = band.shape
        hmax = -(-H // world)
        pad = band.new_zeros((B, hmax, W, C))
        pad[:, :h] = band
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad.contiguous(), group=group)
        rows = [parts[r][:, : row_band(H, r, world)[1] - row_band(H, r, world)[0]] for r in range(world)]
        ctx.band = row_band(H, rank, world)
        return torch.cat(rows, dim=1)

    @staticmethod
    def backward(ctx, g_full):
        r0, r1 = ctx.band
        return g_full[:, r0:r1].contiguous(), None, None


def gather_rows(band, H, group=None):
    """Assemble the full image from per-rank row bands (single collective)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return band
    return _GatherRows.apply(band, H, group)


def gather_rows_async(band, H, group=None):
    """Start the all_gather of the row bands and return `finish() -> [B,H,W,C]`.  No autograd (the
    caller's loss is local to its band); the collective runs on the backend's own stream, so
    whatever is launched before finish() -- the band's backward -- overlaps it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return lambda: band
    world = dist.get_world_size(group)
    B, h, W, C = band.shape
    hmax = -(-H // world)
    if h == hmax:
        pad = band.contiguous()
    else:
        pad = band.new_zeros((B, hmax, W, C))
        pad[:, :h] = band
    parts = [torch.empty_like(pad) for _ in range(world)]
    work = dist.all_gather(parts, pad, group=group, async_op=True)

    def finish():
        work.wait()
        return torch.cat([parts[r][:, : row_band(H, r, world)[1] - row_band(H, r, world)[0]] for r in range(world)], dim=1)
    return finish


def allreduce_grads(tensors, group=None):
    """Sum the per-Gaussian gradients of all ranks with ONE flat all-reduce (bucketed: the
    concatenated [verts, sigmas, colours] gradient is ~N*15 floats, 3 MB at 50k Gaussians)."""
    grads = [t.grad for t in tensors if t is not None and t.grad is not None]
    if not grads or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()

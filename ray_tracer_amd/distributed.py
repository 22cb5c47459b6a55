"""Splitting one frame over the ranks of a torch.distributed job, one process per GPU.

The reference parallelises over pixels on one machine (rayon, src/renderer.rs:44-60);
pixels and samples are independent, so across GPUs the frame shards without any exchange
inside the render:

  tile split    rank r renders image rows r, r + N, r + 2N, ... (interleaved, so every
                rank gets the same mix of sky and ground).  No collective on the data path;
                rank 0 gathers the finished rows.
  sample split  every rank renders all pixels for a contiguous share of the samples into
                an f64 accumulation image; one sum-reduce (RCCL on GPUs) then one resolve.

These helpers hold the arithmetic of both, independent of what renders the pixels.
"""
import numpy as np


def row_tiles(width, world_size, rank):
    """(tile_size, tile_count, tile_index) of mirt_render_opts for this rank's rows."""
    return (int(width), int(world_size), int(rank))


def rows_of(rank, world_size, height):
    return range(rank, height, world_size)


def sample_range(spp, rank, world_size):
    """This rank's [begin, end) of the spp samples; the shares differ by at most one.  None when the
    share is empty (fewer samples than ranks): such a rank renders nothing -- passing (0, 0) on would
    mean "all samples" to mirt_render_accumulate -- and still takes part in the reduce with zeros."""
    lo, hi = rank * spp // world_size, (rank + 1) * spp // world_size
    return (lo, hi) if hi > lo else None


_gather_buffers = {}


def gather_rows(out, rank, world_size, group=None, dst=0):
    """out: (H, W, 3) uint8 holding this rank's rows; on `dst` it comes back complete."""
    import torch
    import torch.distributed as dist

    if world_size == 1:
        return out
    # gather wants equal shapes: every share is padded to the largest (ranks below H % N own one more row).
    # The staging tensors are kept between frames (a render loop gathers the same shapes again and again).
    height = out.shape[0]
    most = (height + world_size - 1) // world_size
    key = (most,) + tuple(out.shape[1:]) + (world_size, rank == dst)
    bufs = _gather_buffers.get(key)
    if bufs is None:
        mine = torch.zeros((most,) + out.shape[1:], dtype=torch.uint8)
        parts = [torch.empty_like(mine) for _ in range(world_size)] if rank == dst else None
        bufs = _gather_buffers[key] = (mine, parts)
    mine, parts = bufs
    image = torch.from_numpy(out)  # (a view: rows are copied tensor to tensor, no intermediate arrays)
    own = image[rank::world_size]
    mine[: own.shape[0]].copy_(own)
    if rank == dst:
        dist.gather(mine, parts, dst=dst, group=group)
        for r in range(world_size):
            if r != dst:
                n_rows = len(rows_of(r, world_size, height))
                image[r::world_size].copy_(parts[r][:n_rows])
    else:
        dist.gather(mine, None, dst=dst, group=group)
    return out


def reduce_accum(accum, group=None, dst=0):
    """Sums the ranks' f64 accumulation images (a torch tensor, on the GPU under RCCL) into `dst`'s."""
    import torch.distributed as dist

    dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum

"""Host logic of the multi-GPU split: which 8x8 tiles a rank owns, and the film combine.

One process per GPU.  Pixels are independent and the sampler is a pure function of (pixel, sample),
so the frame is partitioned by interleaved tiles with no data-path exchange while rendering; the
only collective is the final combine: every rank holds a zero-initialised full frame, its owned tiles
are disjoint from everyone else's, and a SUM-reduce to rank 0 is therefore an exact gather
(x + 0 == x).  torch.distributed is the transport (backend "nccl" = RCCL over xGMI on the GPUs,
"gloo" in the CPU tests).
"""
import numpy as np

TILE = 8


def tile_grid(x0, y0, x1, y1):
    """(tx0, ty0, tiles per row, tile rows) of a pixel region -- same arithmetic as dmt_render()."""
    tx0, ty0 = x0 // TILE, y0 // TILE
    tx1, ty1 = (x1 + TILE - 1) // TILE, (y1 + TILE - 1) // TILE
    return tx0, ty0, tx1 - tx0, ty1 - ty0


def owned_tiles(width, height, rank, world, region=None):
    """Tile coordinates (tx, ty) rendered by `rank`: item i of a rank is tile j = rank + i * world of the
    region's row-major tile list (dmt_set_partition / k_megakernel)."""
    x0, y0, x1, y1 = region if region is not None else (0, 0, width, height)
    tx0, ty0, rtx, rty = tile_grid(x0, y0, x1, y1)
    js = np.arange(rank, rtx * rty, world)
    return np.stack([tx0 + js % rtx, ty0 + js // rtx], axis=1)


def owned_pixel_mask(width, height, rank, world, region=None):
    x0, y0, x1, y1 = region if region is not None else (0, 0, width, height)
    mask = np.zeros((height, width), bool)
    for tx, ty in owned_tiles(width, height, rank, world, region):
        mask[max(ty * TILE, y0):min(ty * TILE + TILE, y1), max(tx * TILE, x0):min(tx * TILE + TILE, x1)] = True
    return mask


def combine_films(mean, m2, dst=0, film=None):
    """SUM-reduce the two film planes (torch tensors, full frames, zero outside the owned tiles) to `dst`.
    `film`: optionally the single tensor both planes are views of ([2, h, w, 4]); then one collective does it."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if film is not None:
            dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(mean, dst=dst, op=dist.ReduceOp.SUM)
            dist.reduce(m2, dst=dst, op=dist.ReduceOp.SUM)
    return mean, m2

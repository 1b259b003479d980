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


def pack_owned_tiles(film, rank, world):
    """[2, h, w, 4] film (h, w multiples of 8) -> [ceil(T / world), 512] tensor holding this rank's 8x8 tiles (both planes of a
    tile side by side), in the order of owned_tiles(); the last row is zero padding when the tile count does not divide."""
    import torch
    _, h, w, _ = film.shape
    th, tw = h // TILE, w // TILE
    tiles = film.view(2, th, TILE, tw, TILE, 4).permute(1, 3, 0, 2, 4, 5).reshape(th * tw, 2 * TILE * TILE * 4)
    per = (th * tw + world - 1) // world
    mine = tiles[rank::world]
    if mine.shape[0] == per:
        return mine.contiguous()
    out = torch.zeros((per, tiles.shape[1]), dtype=film.dtype, device=film.device)
    out[:mine.shape[0]] = mine
    return out


def unpack_owned_tiles(film, packs):
    """Inverse of pack_owned_tiles on the destination rank: packs[r] are rank r's tiles; writes them into `film` in place."""
    _, h, w, _ = film.shape
    th, tw = h // TILE, w // TILE
    world = len(packs)
    tiles = film.new_zeros((th * tw, 2 * TILE * TILE * 4))
    for r, p in enumerate(packs):
        n = tiles[r::world].shape[0]
        tiles[r::world] = p[:n]
    film.copy_(tiles.view(th, tw, 2, TILE, TILE, 4).permute(2, 0, 3, 1, 4, 5).reshape(2, h, w, 4))


def combine_films(mean, m2, dst=0, film=None, mode=None):
    """Bring every rank's owned tiles together on `dst`.  Two exact ways (the tile sets are disjoint):

    * "reduce" (default): SUM-reduce of the zero-initialised full frames, x + 0 == x.  One collective, nothing to pack;
      moves the whole frame per rank (RCCL ring: 2 (N-1)/N x frame bytes over the slowest link).
    * "gather" (DMT_COMBINE=gather; SURVEY 8(e)-1's first choice): each rank packs only its own tiles and they are gathered
      on `dst` (RCCL implements gather as grouped ncclSend / ncclRecv): 1/N of the bytes per rank, each over its own xGMI
      link to `dst`.  Needs `film` = the [2, h, w, 4] tensor both planes are views of, with h and w multiples of 8.

    Both leave the complete film on `dst`; tests/test_multigpu_cpu.py proves them bit-identical."""
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return mean, m2
    mode = mode or os.environ.get("DMT_COMBINE", "reduce")
    if mode not in ("reduce", "gather"):
        raise ValueError(f"DMT_COMBINE must be 'reduce' or 'gather', not {mode!r}")
    world, rank = dist.get_world_size(), dist.get_rank()
    if mode == "gather" and film is not None and film.shape[1] % TILE == 0 and film.shape[2] % TILE == 0:
        pack = pack_owned_tiles(film, rank, world)
        packs = [pack.new_empty(pack.shape) for _ in range(world)] if rank == dst else None
        dist.gather(pack, packs, dst=dst)
        if rank == dst:
            unpack_owned_tiles(film, packs)
        return mean, m2
    if film is not None:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(mean, dst=dst, op=dist.ReduceOp.SUM)
        dist.reduce(m2, dst=dst, op=dist.ReduceOp.SUM)
    return mean, m2

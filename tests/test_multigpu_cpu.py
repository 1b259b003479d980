"""CPU test of the N > 1 path: 2 ranks over gloo.  Each rank renders the tiles it owns (with the CPU
oracle standing in for the GPU kernel -- the logic under test is the partition and the combine) into a
zero-initialised full frame; after multigpu.combine_films() rank 0 must hold exactly the single-rank film."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
W, H, SPP = 52, 37, 3      # ragged: not a multiple of the 8x8 tile


def _worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package()
    O = graft.load_oracle()
    scene = O.cornell_box(W, H)
    film = np.zeros((2, H, W, 4), np.float32)     # both planes in one allocation, as bench.py holds them:
    mean, m2 = film[0], film[1]                   # combine_films then needs one collective instead of two
    for tx, ty in pkg.multigpu.owned_tiles(W, H, rank, world):
        O.render(scene, SPP, region=(tx * 8, ty * 8, tx * 8 + 8, ty * 8 + 8), film=(mean, m2), threads=1)
    tf = torch.from_numpy(film)
    tm, tv = tf[0], tf[1]
    pkg.multigpu.combine_films(tm, tv, dst=0, film=tf)

    if rank == 0:
        np.savez(out_path, mean=tm.numpy(), m2=tv.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_and_combine(tmp_path, O, pkg):
    import torch.multiprocessing as mp
    out = tmp_path / "combined.npz"
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, str(out)), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    scene = O.cornell_box(W, H)
    mean, m2 = O.render(scene, SPP, threads=2)
    assert np.array_equal(got["mean"], mean) and np.array_equal(got["m2"], m2)


def _worker_gather(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package()
    w, h = 64, 40                                  # 8 x 5 tiles: 40 tiles do not divide by 3 (padding row in the packs)
    rng = np.random.default_rng(1234)              # every rank draws the SAME full frame and keeps its own tiles of it
    full = rng.standard_normal((2, h, w, 4)).astype(np.float32)
    full[full == 0] = 1.0
    mask = pkg.multigpu.owned_pixel_mask(w, h, rank, world)
    mine = np.where(mask[None, :, :, None], full, np.float32(0))
    results = {}
    for mode in ("reduce", "gather"):
        tf = torch.from_numpy(mine.copy())
        pkg.multigpu.combine_films(tf[0], tf[1], dst=0, film=tf, mode=mode)
        results[mode] = tf.numpy().copy()
    os.environ["DMT_COMBINE"] = "gather"           # the environment knob bench.py's users set
    tf = torch.from_numpy(mine.copy())
    pkg.multigpu.combine_films(tf[0], tf[1], dst=0, film=tf)
    results["env"] = tf.numpy().copy()
    if rank == 0:
        np.savez(out_path, full=full, **results)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_combine_equals_reduce_bit_for_bit(tmp_path, world):
    """DMT_COMBINE=gather (pack own tiles -> gather on rank 0 -> unpack) leaves exactly the film the
    default SUM-reduce of zero-initialised frames leaves, and both equal the frame the tiles were cut from."""
    import torch.multiprocessing as mp
    out = tmp_path / f"gather{world}.npz"
    port = 31500 + (os.getpid() % 2000) + world
    mp.start_processes(_worker_gather, args=(world, port, str(out)), nprocs=world, join=True, start_method="spawn")
    got = np.load(out)
    assert np.array_equal(got["reduce"].view(np.uint32), got["full"].view(np.uint32))
    assert np.array_equal(got["gather"].view(np.uint32), got["reduce"].view(np.uint32))
    assert np.array_equal(got["env"].view(np.uint32), got["reduce"].view(np.uint32))


def test_pack_unpack_round_trip(pkg):
    import torch
    w, h, world = 72, 56, 5
    full = torch.randn((2, h, w, 4))
    packs = [pkg.multigpu.pack_owned_tiles(full, r, world) for r in range(world)]
    tiles = (w // 8) * (h // 8)
    assert all(p.shape == ((tiles + world - 1) // world, 512) for p in packs)
    # a rank's pack holds exactly the pixels of its ownership mask (the kernel's tile map, owned_pixel_mask)
    for r in range(world):
        alone = torch.zeros_like(full)
        pkg.multigpu.unpack_owned_tiles(alone, [p if q == r else torch.zeros_like(p) for q, p in enumerate(packs)])
        mask = torch.from_numpy(pkg.multigpu.owned_pixel_mask(w, h, r, world))[None, :, :, None]
        assert torch.equal(alone, torch.where(mask, full, torch.zeros(())))
    out = torch.zeros_like(full)
    pkg.multigpu.unpack_owned_tiles(out, packs)
    assert torch.equal(out, full)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_partition_covers_every_pixel_once(pkg, world):
    for (w, h, region) in ((52, 37, None), (1024, 1024, None), (200, 120, (13, 9, 150, 77))):
        total = np.zeros((h, w), np.int32)
        for r in range(world):
            total += pkg.multigpu.owned_pixel_mask(w, h, r, world, region)
        x0, y0, x1, y1 = region if region else (0, 0, w, h)
        inside = np.zeros((h, w), bool)
        inside[y0:y1, x0:x1] = True
        assert np.array_equal(total, inside.astype(np.int32))
        counts = [len(pkg.multigpu.owned_tiles(w, h, r, world, region)) for r in range(world)]
        assert max(counts) - min(counts) <= 1          # interleaving balances the tile counts

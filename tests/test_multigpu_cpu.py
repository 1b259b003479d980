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

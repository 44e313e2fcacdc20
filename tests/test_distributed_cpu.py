"""The N>1 path on CPU: world_size-2 gloo, row tiles rendered by the oracle (stand-in for the
GPU renderer), one gather, image identical to a single full render."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lasgun_amd.distributed import InterleavedGather, gather_tiles, interleaved_rows, max_tile_rows, row_tile

W, H = 40, 37  # odd height: tiles of 19 and 18 rows


def test_row_tiles_cover_exactly():
    for world in (1, 2, 3, 4, 8):
        for h in (1, 7, 8, 37, 4096):
            rows = [row_tile(r, world, h) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            for a, b in zip(rows, rows[1:]):
                assert a[1] == b[0]
            assert max(y1 - y0 for y0, y1 in rows) == max_tile_rows(world, h)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_lib import oracle
    from lasgun_amd import scenes as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = oracle()
        acc = o.Accel(S.cornell_scene(o, "plastic"))
        # render this rank's rows through capture_subset semantics: full oracle film, keep the tile
        film = o.Film(W, H)
        o.capture_subset(0, 1, acc, film)
        y0, y1 = row_tile(rank, world, H)
        tile = torch.from_numpy(film.pixels()[y0:y1].copy())
        full = gather_tiles(tile, W, H, rank, world)
        if rank == 0:
            np.save(out_path, full.numpy())
        else:
            assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather(tmp_path):
    out = str(tmp_path / "full.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    from oracle_lib import oracle
    from lasgun_amd import scenes as S
    o = oracle()
    want = o.render(S.cornell_scene(o, "plastic"), (W, H)).pixels()
    assert np.array_equal(np.load(out), want)


def test_interleaved_rows_partition():
    for world in (1, 2, 4, 8):
        h, b = 64 * world, 8
        seen = sorted(y for r in range(world) for y in interleaved_rows(r, world, h, b))
        assert seen == list(range(h))
        assert all(len(interleaved_rows(r, world, h, b)) == h // world for r in range(world))


def _ilv_worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_lib import oracle
    from lasgun_amd import scenes as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = oracle()
        w, h, b = 24, 32, 4
        acc = o.Accel(S.cornell_scene(o, "glass"))
        film = o.Film(w, h)
        o.capture_subset(0, 1, acc, film)
        mine = torch.from_numpy(film.pixels()[interleaved_rows(rank, world, h, b)].copy())
        ig = InterleavedGather(w, h, rank, world, b, "cpu")
        for frame in range(3):  # three frames through the two-buffer pipeline
            t = ig.tile()
            t.copy_(mine if frame == 2 else torch.zeros_like(mine))
            ig.submit()
        full = ig.finish()
        if rank == 0:
            np.save(out_path, full.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_interleaved_gather_pipeline(tmp_path):
    out = str(tmp_path / "ilv.npy")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_ilv_worker, args=(2, port, out), nprocs=2, join=True)
    from oracle_lib import oracle
    from lasgun_amd import scenes as S
    o = oracle()
    want = o.render(S.cornell_scene(o, "glass"), (24, 32)).pixels()
    assert np.array_equal(np.load(out), want)


def _allgather_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_lib import oracle
    from lasgun_amd import scenes as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = oracle()
        w, h, b = 24, 32, 4
        acc = o.Accel(S.cornell_scene(o, "glass"))
        film = o.Film(w, h)
        o.capture_subset(0, 1, acc, film)
        mine = torch.from_numpy(film.pixels()[interleaved_rows(rank, world, h, b)].copy())
        ig = InterleavedGather(w, h, rank, world, b, "cpu", all_ranks=True)
        for frame in range(3):
            t = ig.tile()
            t.copy_(mine if frame == 2 else torch.zeros_like(mine))
            ig.submit()
        full = ig.finish()
        assert full is not None  # EVERY rank owns a film
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_all_gather_gives_every_rank_the_film(tmp_path):
    """The all-gather form of the end-of-frame exchange: ONE collective, every rank ends with the film."""
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_allgather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from oracle_lib import oracle
    from lasgun_amd import scenes as S
    o = oracle()
    want = o.render(S.cornell_scene(o, "glass"), (24, 32)).pixels()
    for rank in range(2):
        assert np.array_equal(np.load(str(tmp_path / ("rank%d.npy" % rank))), want), rank


@pytest.mark.parametrize("world", [4, 8])
def test_more_ranks_interleaved_gather_pipeline(tmp_path, world):
    """The driver's N = 4 and N = 8 layouts rehearsed on CPU: 4-row blocks dealt round-robin over `world` gloo ranks (two groups of
    blocks per rank at 4, one at 8), three frames through the buffer rotation, ONE gather per frame, film == the oracle's."""
    out = str(tmp_path / "ilv.npy")
    port = 35500 + (os.getpid() % 2000) + world
    mp.spawn(_ilv_worker, args=(world, port, out), nprocs=world, join=True)
    from oracle_lib import oracle
    from lasgun_amd import scenes as S
    o = oracle()
    want = o.render(S.cornell_scene(o, "glass"), (24, 32)).pixels()
    assert np.array_equal(np.load(out), want)


def test_four_rank_all_gather_gives_every_rank_the_film(tmp_path):
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_allgather_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    from oracle_lib import oracle
    from lasgun_amd import scenes as S
    o = oracle()
    want = o.render(S.cornell_scene(o, "glass"), (24, 32)).pixels()
    for rank in range(4):
        assert np.array_equal(np.load(str(tmp_path / ("rank%d.npy" % rank))), want), rank

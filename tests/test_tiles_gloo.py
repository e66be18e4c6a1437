"""CPU tests of the multi-GPU path's host logic with world_size 2 and 3 on the gloo backend: tile deal, padded gather,
scatter into planes, scene broadcast.  The per-rank 'renderer' here is the oracle (test infrastructure) cut to the
rank's tiles; on the GPU box the same functions move the HIP kernel's tile buffers over RCCL (bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden_scene
from opencl_render_amd import raytrace as R, tiles as T


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tile_buffer_from_planes(planes, width, height, ids):
    tiles_x = (width + R.TILE - 1) // R.TILE
    buf = np.zeros((len(ids), 3, R.TILE, R.TILE), np.uint16)
    for slot, t in enumerate(ids):
        ty, tx = divmod(int(t), tiles_x)
        y0, x0 = ty * R.TILE, tx * R.TILE
        h, w = min(R.TILE, height - y0), min(R.TILE, width - x0)
        for c in range(3):
            buf[slot, c, :h, :w] = planes[c][y0:y0 + h, x0:x0 + w]
    return buf


def _worker(rank, world, port, name, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as O
        sc, want = (load_golden_scene(name) if rank == 0 else (None, None))
        sc = T.broadcast_scene(sc, rank, torch.device("cpu"))          # every rank gets the scene from rank 0
        ids = R.tiles_of_rank(sc.width, sc.height, rank, world)
        planes = O.oracle_render(sc)                                     # stand-in renderer (CPU); rank keeps its tiles only
        local = torch.from_numpy(_tile_buffer_from_planes(planes, sc.width, sc.height, ids).view(np.uint8).reshape(-1).copy())
        gathered = T.gather_tiles(local, sc.width, sc.height, rank, world)
        if rank == 0:
            got = T.detile_host(gathered.numpy(), sc.width, sc.height, world)
            ok = all(np.array_equal(g, w) for g, w in zip(got, want))
            np.save(result_path, np.array([int(ok), gathered.shape[0], gathered.shape[1]]))
        else:
            assert gathered is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_and_detile_reproduce_the_frame(world, tmp_path):
    result = str(tmp_path / "result.npy")
    mp.spawn(_worker, args=(world, _free_port(), "odd_size_multi_tile", result), nprocs=world, join=True)
    ok, rows, cap = np.load(result)
    assert ok == 1 and rows == world
    assert cap == T.max_tiles_per_rank(200, 150, world) * T.TILE_BYTES


def test_tile_deal_covers_every_tile_once():
    for (w, h) in [(1920, 1080), (3840, 2160), (200, 150), (128, 128), (1, 1)]:
        n = R.tile_count(w, h)
        for world in (1, 2, 4, 8):
            seen = np.concatenate([R.tiles_of_rank(w, h, r, world) for r in range(world)])
            assert sorted(seen.tolist()) == list(range(n))
            sizes = [len(R.tiles_of_rank(w, h, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == T.max_tiles_per_rank(w, h, world)

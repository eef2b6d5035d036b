'''
world_size-2 gloo test of the slab-tiling host logic: each rank renders its column slab (with the
CPU oracle -- there is no GPU here), the slabs are gathered to rank 0, and the result must be
bit-identical to the single-rank render.
'''

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_bounds_partition():
    from ptina_amd.dist import slab_bounds
    for nx in (1, 7, 512, 2048, 100):
        for world in (1, 2, 3, 8):
            b = [slab_bounds(nx, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == nx
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(x1 - x0 for x0, x1 in b) - min(x1 - x0 for x0, x1 in b) <= 1


def test_stripe_columns_partition():
    from ptina_amd.dist import stripe_columns
    for nx in (1, 7, 50, 512, 2048):
        for world in (1, 2, 3, 8):
            cols = [stripe_columns(nx, world, r) for r in range(world)]
            assert sorted(np.concatenate(cols).tolist()) == list(range(nx))
            if nx >= 16 * world * 4:
                assert max(map(len, cols)) - min(map(len, cols)) <= 16


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import oracle
    from ptina_amd import scenes
    from ptina_amd.dist import slab_bounds, gather_film_torch
    from helpers import setup_oracle
    dist.init_process_group('gloo', rank=rank, world_size=world)
    nx, ny, spp = 30, 20, 2
    o = setup_oracle(oracle, scenes.scene_s34(), nx, ny, threads=1)
    o.set_window(*slab_bounds(nx, world, rank))
    o.render(spp)
    film = gather_film_torch(o.get_film_raw(), nx, ny, rank, world)
    # the striped split of the same film (ragged: 70 columns = 4 stripes of 16 + one of 6)
    nx2 = 70
    o2 = setup_oracle(oracle, scenes.scene_s34(), nx2, ny, threads=1)
    o2.set_stripes(16, rank, world)
    o2.render(spp)
    film2 = gather_film_torch(o2.get_film_raw(), nx2, ny, rank, world, stripe=16)
    if rank == 0:
        np.save(out, film)
        np.save(out + '.stripes.npy', film2)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slab_gather_is_bit_identical(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    from ptina_amd import scenes
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import setup_oracle
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'film.npy')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    tiled = np.load(out)
    o = setup_oracle(oracle_mod, scenes.scene_s34(), 30, 20, threads=1)
    o.render(2)
    full = o.get_film_raw()
    assert np.array_equal(tiled, full)
    assert np.all(tiled[:, 3] == 2.0)
    striped = np.load(out + '.stripes.npy')
    o = setup_oracle(oracle_mod, scenes.scene_s34(), 70, 20, threads=1)
    o.render(2)
    assert np.array_equal(striped, o.get_film_raw())
    assert np.all(striped[:, 3] == 2.0)


def test_unique_id_file_rendezvous(tmp_path, monkeypatch):
    from ptina_amd import dist as D
    monkeypatch.setenv('MIPTINA_RDZV_DIR', str(tmp_path))
    monkeypatch.setenv('MASTER_PORT', '12345')
    uid = bytes(range(128))
    assert D.exchange_unique_id(lambda: uid, 0, 2) == uid
    assert D.exchange_unique_id(lambda: b'', 1, 2, timeout=2.0) == uid
    with pytest.raises(RuntimeError):
        monkeypatch.setenv('MASTER_PORT', '54321')
        D.exchange_unique_id(lambda: b'', 1, 2, timeout=0.2)

"""N > 1 path on CPU: two gloo ranks each gather their own share of the image -- the 4x4-pixel tiles dealt
round-robin, as bench.py shards -- (oracle standing in for the device kernel), all-reduce the accumulators
and must reproduce the single-rank film (disjoint supports: sum == gather)."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    import oracle_lib as O
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tile = 16
    tx, ty = bench.tile_grid(world)
    c = cases.make_case("cbox", tile * tx, tile * ty, 4000, 5.0)
    acc = None
    for it in (1, 2):
        ph, nb = c.sc.shoot_photons(it, 4000)           # photon map replicated on every rank
        rays = c.sc.camera_beams_interleaved(it, world, rank)  # beam sets sharded: 4x4 tiles round-robin
        acc, _, _ = O.gather_bre(c.p, c.m, c.tris, ph, rays, c.r, it, nb, 64, accum=acc)
    yy, xx = np.mgrid[0:acc.shape[0], 0:acc.shape[1]]
    own = ((yy // 4) * ((acc.shape[1] + 3) // 4) + xx // 4) % world == rank
    assert not acc[~own].any() and acc[own].any()
    t = torch.from_numpy(acc.copy())
    dist.all_reduce(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "film.npy"), t.numpy())
    dist.destroy_process_group()


def test_two_rank_tile_sharding_equals_single_rank(tmp_path):
    import cases
    import oracle_lib as O
    import bench
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    film = np.load(tmp_path / "film.npy")
    tx, ty = bench.tile_grid(world)
    c = cases.make_case("cbox", 16 * tx, 16 * ty, 4000, 5.0)
    acc = None
    for it in (1, 2):
        ph, nb = c.sc.shoot_photons(it, 4000)
        rays = c.sc.camera_beams(it)
        acc, _, _ = O.gather_bre(c.p, c.m, c.tris, ph, rays, c.r, it, nb, 64, accum=acc)
    assert np.array_equal(film, acc)   # bit-identical: per-pixel sums do not depend on the sharding
    thr, dx, dy = O.assemble(film, 2, True)
    assert np.isfinite(thr).all() and np.abs(dx).max() > 0

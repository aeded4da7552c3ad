"""N>1 path on CPU: world_size-2 gloo run of the frame sharding + map-point gather used by bench.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vslam_amd.sharding import gather_map_points, shard

CAP, B = 16, 5


def fake_points(first_pair, n_pairs, rows):
    """Deterministic stand-in for the two-view output of pair g: depends only on the GLOBAL pair index."""
    out = torch.full((rows, CAP, 3), float("nan"))
    for i in range(n_pairs):
        g = first_pair + i
        k = (g * 7) % CAP
        out[i, :k] = torch.arange(k * 3, dtype=torch.float32).reshape(k, 3) + 1000.0 * g
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n, n_pairs, first_pair = shard(rank, world, B)
    pts = fake_points(first_pair, n_pairs, B)
    got = gather_map_points(pts, n_pairs, dst=0)
    # the caller-supplied pair counts (what bench.py passes) must give the same lists without the count exchange
    got2 = gather_map_points(pts, n_pairs, dst=0, pairs_per_rank=[shard(r, world, B)[2] for r in range(world)])
    # the overlapped form bench.py uses for N > 1: two gathers in flight on two buffers, waited for in order
    ppr = [shard(r, world, B)[2] for r in range(world)]
    pts_b = pts.clone()
    w1 = gather_map_points(pts, n_pairs, dst=0, pairs_per_rank=ppr, async_op=True)
    w2 = gather_map_points(pts_b, n_pairs, dst=0, pairs_per_rank=ppr, async_op=True)
    got3, got4 = w1[1](), w2[1]()
    if rank == 0:
        assert all(torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) for a, b in zip(got, got2)) and len(got) == len(got2)
        for g in (got3, got4):
            assert len(g) == len(got) and all(torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) for a, b in zip(got, g))
        q.put(torch.cat(got).numpy())
    else:
        assert got is None and got2 is None and got3 is None and got4 is None
    dist.barrier()
    dist.destroy_process_group()


def test_shards_tile_the_sequence():
    for world in (1, 2, 4, 8):
        pairs = []
        for r in range(world):
            first, n, n_pairs, first_pair = shard(r, world, 256)
            assert n == 256 + (r > 0) and first == r * 256 - (r > 0)
            pairs += list(range(first_pair, first_pair + n_pairs))
        assert pairs == list(range(world * 256 - 1))


def test_gather_world2_equals_single_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = fake_points(0, 2 * B - 1, 2 * B - 1).numpy()
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(ref))


def test_async_gather_in_a_world_of_one_has_the_same_return_shape():
    """async_op=True returns (work, finish) whatever the world size: `work, finish = gather_map_points(..., async_op=True)` must not
    depend on how many processes run."""
    pts = fake_points(0, B - 1, B)
    work, finish = gather_map_points(pts, B - 1, async_op=True)
    assert work is None
    got = finish()
    ref = gather_map_points(pts, B - 1)
    assert len(got) == len(ref) == 1 and torch.equal(torch.nan_to_num(got[0]), torch.nan_to_num(ref[0]))

"""Frame sharding for the batched multi-GPU mode (one process per GPU, torch.distributed over RCCL/xGMI).

Frames are independent and matching/pose work is per consecutive pair (Tracker matches the current frame against
the last one, reference tracker.py:214), so a global frame sequence is cut into contiguous shards.  Rank r > 0
re-extracts the single frame preceding its shard (1-frame halo, ~4.5 MB of recomputation instead of a P2P hop);
there is no data-path collective.  The only exchange is the final gather of map points to rank 0.
"""
import torch
import torch.distributed as dist


def shard(rank, world, frames_per_rank):
    """-> (first_frame, n_frames, n_pairs, first_pair): the frames rank `rank` extracts (halo included) and the
    global indices of the pairs (i, i+1) it owns.  Pairs of all ranks tile [0, world*frames_per_rank - 1)."""
    halo = 1 if rank > 0 else 0
    first = rank * frames_per_rank - halo
    n = frames_per_rank + halo
    return first, n, n - 1, first


def gather_map_points(points, n_pairs, dst=0, group=None, pairs_per_rank=None, async_op=False):
    """points: [rows >= n_pairs, cap, 3] float tensor, NaN where a query keypoint produced no map point.
    Padded gather to `dst` (fixed-size collective: payload is MBs, latency-bound on xGMI).  On dst returns a list
    with one [n_pairs_r, cap, 3] tensor per rank in global pair order, elsewhere None.
    pairs_per_rank: the pair count of every rank when the caller knows it (contiguous sharding does: shard(r, ...)[2]);
    without it the counts are exchanged with one extra all_gather and a host read per rank.
    async_op: -> (work, finish) whatever the world size; finish() returns what the synchronous call returns (work is None in a
    world of one: nothing is in flight)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        res = [points[:n_pairs]]
        return (None, lambda: res) if async_op else res
    rank = dist.get_rank(group)
    if pairs_per_rank is None:
        meta = torch.tensor([n_pairs], dtype=torch.int64, device=points.device)
        metas = [torch.zeros_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta, group=group)
    bufs = [torch.empty_like(points) for _ in range(world)] if rank == dst else None
    if async_op:
        # the collective runs on the backend's own stream / thread beside whatever the caller enqueues next; `points` must stay
        # untouched until work.wait() (RCCL: a stream-level wait, the host does not block).  -> (work, finish) where finish()
        # returns what the synchronous call returns
        work = dist.gather(points, bufs, dst=dst, group=group, async_op=True)

        def finish():
            work.wait()
            if rank != dst:
                return None
            ppr = pairs_per_rank if pairs_per_rank is not None else [int(m.item()) for m in metas]
            return [bufs[r][:ppr[r]] for r in range(world)]
        return work, finish
    dist.gather(points, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    if pairs_per_rank is None:
        pairs_per_rank = [int(m.item()) for m in metas]
    return [bufs[r][:pairs_per_rank[r]] for r in range(world)]

"""GPU: the N > 1 path with REAL pipeline output.  (1) the C-ABI gather (mo_comm_* / mo_gather_map_points, RCCL) on a world of one -
all a one-GPU box can host, RCCL refuses two ranks on one device; (2) two ranks sharing the GPU, frames sharded as bench.py
shards them, map points gathered with torch.distributed over gloo: rank 0 must hold exactly what one process computes for the
whole sequence (the sharding, the 1-frame halo and the per-pair seeds are what is under test, not the transport)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CAP, K9 = 1024, [320.0, 0, 320.0, 0, 320.0, 240.0, 0, 0, 1.0]


def _run_batch(torch, V, frames, first_global_pair=0):
    """frames [n, 480, 640] uint8 numpy -> points [n-1, CAP, 3] float32 tensor (cuda), n_points"""
    dev = torch.device("cuda", 0)
    n = len(frames)
    ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=n)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    prm = V.orb_params(nfeatures=1000)
    d_fr = torch.from_numpy(frames).to(dev)
    z = lambda *s, dt=torch.int32: torch.zeros(s, dtype=dt, device=dev)
    kps = z(n, CAP, 7, dt=torch.float32); desc = z(n, CAP, 32, dt=torch.uint8); counts = z(n)
    midx = z(n - 1, CAP, 2); mdist = z(n - 1, CAP, 2); mpass = z(n - 1, CAP, dt=torch.uint8)
    pose = z(n - 1, 12, dt=torch.float64); pts = z(n - 1, CAP, 3, dt=torch.float32); npts = z(n - 1)
    io = V.BatchIO()
    io.d_gray = d_fr.data_ptr(); io.w = 640; io.h = 480; io.batch = n; io.cap = CAP
    io.ratio = 0.75; io.thr_px = 3.0; io.n_hyp = 512; io.seed = 4096; io.pair_index_base = first_global_pair
    for i in range(9): io.K[i] = K9[i]
    io.d_kps = kps.data_ptr(); io.d_desc = desc.data_ptr(); io.d_counts = counts.data_ptr()
    io.d_match_idx = midx.data_ptr(); io.d_match_dist = mdist.data_ptr(); io.d_match_pass = mpass.data_ptr()
    io.d_pose = pose.data_ptr(); io.d_points = pts.data_ptr(); io.d_n_points = npts.data_ptr()
    ctx._check(ctx.lib.mo_dev_frontend_batch(ctx.h, C.byref(prm), C.byref(io)))
    torch.cuda.synchronize()
    assert ctx.dev_status() == 0
    return ctx, pts, npts, pose


def test_cabi_gather_world_of_one():
    import torch
    import vslam_amd as V
    from tests.helpers import parallax_frames
    frames = parallax_frames(4, seed=3, bg_step=8, fg_step=16)
    ctx, pts, npts, _ = _run_batch(torch, V, frames)
    try:
        uid = V.Context.comm_unique_id()
    except V.NativeUnavailable as e:
        pytest.skip(str(e))
    ctx.comm_init(uid, 0, 1)
    rows_max = 5
    local = torch.full((rows_max, CAP, 3), float("nan"), dtype=torch.float32, device=pts.device)
    local[:3] = pts
    out = torch.zeros((1, rows_max, CAP, 3), dtype=torch.float32, device=pts.device)
    rows = torch.zeros(1, dtype=torch.int32, device=pts.device)
    ctx.gather_map_points(local.data_ptr(), 3, rows_max, CAP, 0, out.data_ptr(), rows.data_ptr())
    ctx.sync()
    assert int(rows[0].item()) == 3
    a, b = out[0, :3].cpu().numpy(), pts.cpu().numpy()
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))
    assert int(npts.sum().item()) > 100
    ctx.close()


def _worker(rank, world, port, per_rank, q):
    import torch
    import torch.distributed as dist
    import vslam_amd as V
    from tests.helpers import parallax_frames
    from vslam_amd.sharding import gather_map_points, shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n, n_pairs, first_pair = shard(rank, world, per_rank)
    frames = parallax_frames(world * per_rank, seed=3, bg_step=8, fg_step=16)[first:first + n]
    ctx, pts, npts, pose = _run_batch(torch, V, frames, first_global_pair=first_pair)
    rows = torch.full((per_rank, CAP, 3), float("nan"), dtype=torch.float32)
    rows[:n_pairs] = pts.cpu()
    got = gather_map_points(rows, n_pairs, dst=0, pairs_per_rank=[shard(r, world, per_rank)[2] for r in range(world)])
    if rank == 0:
        q.put(torch.cat(got).numpy())
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def test_two_ranks_on_one_gpu_equal_one_process():
    """Every pair carries its GLOBAL index into the sampler (mo_batch_io.pair_index_base), so the sharded run is bit-identical to the
    single-process run: same keypoints (frames are independent), same matches, same hypotheses, same map points."""
    import torch
    import torch.multiprocessing as mp
    import vslam_amd as V
    from tests.helpers import parallax_frames
    world, per_rank = 2, 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for p in procs: p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    frames = parallax_frames(world * per_rank, seed=3, bg_step=8, fg_step=16)
    ctx, pts, npts, _ = _run_batch(torch, V, frames)
    ref = pts.cpu().numpy()
    assert got.shape == ref.shape == (world * per_rank - 1, CAP, 3)
    assert (~np.isnan(ref[..., 0])).sum() > 1000
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(ref))
    ctx.close()


def test_bench_launch_contract_two_ranks_gloo():
    """bench.py exactly as the driver starts it for N > 1 (torch.distributed.run, one rank per process), rehearsed with two gloo ranks
    sharing this box's GPU: every rank must run the same number of collectives (a time-based untimed loop once did not and hung), rank 0
    prints ONE JSON line with the contract's keys, and the batch of 8k + 1 frames a rank > 0 extracts goes through the XCD mapping."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
           "--batch", "16", "--prewarm-ms", "30"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 16 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01  # whole-job frames / time


def test_bench_rccl_gather_path_on_a_world_of_one():
    """bench.py with BENCH_FORCE_GATHER=1: the N > 1 gather path of the benchmark - mo_comm_unique_id / mo_comm_init, mo_gather_map_points
    on a side stream beside the next step, two alternating map-point buffers, the drained gathers inside the timed region and rank 0's
    checks of the gathered slab - on a world of one (RCCL refuses two ranks on one device; the multi-GPU box is the driver's)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_FORCE_GATHER="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "32", "--prewarm-ms", "20",
                          "--no-cpu-baseline", "--no-optin", "--no-extras"], capture_output=True, text=True, timeout=400, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    if d["config"]["rccl_ranks"] == 0:
        pytest.skip("RCCL could not be loaded on this box: " + out.stderr[-300:])
    assert d["config"]["rccl_ranks"] == 1 and d["n_gpus"] == 1 and d["value"] > 0
    assert "mo_gather_map_points" in d["config"]["parallelism"]

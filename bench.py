#!/usr/bin/env python3
"""bench.py -- frames/sec (extract + match + pose) at 640x480, 2000 ORB, on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic frames already resident in HBM:
extract every frame (8-level pyramid, FAST-9, retainBest x2, Harris, IC angle, blur, rBRIEF), match every
consecutive frame pair (brute-force Hamming 2-NN + ratio test), two-view pose + map points per pair
(8-point E RANSAC over 4096 hypotheses, pose recovery, DLT), then the final map-point gather to rank 0.

Workload: BASELINE.json configs[2] "batch of 256 synthetic 640x480 frames, extract+match pipeline" plus the two-view stage of
configs[3] on every pair, on the scene SURVEY.md 8d specifies (--scene survey8d, vslam_amd/synth.py: 8-px 0..255 texture, 400
rectangles, N(0,3) noise, sub-pixel pan on two depth layers, <= 3 deg roll between the frames of a pair); rounds 1 - 2's easier
scene is timed as a side leg (`scene_smooth`).  N > 1: frames are independent, so the global frame sequence is sharded
contiguously, the SAME 256 frames per rank at every N (weak scaling; the N = 1 point is the single-GPU headline); each rank re-extracts
the one frame preceding its shard (halo) instead of receiving it, and the only collective is the gather of map points to rank 0 -
under --backend nccl through the library's own RCCL entry point (mo_comm_init / mo_gather_map_points) on a side stream beside
the next step's kernels, drained inside the timed region.  At N > 1 a second timed region with 512 frames per rank is reported
beside the headline (`config5_512_per_gpu`: 8 ranks = BASELINE config 5's 4096 frames).

Launch: python bench.py [--gpus N --steps K --warmup W]; for N>1 via torch.distributed.run (one rank per GPU).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd"))

W, H, NFEAT, CAP = 640, 480, 2000, 2048
N_HYP = 4096
PEAK_HBM_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PEAK_HBM_MEASURED_GBPS = 6290.0  # measured copy rate of the same guide (SURVEY 8d asks for both)
PMC_FILE = "profiles/r04_pmc_per_kernel.json"  # rocprofv3 --pmc passes of profiles/collect.sh on this build and scene
# SURVEY 8d's whole-pipeline byte models: stage-materialised extraction (pyramid and blurred pyramid each written once and read once),
# the fused-ideal lower bound (input + outputs only), and the matcher's bytes per pair
SURVEY_EXTRACT_BYTES, SURVEY_FUSED_IDEAL_BYTES, SURVEY_MATCH_BYTES = 4541474, 427200, 162000

# algorithmic bytes per frame of each extraction stage (stage-materialised model)
STAGE_BYTES = {
    "pyramid": 926546 + 643332,      # resize reads L0..6 + writes L1..7
    "fast_nms": 950532,              # FAST reads L0..7
    "select_harris": 0,              # candidate lists only (latency-bound replay), priced at 0 algorithmic bytes
    "blur": 950532 + 950532,         # reads + writes of the blurred pyramid
    "angle_rbrief": 950532 + 120000, # descriptor reads + keypoint/descriptor outputs
    "match_knn2_ratio": 162000,      # per PAIR: 2 x 64000 B descriptors in + 34000 B out
    "two_view": 32000,               # per PAIR: correspondences in (<= 2000 x 16 B)
}
STAGE_KERNELS = {"pyramid": ["k_resize2", "k_resize"], "fast_nms": ["k_fast"], "select_harris": ["k_select"], "blur": ["k_blur"],
                 "angle_rbrief": ["k_describe_tiles", "k_describe_tiles_rare"], "match_knn2_ratio": ["k_match_lds"],
                 "two_view": ["k_tv_prep", "k_tv_hyp", "k_tv_tasks", "k_tv_score", "k_tv_finish"]}


def make_frames(torch, device, first, count, scene="survey8d", seed=20250523):
    """frames [first, first + count) of the global synthetic sequence (vslam_amd/synth.py), uint8 [count, H, W] on `device`"""
    from vslam_amd import synth
    return synth.make_frames(torch, device, first, count, scene=scene, seed=seed)


def metric_name():
    """BASELINE.json's metric string when the file is present (it is part of the repo), else the same wording."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "frames/sec (extract+match+pose) at 640x480, 2000 ORB, 1/2/4/8 MI355X"


def cv2_baseline(frames_u8, n_frames, K):
    """if cv2 happens to be importable on the GPU box, time cv2 itself with the reference's parameters
    (extractor.py:38-48, matcher.py:29,70-81, utils.py:120-134) -- the calls the reference makes, not its files."""
    import cv2
    cv2.setNumThreads(1)
    orb = cv2.ORB_create(nfeatures=NFEAT, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2,
                         scoreType=cv2.ORB_HARRIS_SCORE, patchSize=31, fastThreshold=7)
    bf = cv2.BFMatcher(cv2.NORM_HAMMING)
    t0 = time.perf_counter()
    feats = [orb.detectAndCompute(frames_u8[i], None) for i in range(n_frames)]
    t1 = time.perf_counter()
    good = []
    for i in range(n_frames - 1):
        knn = bf.knnMatch(feats[i][1], feats[i + 1][1], k=2)
        good.append([m[0] for m in knn if len(m) == 1 or m[0].distance < 0.75 * m[1].distance])
    t2 = time.perf_counter()
    n_pose = min(12, n_frames - 1)
    for i in range(n_pose):
        p1 = np.float32([feats[i][0][m.queryIdx].pt for m in good[i]])
        p2 = np.float32([feats[i + 1][0][m.trainIdx].pt for m in good[i]])
        E, mask = cv2.findEssentialMat(p1, p2, K, method=cv2.RANSAC, prob=0.999, threshold=3.0)
        if E is not None and E.shape == (3, 3):
            cv2.recoverPose(E, p1, p2, K, mask=mask)
    t3 = time.perf_counter()
    per_frame = (t1 - t0) / n_frames + (t2 - t1) / max(n_frames - 1, 1) + (t3 - t2) / max(n_pose, 1)
    return {"value": 1.0 / per_frame, "unit": "frames/s", "cores": 1, "kind": "reference",
            "sample": "%d frames cv2.ORB (%.3f s), %d pairs BFMatcher knn + ratio (%.3f s), %d pairs findEssentialMat + "
                      "recoverPose (%.3f s); cv2 %s, 1 thread" % (n_frames, t1 - t0, n_frames - 1, t2 - t1, n_pose, t3 - t2,
                                                                  cv2.__version__)}


def _baseline_lib():
    """oracle/_build/libcpu_baseline*.so: rebuilt with -march=native ON this box when a compiler is present (the shipped build
    targets x86-64-v3 because it is compiled in another container)."""
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    native = os.path.join(odir, "_build", "libcpu_baseline_native.so")
    flags = "-O3 -march=x86-64-v3 -fopenmp (built in the build container)"
    path = os.path.join(odir, "_build", "libcpu_baseline.so")
    try:
        subprocess.run(["make", "-C", odir, "native"], check=True, capture_output=True, timeout=300)
        path, flags = native, "-O3 -march=native -fopenmp (built on this box)"
    except Exception:
        pass
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.orc_baseline_run.restype = C.c_int
    lib.orc_baseline_run.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    return lib, flags


def cpu_baseline(frames_u8, n_frames, K):
    """cv2 itself when the box has it (kind "reference"); otherwise the repo's own CPU restatement (kind "port",
    B4): oracle/cpu_baseline.cpp = the parity oracle's extract + match plus a C++ two-view stage, -O3, one OpenMP thread per host
    core over frames / pairs, on a bounded sample of the SAME batch the GPU step processes; a 1-thread run on 8 frames beside it."""
    try:
        import cv2  # noqa: F401
    except ImportError:
        cv2 = None
    if cv2 is not None:
        try:
            return cv2_baseline(frames_u8, n_frames, K)
        except Exception as e:  # an unexpected cv2 build: fall back to the port and say so
            print("cv2 baseline failed (%s); timing the CPU restatement instead" % e, file=sys.stderr)
    lib, flags = _baseline_lib()
    fr = np.ascontiguousarray(frames_u8[:n_frames])
    Kc = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))

    def run(n, threads):
        times = (C.c_double * 3)(); counts = (C.c_longlong * 3)()
        used = lib.orc_baseline_run(fr.ctypes.data, n, W, H, NFEAT, 0.75, n - 1, Kc.ctypes.data, N_HYP, threads, C.addressof(times),
                                    C.addressof(counts))
        return used, list(times), list(counts)

    # one GPU's share of the host: the pool's boxes give a 1-GPU job 16 of the host's cores (a dedicated 8-GPU node would give
    # each GPU an eighth); os.cpu_count() is reported beside it
    cores = min(os.cpu_count() or 1, int(os.environ.get("VSLAM_AMD_CPU_THREADS", "16")))
    used, t, cnt = run(n_frames, cores)
    n1 = min(8, n_frames)
    _, t1, _ = run(n1, 1)
    return {"value": n_frames / sum(t), "unit": "frames/s", "cores": used, "host_cpu_count": os.cpu_count(), "kind": "port",
            "single_core_value": n1 / sum(t1),
            "sample": "%d frames: extract %.3f s, %d pairs match %.3f s, %d pairs two-view (C++ f64, %d hyp) %.3f s on %d OpenMP threads; "
                      "1 thread on %d frames: %.3f s; own CPU restatement (oracle/cpu_baseline.cpp, %s), cv2 unavailable; "
                      "%d keypoints, %d matches, %d pose inliers"
                      % (n_frames, t[0], n_frames - 1, t[1], n_frames - 1, N_HYP, t[2], used, n1, sum(t1), flags, cnt[0], cnt[1], cnt[2])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="frames per rank per step (default 256 at every N = BASELINE config 3; at N > 1 a "
                                                        "second region with 512 per rank is reported too: 8 ranks = config 5's 4096 frames)")
    ap.add_argument("--scene", default="survey8d", choices=["survey8d", "smooth"], help="synthetic scene of the headline (vslam_amd/synth.py)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--prewarm-ms", type=float, default=150.0, help="untimed steps for this long ahead of the warm-up steps (clock ramp)")
    ap.add_argument("--sync-gather", type=int, default=0, help="N > 1: 1 = the gather of a step is waited for before the next step starts")
    ap.add_argument("--sync-steps", type=int, default=0, help="1: the host waits for every step before it enqueues the next one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optin", action="store_true", help="skip the extra timing of the opt-in matrix-core matcher")
    ap.add_argument("--cpu-frames", type=int, default=128, help="frames of the batch the CPU baseline processes (bounded sample)")
    ap.add_argument("--frames-cache", default="", help="file the generated frames of this rank are kept in (.npy; created when missing): "
                                                       "profiles/collect.sh generates once and profiles only the pipeline")
    ap.add_argument("--no-check", action="store_true", help="timing-only ablation builds (tools/fast_ablate.sh) produce garbage on purpose: skip the output asserts")
    ap.add_argument("--no-extras", action="store_true", help="skip the side legs (other scene, next rows, H2D-inclusive, single-frame)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import vslam_amd as V

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP front-end has no CPU fallback")
    local = local % torch.cuda.device_count()  # (rehearsal: several ranks may share one GPU under --backend gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    from vslam_amd.sharding import gather_map_points, shard
    K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])  # configs/monocular.yaml:3
    prm = V.orb_params(nfeatures=NFEAT, scale_factor=1.2, nlevels=8, edge_threshold=31, fast_threshold=7,
                       select_order=V.ORDER_LIBSTDCXX)
    # An explicit stream becomes this thread's current stream: everything torch enqueues from here on (copies, fills) and the
    # library's launches share it.
    torch.cuda.synchronize()
    main_s = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_s)
    # N > 1 under RCCL: the gather goes through the library's own entry point (mo_comm_init / mo_gather_map_points); if that
    # communicator cannot be set up on some rank, every rank falls back to torch.distributed's gather (agreed by an all-reduce)
    # BENCH_FORCE_GATHER=1 (rehearsal on a one-GPU box): the RCCL gather path - communicator, side stream, alternating buffers, the
    # checks on rank 0 - runs on a world of one (a self send / recv); everything except torch.distributed's part of it
    force_gather = world == 1 and os.environ.get("BENCH_FORCE_GATHER") == "1"
    rccl = {"on": (world > 1 and args.backend == "nccl") or force_gather}

    status = {}   # leg -> mo_dev_status word read after the leg (all 0, asserted)

    class Pipeline:
        """This rank's context, input frames and output buffers for nb frames (n_pairs = nb - 1 consecutive pairs)."""

        def __init__(self, frames, first_pair, rows):
            n = frames.shape[0]
            self.frames, self.n = frames, n
            self.ctx = V.Context(device=local, max_w=W, max_h=H, max_batch=n)
            self.ctx.set_stream(main_s.cuda_stream)
            z = lambda *s, dt=torch.int32: torch.zeros(s, dtype=dt, device=dev)
            self.kps = z(n, CAP, 7, dt=torch.float32)   # 28-byte mo_keypoint records
            self.desc = z(n, CAP, 32, dt=torch.uint8)
            self.counts = z(n)
            self.midx = z(n - 1, CAP, 2); self.mdist = z(n - 1, CAP, 2); self.mpass = z(n - 1, CAP, dt=torch.uint8)
            self.pose = z(n - 1, 12, dt=torch.float64)
            self.pts = [z(rows, CAP, 3, dt=torch.float32) for _ in range(2 if world > 1 or force_gather else 1)]  # N > 1: steps alternate (see Gather)
            self.npts = z(rows)
            io = V.BatchIO()
            io.d_gray = frames.data_ptr(); io.w = W; io.h = H; io.batch = n; io.cap = CAP
            io.ratio = 0.75; io.thr_px = 3.0; io.n_hyp = N_HYP; io.seed = 4096
            io.pair_index_base = first_pair  # global pair index of this rank's pair 0
            for i in range(9):
                io.K[i] = float(K.reshape(9)[i])
            io.d_kps = self.kps.data_ptr(); io.d_desc = self.desc.data_ptr(); io.d_counts = self.counts.data_ptr()
            io.d_match_idx = self.midx.data_ptr(); io.d_match_dist = self.mdist.data_ptr()
            io.d_match_pass = self.mpass.data_ptr(); io.d_pose = self.pose.data_ptr()
            io.d_points = self.pts[0].data_ptr(); io.d_n_points = self.npts.data_ptr()
            self.io = io

        def launch(self):
            self.ctx._check(self.ctx.lib.mo_dev_frontend_batch(self.ctx.h, C.byref(prm), C.byref(self.io)))

        def check(self, leg):
            """after a leg's synchronisation: no capacity flag was raised by any call of the leg (a clamped overflow would otherwise
            pass unnoticed: the kernels never fault, they clamp and raise a bit) and the outputs are populated"""
            st = self.ctx.dev_status()
            if args.no_check:
                status[leg] = st
                return
            assert st == 0, "leg %s: mo_dev_status = %d (capacity flag raised inside a timed region)" % (leg, st)
            status[leg] = st
            assert int(self.counts.min().item()) > 0, "leg %s: a frame without keypoints" % leg

        def stage_ms(self, n_calls):
            acc, n = {}, min(n_calls, V.TIMING_SLOTS)
            for back in range(n):  # the most recent calls, newest first
                for name, ms in self.ctx.stage_times(back):
                    acc[name] = acc.get(name, 0.0) + ms / n
            return acc

    class Gather:
        """The final map-point gather of a step, overlapped with the next step: steps alternate between two map-point buffers and a
        buffer re-enters the pipeline only after the gather that read it has finished (a stream-level wait).
        nccl: mo_gather_map_points (RCCL send / recv group behind the C-ABI) enqueued on a side stream; gloo (rehearsal on one GPU):
        torch.distributed.gather of a host copy."""

        def __init__(self, pl, B, n_pairs, pairs_all):
            self.pl, self.B, self.n_pairs, self.pairs_all = pl, B, n_pairs, pairs_all
            self.k = 0
            self.pending = [None, None]
            if rccl["on"]:
                self.comm_s = torch.cuda.Stream(device=dev)
                self.done = [torch.cuda.Event(), torch.cuda.Event()]
                self.all = torch.zeros((world, B, CAP, 3), dtype=torch.float32, device=dev) if rank == 0 else None
                self.rows_all = torch.zeros(world, dtype=torch.int32, device=dev)

        def before_step(self):
            k = self.k
            if self.pending[k] is not None:
                self.finish(k)
            self.pl.io.d_points = self.pl.pts[k].data_ptr()

        def finish(self, k):
            p = self.pending[k]
            self.pending[k] = None
            if p == "rccl":
                main_s.wait_event(self.done[k])
            elif p is not None:
                p[1]()

        def after_step(self):
            k = self.k
            self.k ^= 1
            buf = self.pl.pts[k]
            if rccl["on"]:
                self.comm_s.wait_stream(main_s)
                ctx = self.pl.ctx
                ctx.set_stream(self.comm_s.cuda_stream)
                ctx.gather_map_points(buf.data_ptr(), self.n_pairs, self.B, CAP, 0, self.all.data_ptr() if rank == 0 else 0,
                                      self.rows_all.data_ptr())
                ctx.set_stream(main_s.cuda_stream)
                self.done[k].record(self.comm_s)
                self.pending[k] = "rccl"
            else:
                self.pending[k] = gather_map_points(buf if args.backend == "nccl" else buf.cpu(), self.n_pairs, dst=0,
                                                    pairs_per_rank=self.pairs_all, async_op=True)
            if args.sync_gather:
                self.finish(k)

        def drain(self):
            for k in (0, 1):
                if self.pending[k] is not None:
                    self.finish(k)

    def timed_region(B, scene):
        """-> (pipeline, elapsed seconds (max over ranks), n_pairs): W warm-up + K timed steps of B frames per rank"""
        first, nb, n_pairs, first_pair = shard(rank, world, B)   # rank > 0 re-extracts the frame preceding its shard (halo)
        pairs_all = [shard(r, world, B)[2] for r in range(world)]
        cache = "%s.%s.%d.%d.npy" % (args.frames_cache, scene, first, nb) if args.frames_cache else ""
        if cache and os.path.exists(cache):
            frames = torch.from_numpy(np.load(cache)).to(dev)
        else:
            frames = make_frames(torch, dev, first, nb, scene=scene)
            if cache:
                np.save(cache, frames.cpu().numpy())
        pl = Pipeline(frames, first_pair, B)
        if rccl["on"]:
            ids = [None]
            if rank == 0:
                try:
                    ids[0] = V.Context.comm_unique_id()
                except Exception as e:
                    print("mo_comm_unique_id failed (%s): falling back to torch.distributed.gather" % e, file=sys.stderr)
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            ok = 0
            if ids[0] is not None:
                try:
                    pl.ctx.comm_init(ids[0], rank, world)
                    ok = 1
                except Exception as e:
                    print("rank %d: mo_comm_init failed (%s)" % (rank, e), file=sys.stderr)
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            rccl["on"] = bool(ok)
        ga = Gather(pl, B, n_pairs, pairs_all) if world > 1 or rccl["on"] else None

        def step():
            if ga:
                ga.before_step()
            pl.launch()
            if ga:
                ga.after_step()

        # Clock ramp: after the idle seconds of frame generation and context creation the first ~100 ms of work run 3 % below the
        # steady-state rate.  A fixed stretch of untimed steps ahead of the W warm-up steps puts the timed region on the steady-state
        # clock whatever K and W are.  (No collective in here: a time-based loop runs a different number of trips on every rank.)
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            pl.launch()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        if ga:
            ga.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # The K timed steps are enqueued back to back with no host synchronisation in between (the library call only enqueues; the
        # per-stage hipEvents of every step are read AFTER the closing synchronisation from the context's ring of event sets).
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            if args.sync_steps:
                pl.ctx.sync()
        if ga:
            ga.drain()  # (inside the timed region: every gather has completed before the clock stops)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            te = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            elapsed = float(te.item())
        pl.check("headline_%d" % B)
        if rccl["on"] and rank == 0:  # the gathered slabs are what the ranks computed: row counts and rank 0's own slab
            assert ga.rows_all.cpu().tolist() == pairs_all, (ga.rows_all.cpu().tolist(), pairs_all)
            k_last = ga.k ^ 1
            a, b = ga.all[0, :n_pairs], pl.pts[k_last][:n_pairs]
            assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
        return pl, elapsed, n_pairs

    B = args.batch if args.batch > 0 else 256
    second = None
    if world > 1 and args.batch <= 0:   # BASELINE config 5's shape first: 512 frames per rank (4096 at N = 8)
        pl5, el5, _ = timed_region(512, args.scene)
        second = {"frames_per_gpu": 512, "value": round(world * 512 * args.steps / el5, 2), "unit": "frames/s",
                  "ms_per_step": round(el5 / args.steps * 1e3, 3), "total_frames_per_step": world * 512}
        pl5.ctx.close()
        del pl5
        torch.cuda.empty_cache()
    pl, elapsed, n_pairs = timed_region(B, args.scene)   # the headline region (its per-stage figures are reported below)

    if rank == 0:
        nb = pl.n
        ms_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        cnt = pl.counts.cpu().numpy()
        npt = pl.npts[:n_pairs].cpu().numpy()
        mpass_mean = float(pl.mpass.sum(dim=1).float().mean().item())
        per_stage = pl.stage_ms(args.steps)
        units_of = lambda k: n_pairs if k in ("match_knn2_ratio", "two_view") else nb
        # pipeline bytes on SURVEY 8d's model (4 541 474 B per frame + 162 000 B per matched pair); the per-stage table below keeps its
        # own per-stage bytes (it charges the blur its reads too, which the blueprint's whole-pipeline model does not)
        total_alg = SURVEY_EXTRACT_BYTES * nb + SURVEY_MATCH_BYTES * n_pairs
        fused_alg = SURVEY_FUSED_IDEAL_BYTES * nb + SURVEY_MATCH_BYTES * n_pairs
        # roofline of the image kernel with the longest time: k_fast.  Every kernel runs in line on one stream (the blur too), so the
        # hipEvent span of a stage in the timed region IS its kernels' own duration.
        crit = "fast_nms"
        crit_ms = per_stage[crit]
        alg_bytes = STAGE_BYTES[crit] * nb
        achieved = alg_bytes / (crit_ms * 1e-3) / 1e9
        pmc = {}
        try:
            with open(os.path.join(ROOT, PMC_FILE)) as f:
                pmc = json.load(f)
        except Exception:
            pmc = {}
        pmc_ok = pmc.get("batch_frames") == B and pmc.get("scene") == args.scene
        kf = pmc.get("kernels", {}).get("k_fast", {}) if pmc_ok else {}
        traffic = kf.get("hbm_bytes")  # FETCH_SIZE x 2 (guide: 16-byte-per-lane streams count half) + WRITE_SIZE, per launch
        valu = kf.get("valu_insts")    # SQ_INSTS_VALU per launch (wave instructions)
        valu_frac = valu * 2.0 / (crit_ms * 1e-3 * 2.4e9 * 1024) if valu else None  # 2 cycles per wave64 VALU instruction, 1024 SIMDs, 2.4 GHz
        per_kernel = None
        if pmc_ok:
            # steps per profiled pass = launches of a once-per-step kernel (prewarm + warm-up + timed steps of profiles/collect_r03.sh)
            per_kernel, steps_prof = {}, float(pmc["kernels"].get("k_fast", {}).get("launches", 7.0))
            for st, names in STAGE_KERNELS.items():
                ks = [pmc["kernels"][n] for n in names if n in pmc.get("kernels", {})]
                if not ks or st not in per_stage:
                    continue
                per_step = lambda key: sum((k.get(key) or 0.0) * k.get("launches", steps_prof) / steps_prof for k in ks)
                ms_st, algb = per_stage[st], STAGE_BYTES.get(st, 0) * units_of(st)
                per_kernel[st] = {"kernels": names, "ms": round(ms_st, 4), "algorithmic_bytes": algb,
                                  "hbm_frac": round(algb / (ms_st * 1e-3) / 1e9 / PEAK_HBM_GBPS, 5),
                                  "traffic": round(per_step("hbm_bytes")),
                                  "valu_frac": round(per_step("valu_insts") * 2.0 / (ms_st * 1e-3 * 2.4e9 * 1024), 4)}
        out = {
            "metric": metric_name(),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_ms": args.prewarm_ms,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "batch of %d synthetic 640x480 frames per GPU: ORB extract (2000 feat, 8 levels, FAST-7) "
                                   "+ BF-Hamming 2-NN ratio 0.75 on consecutive pairs + 8-pt E RANSAC (%d hyp) pose/DLT per pair"
                                   % (B, N_HYP),
                       "scene": {"survey8d": "survey8d: SURVEY 8d texture (8-px cells 0..255, 400 rectangles / VGA, N(0,3) noise per frame), two "
                                             "depth layers, sub-pixel pan 8.37 / 16.74 px per frame (bilinear resample), roll <= 3 deg per pair",
                                 "smooth": "smooth: rounds 1-2 scene (32-px cells 90..170, small rectangles, N(0,1), whole-pixel pan, no roll)"}[args.scene],
                       "frames_per_gpu": B, "n_features": NFEAT, "hypotheses": N_HYP,
                       "keypoints_per_frame_mean": float(cnt.mean()), "matches_per_pair_mean": mpass_mean,
                       "map_points_per_pair_mean": float(npt.mean()),
                       "parallelism": "frame-sharded x%d, %s" % (world, "RCCL gather of map points through mo_gather_map_points" if rccl["on"]
                                                                 else "gather of map points (torch.distributed %s)" % args.backend if world > 1
                                                                 else "single GPU"),
                       "rccl_ranks": world if rccl["on"] else 0},
            "roofline": {"bound": "hbm", "kernel": "k_fast", "achieved": round(achieved, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(achieved / PEAK_HBM_GBPS, 5), "hbm_frac": round(achieved / PEAK_HBM_GBPS, 5),
                         "peak_measured": PEAK_HBM_MEASURED_GBPS, "frac_of_measured": round(achieved / PEAK_HBM_MEASURED_GBPS, 5),
                         "valu_frac": round(valu_frac, 4) if valu_frac else None, "traffic": traffic,
                         "traffic_source": (PMC_FILE + " (separate rocprofv3 --pmc passes of the same command, not this run)") if traffic else None,
                         "algorithmic_bytes": alg_bytes, "kernel_ms": round(crit_ms, 4),
                         "kernel_ms_source": "hipEvent span in the timed region (every kernel in line on one stream)",
                         "pipeline_achieved": round(total_alg / (ms_step * 1e-3) / 1e9, 2),
                         "pipeline_bytes_model": "SURVEY 8d stage-materialised: %d B/frame + %d B/pair" % (SURVEY_EXTRACT_BYTES, SURVEY_MATCH_BYTES),
                         "pipeline_frac": round(total_alg / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBPS, 5),
                         "pipeline_frac_of_measured": round(total_alg / (ms_step * 1e-3) / 1e9 / PEAK_HBM_MEASURED_GBPS, 5),
                         "fused_ideal": {"bytes_per_frame": SURVEY_FUSED_IDEAL_BYTES, "achieved": round(fused_alg / (ms_step * 1e-3) / 1e9, 2),
                                         "frac": round(fused_alg / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBPS, 5),
                                         "note": "SURVEY 8d fused-ideal lower bound (input 307 200 B + outputs 120 000 B per frame, + the "
                                                 "matcher's bytes): what a pipeline that never materialised a pyramid would have to move"},
                         "note": "k_fast = the image kernel with the longest time; algorithmic bytes (SURVEY 8d: 950 532 B per frame) / its "
                                 "own duration.  valu_frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz x duration); "
                                 "tools/ubench.hip measures 2.3 cycles only for add/sub/logic/shift-right/f32 add-mul and 4.5 for every other "
                                 "vector instruction at the occupancy these kernels run at, so an integer kernel tops out near 0.5 on this "
                                 "scale (DESIGN.md 4)"},
            "roofline_per_stage": per_kernel,
            "roofline_per_stage_source": PMC_FILE if per_kernel else None,
            "stage_ms": {k: round(v, 4) for k, v in per_stage.items()},
        }
        if second:
            out["config5_512_per_gpu"] = second
        single = world == 1 and not args.no_extras

        def timed_calls(leg, n_warm=None):
            """W warm-up + K timed launches of the headline pipeline in its current configuration -> (seconds, stage ms)"""
            t_pw = time.perf_counter()
            while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
                pl.launch()
                torch.cuda.synchronize()
            for _ in range(max(1, args.warmup) if n_warm is None else n_warm):
                pl.launch()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pl.launch()
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            pl.check(leg)
            return el, pl.stage_ms(args.steps)

        if single:
            # (s) the other scene on the headline context (input pointer switched)
            other = "smooth" if args.scene == "survey8d" else "survey8d"
            fr2 = make_frames(torch, dev, 0, nb, scene=other)
            pl.io.d_gray = fr2.data_ptr()
            el_s, st_s = timed_calls("scene_" + other)
            out["scene_" + other] = {"value": round(B * args.steps / el_s, 2), "unit": "frames/s", "ms_per_step": round(el_s / args.steps * 1e3, 3),
                                     "keypoints_per_frame_mean": float(pl.counts.float().mean().item()),
                                     "matches_per_pair_mean": float(pl.mpass.sum(dim=1).float().mean().item()),
                                     "stage_ms": {k: round(v, 4) for k, v in st_s.items()}}
            pl.io.d_gray = pl.frames.data_ptr()
            del fr2
        # Opt-in variant, timed outside the headline region on the same inputs: the matrix-core matcher
        # (VSLAM_AMD_MATCHER=mfma at context creation; identical results).  Its int8 operations (2 * 256 per descriptor
        # pair) are priced against the dense int8 MFMA peak (2 x the bf16 rate, MI355X_MICROARCH.md).
        if world == 1 and not args.no_optin:
            os.environ["VSLAM_AMD_MATCHER"] = "mfma"
            try:
                pl2 = Pipeline(pl.frames, 0, B)
            finally:
                os.environ.pop("VSLAM_AMD_MATCHER", None)
            for _ in range(max(1, args.warmup)):
                pl2.launch()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pl2.launch()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            pl2.check("matcher_mfma_optin")
            m_ms = pl2.stage_ms(args.steps).get("match_knn2_ratio", 0.0)
            c64 = pl2.counts.cpu().numpy().astype(np.float64)
            ops = float((c64[:-1] * c64[1:]).sum()) * 512.0
            tops = ops / (m_ms * 1e-3) / 1e12 if m_ms > 0 else 0.0
            out["matcher_mfma_optin"] = {"value": round(B * args.steps / el2, 2), "unit": "frames/s",
                                         "ms_per_step": round(el2 / args.steps * 1e3, 3), "match_ms": round(m_ms, 4),
                                         "roofline": {"bound": "mfma", "kernel": "k_match_mfma", "achieved": round(tops, 1),
                                                      "peak": 5000.0, "unit": "TOP/s", "frac": round(tops / 5000.0, 4),
                                                      "dtype": "int8"}}
            pl2.ctx.close()
            del pl2
        if single:
            # (c) the "next" rows of SURVEY 8f on the same frames, each as one batched call on the headline context:
            #     tracking step (MO_MODE_TRACK: ratio test, displacement filter at 2 % of (w + h) / 2, 2 x median distance filter,
            #     8-point E RANSAC at 1 px - tracker.py:214-254)
            sel = torch.zeros((n_pairs, CAP, 2), dtype=torch.int32, device=dev)
            seln = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
            pl.io.mode = V.MODE_TRACK; pl.io.disp_frac = 0.02; pl.io.thr_px = 1.0
            pl.io.d_sel_idx = sel.data_ptr(); pl.io.d_sel_n = seln.data_ptr()
            elt, acct = timed_calls("track_mode")
            out["next_rows"] = {"track_mode": {"value": round(B * args.steps / elt, 2), "unit": "frames/s", "ms_per_step": round(elt / args.steps * 1e3, 3),
                                               "kept_matches_per_pair_mean": float(seln.float().mean().item()),
                                               "stage_ms": {k: round(v, 4) for k, v in acct.items()}}}
            pl.io.mode = V.MODE_INIT; pl.io.thr_px = 3.0; pl.io.d_sel_idx = None; pl.io.d_sel_n = None
            out["next_rows"].update(next_row_legs(torch, V, pl, prm, dev, args, timed_calls, n_pairs, K))
        if single:
            # (a) PCIe-inclusive rate: every step first copies its frames from pinned host memory into HBM on the same stream
            #     (307 200 B per frame over Gen5 x16); never the headline value
            host = torch.empty((nb, H, W), dtype=torch.uint8).pin_memory()
            host.copy_(pl.frames.cpu())
            bufs = [torch.empty_like(pl.frames), torch.empty_like(pl.frames)]

            def h2d_step():
                bufs[0].copy_(host, non_blocking=True)  # (current stream = the context's stream)
                pl.launch()
            pl.io.d_gray = bufs[0].data_ptr()
            for _ in range(max(1, args.warmup)):
                h2d_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                h2d_step()
            torch.cuda.synchronize()
            el3 = time.perf_counter() - t1
            pl.check("h2d_inclusive")
            # (a') the same with the copy of batch i + 1 overlapped with the compute of batch i: two device buffers, a copy stream,
            #      events both ways (compute waits for its buffer's copy, the copy waits until the buffer's last reader is done)
            copy_s = torch.cuda.Stream(device=dev)
            copied = [torch.cuda.Event(), torch.cuda.Event()]
            freed = [torch.cuda.Event(), torch.cuda.Event()]

            def enqueue_copy(k):
                with torch.cuda.stream(copy_s):
                    copy_s.wait_event(freed[k])
                    bufs[k].copy_(host, non_blocking=True)
                    copied[k].record(copy_s)
            for k in (0, 1):
                freed[k].record(main_s)
            torch.cuda.synchronize()
            n_ov = args.steps + 2
            t1 = time.perf_counter()
            enqueue_copy(0)
            for i in range(n_ov):
                k = i & 1
                if i + 1 < n_ov:
                    enqueue_copy(1 - k)
                main_s.wait_event(copied[k])
                pl.io.d_gray = bufs[k].data_ptr()
                pl.launch()
                freed[k].record(main_s)
            torch.cuda.synchronize()
            el4 = time.perf_counter() - t1
            pl.check("h2d_overlapped")
            pl.io.d_gray = pl.frames.data_ptr()
            out["h2d_inclusive"] = {"value": round(B * args.steps / el3, 2), "unit": "frames/s", "ms_per_step": round(el3 / args.steps * 1e3, 3),
                                    "overlapped_value": round(B * n_ov / el4, 2), "overlapped_ms_per_step": round(el4 / n_ov * 1e3, 3),
                                    "note": "pinned host -> HBM copy of the batch (%.1f MB) inside every step; 'overlapped': double-buffered, "
                                            "the copy of the next batch runs on its own stream beside the compute of the current one"
                                            % (nb * H * W / 1e6)}
            del bufs, host
            # (b) single-frame latency through the drop-in classes, host arrays in and Python objects out: what the reference's
            #     Tracker would see per call (BASELINE config 2; extract_features(distributed=True) is Tracker's default path)
            from orbslam2.extractor import ORBExtractor
            from orbslam2.matcher import DescriptorMatcher
            from orbslam2 import utils as geom
            f0, f1 = pl.frames[0].cpu().numpy(), pl.frames[1].cpu().numpy()
            ex = ORBExtractor(n_features=NFEAT)
            mt = DescriptorMatcher("bruteforce-hamming", ratio_threshold=0.75)

            def med_ms(fn, n):
                fn()
                ts = []
                for _ in range(n):
                    t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
                return round(sorted(ts)[len(ts) // 2], 3)
            (k0, d0), (k1, d1) = ex.detect_and_compute(f0), ex.detect_and_compute(f1)
            ctx1 = V.default_context()

            def breakdown(fn, n=20):
                """batch-1 latency split of one host call: wall and the C call's own clocks (entry -> enqueued -> stream waited ->
                unpacked) with the stage events off (the default), then the device spans of the same call with them on"""
                ctx1.set_host_timing(False)
                fn(); ts, hs = [], []
                for _ in range(n):
                    t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3); hs.append(ctx1.host_times())
                wall = float(np.median(ts))
                hm = {k: float(np.median([h[k] for h in hs])) / 1e3 for k in hs[0]}
                ctx1.set_host_timing(True)
                fn(); ev = []
                for _ in range(max(5, n // 2)):
                    fn(); ev.append(dict(ctx1.stage_times()))
                ctx1.set_host_timing(False)
                dev = {k: round(float(np.median([e[k] for e in ev])), 4) for k in ev[0]}
                kern = sum(v for k, v in dev.items() if k not in ("h2d", "d2h"))
                return {"wall_ms": round(wall, 4), "python_binding_ms": round(wall - hm["total_us"], 4), "c_enqueue_ms": round(hm["enqueue_us"], 4),
                        "c_wait_ms": round(hm["wait_us"], 4), "c_unpack_ms": round(hm["unpack_us"], 4),
                        "device_h2d_ms": dev.get("h2d"), "device_d2h_ms": dev.get("d2h"), "device_kernels_ms": round(kern, 4),
                        "device_stage_ms": dev,
                        "launch_sync_overhead_ms": round(hm["enqueue_us"] + hm["wait_us"] - sum(dev.values()), 4)}
            # a Tracker in TRACKING state (tracker.py:87,198-266): every frame extracted once and tracked against the previous one, whose
            # keypoints / descriptors are still resident on the device (the very arrays detect_and_compute handed out come back)
            trk = {"last": ex.detect_and_compute(f0), "i": 0}

            def tracker_frame():
                trk["i"] ^= 1
                cur = ex.detect_and_compute(f1 if trk["i"] else f0)
                r = geom.track_from_last_frame(trk["last"][0], trk["last"][1], cur[0], cur[1], K, f1.shape)
                trk["last"] = cur
                return r
            # the same on the detector Tracker.process_frame takes by default (tracker.py:87: extract_features() with distributed=True), in
            # the aligned form a tracker needs (keypoint i belongs to descriptor row i; the reference's own list holds ALL corners)
            trg = {"last": ex.distribute_keypoints(f0, aligned=True), "i": 0}

            def tracker_frame_grid():
                trg["i"] ^= 1
                cur = ex.distribute_keypoints(f1 if trg["i"] else f0, aligned=True)
                r = geom.track_from_last_frame(trg["last"][0], trg["last"][1], cur[0], cur[1], K, f1.shape)
                trg["last"] = cur
                return r
            from orbslam2.initializer import MapInitializer
            import contextlib, io as _io

            def initialize_pair():
                ini = MapInitializer(K)
                ini.set_first_frame(k0, d0, f0)
                with contextlib.redirect_stdout(_io.StringIO()):
                    return ini.initialize(k1, d1, mt, f1)
            out["single_frame_ms"] = {
                "detect_and_compute": med_ms(lambda: ex.detect_and_compute(f0), 20),
                "detect_and_compute_all_objects": med_ms(lambda: tuple(ex.detect_and_compute(f0)[0]), 20),
                "detect_and_compute_native_arrays": med_ms(lambda: ctx1.orb_detect_compute(f0, ex.orb.prm), 20),
                "extract_features_distributed": med_ms(lambda: ex.extract_features(f0, distributed=True), 10),
                "extract_features_distributed_all_objects": med_ms(lambda: list(ex.extract_features(f0, distributed=True)[0]), 10),
                "tracker_frame": med_ms(tracker_frame, 20),
                "tracker_frame_grid_detector": med_ms(tracker_frame_grid, 20)}
            (k0, d0), (k1, d1) = ex.detect_and_compute(f0), ex.detect_and_compute(f1)   # (the last two extractions: both resident)
            out["single_frame_ms"].update({
                "match_2000x2000": med_ms(lambda: mt.match(d0, d1), 20),
                "match_native_arrays": med_ms(lambda: ctx1.match_knn2_ratio(d0, d1, 0.75), 20),
                "track_from_last_frame": med_ms(lambda: geom.track_from_last_frame(k0, d0, k1, d1, K, f1.shape), 20),
                "initialize": med_ms(initialize_pair, 10)})
            out["single_frame_ms"]["breakdown"] = {
                "detect_and_compute": breakdown(lambda: ctx1.orb_detect_compute(f0, ex.orb.prm))}
            (k0, d0), (k1, d1) = ex.detect_and_compute(f0), ex.detect_and_compute(f1)
            out["single_frame_ms"]["breakdown"]["track_pair_resident"] = breakdown(lambda: ctx1.track_pair(k0.array, d0, k1.array, d1, W, H, K))
            out["single_frame_ms"]["breakdown"]["match_resident"] = breakdown(lambda: ctx1.match_knn2_ratio(d0, d1, 0.75))
            # (the call behind extract_features(distributed=True), which Tracker.process_frame makes for every frame: tracker.py:87)
            out["single_frame_ms"]["breakdown"]["grid_detect_compute"] = breakdown(lambda: ctx1.grid_detect_compute(f0, ex.orb.prm, NFEAT, records=True))
            out["single_frame_ms"]["note"] = (
                "BASELINE config 2, median wall ms per call through the drop-in classes, host numpy in / Python objects out: the image goes "
                "through pinned staging and an upload kernel, results stay resident on the device (tokens) and come back through one pack "
                "kernel, one synchronisation per call; tracker_frame = what a Tracker pays per frame in TRACKING state: detect_and_compute + "
                "track_from_last_frame on resident frames (tracker.py:87,198-266) with the ORB detector, tracker_frame_grid_detector = the same with "
                "distribute_keypoints(aligned=True), the detector Tracker.process_frame takes by default; extract_features_distributed returns the "
                "reference's list of ALL corners as a lazy KeyPointList (*_all_objects forces the objects); initialize = MapInitializer.initialize as ONE device "
                "call; detect_and_compute returns a lazy KeyPoint sequence (*_all_objects forces all 2000 objects); breakdown: wall and the "
                "C call's clocks with the stage events off, device spans (hipEvents) of the same call with them on")
            # (d) the same frames as a SEQUENCE through vslam_amd.stream.FrameStream (mo_stream): host frames in, per-frame results out,
            #     chunks of 64 through the batched mode with the upload double-buffered - what a driver that can look ahead gets
            from vslam_amd.stream import FrameStream
            host_frames = pl.frames.cpu().numpy()
            host_frames = np.concatenate([host_frames, host_frames[::-1]] * 4)   # a 2048-frame stack (the pan there and back, four times)
            legs = {}
            for name, kw in (("track_orb_chunk64", dict(copy=False)), ("track_orb_chunk128", dict(copy=False, chunk=128)),
                             ("track_orb_chunk64_copies", dict()), ("track_orb_chunk64_iterator", dict(copy=False)),
                             ("track_grid_chunk64", dict(copy=False, detector=V.DETECT_GRID))):
                kw.setdefault("chunk", 64)
                fs = FrameStream(K, width=W, height=H, n_features=NFEAT, cap=CAP, n_hyp=N_HYP, **kw)
                try:
                    src = (lambda a: iter(a)) if name.endswith("_iterator") else (lambda a: a)
                    # warm-up, untimed: plan and lanes - and, for the FIRST stream of the process, the three ~ 6 ms host stalls of the runtime's
                    # download call at its chunks 2, 6 and 11 (profiles/r04_stream_where.txt, r04_ab_stream_d2h.txt), which a 5-chunk warm-up left
                    # inside this leg's 32 timed chunks
                    n_warm = (16 if not legs else 4) * kw["chunk"] + 2
                    n_seen = sum(1 for _ in fs.run(src(host_frames[:n_warm])))
                    t1 = time.perf_counter()
                    n_seen, n_in = 0, 0
                    for r in fs.run(src(host_frames)):
                        n_seen += 1
                        p = r.pair                          # (a tracker's use of a frame: its pose and the kept matches' inlier flags;
                        if p is not None and p.ok:          #  the feature arrays are there as views, touched for every 16th frame)
                            n_in += p.n_inliers + int(p.inlier[0]) + (p.R[0, 0] > 2.0)
                        if (n_seen & 15) == 0:
                            n_in += len(r.keypoints) + len(r.descriptors)
                    el = time.perf_counter() - t1
                finally:
                    fs.close()
                legs[name] = {"value": round(n_seen / el, 1), "unit": "frames/s", "frames": n_seen, "ms_per_chunk": round(el / n_seen * kw["chunk"] * 1e3, 3)}
            legs["note"] = ("FrameStream (mo_stream) over a 2048-frame pageable host frame stack (N, H, W) built from the bench's frames, H2D-inclusive: "
                            "staging into pinned memory (pool of host threads), upload stream, one mo_dev_frontend_batch per chunk in tracking mode "
                            "(each leg after an untimed warm-up pass on the same stream: 16 chunks for the first stream of the process, 4 for the others) "
                            "(ratio test, 2 filters, 8-pt RANSAC %d hyp at 1 px), download stream, three chunks in flight, one Python FrameResult per "
                            "frame whose pose and inlier flags the loop reads (feature arrays touched on every 16th frame); result arrays are views of "
                            "the pinned buffers except in *_copies (the caller's own copies, made per chunk); *_iterator: frames arrive one by one from a "
                            "Python iterator and are gathered into chunks; track_grid: the detector Tracker uses by default (distribute_keypoints)" % N_HYP)
            out["streaming"] = legs
        out["dev_status"] = dict(status, note="mo_dev_status after every leg's synchronisation: 0 = no capacity flag raised (asserted)")
        if not args.no_cpu_baseline and world == 1:
            nf = min(args.cpu_frames, nb)
            out["cpu_baseline"] = cpu_baseline(pl.frames[:nf].cpu().numpy(), nf, K)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    pl.ctx.close()


def next_row_legs(torch, V, pl, prm, dev, args, timed_calls, n_pairs, K):
    """SURVEY 8f rows beyond the tracking step, on the headline frames: one keyframe pair through the fundamental-matrix RANSAC of
    local_mapper.py:116-149 (host API).  (The batched grid detector and keyframe mode add their legs here as they are built.)"""
    legs = {}
    kp0 = pl.kps[0, :int(pl.counts[0].item())].cpu().numpy()
    kp1 = pl.kps[1, :int(pl.counts[1].item())].cpu().numpy()
    keep = pl.mpass[0, :len(kp0)].cpu().numpy().astype(bool)
    tr = pl.midx[0, :len(kp0), 0].cpu().numpy()
    p1 = kp0[keep][:, :2].copy(); p2 = kp1[tr[keep]][:, :2].copy()
    ctxf = V.default_context()
    ctxf.find_fundamental(p1, p2, 3.0)
    tsf = []
    for _ in range(10):
        t = time.perf_counter(); ctxf.find_fundamental(p1, p2, 3.0); tsf.append((time.perf_counter() - t) * 1e3)
    legs["find_fundamental_ms"] = {"value": round(sorted(tsf)[len(tsf) // 2], 3), "correspondences": int(len(p1)),
                                   "note": "one pair, host API (H2D + 4096-hypothesis F RANSAC + D2H + sync)"}
    # the detector Tracker.process_frame really uses (tracker.py:87 -> extract_features(distributed=True) -> extractor.py:85-144):
    # 8x8 grid of Shi-Tomasi corners + orb.compute at angle -1, as one batched call on the same frames (MO_DETECT_GRID), followed by
    # the same match + two-view stages
    pl.io.detector = V.DETECT_GRID
    elg, stg = timed_calls("grid_mode")
    legs["grid_mode"] = {"value": round(pl.n * args.steps / elg, 2), "unit": "frames/s", "ms_per_step": round(elg / args.steps * 1e3, 3),
                         "keypoints_per_frame_mean": float(pl.counts.float().mean().item()),
                         "matches_per_pair_mean": float(pl.mpass.sum(dim=1).float().mean().item()),
                         "stage_ms": {k: round(v, 4) for k, v in stg.items()}}
    pl.io.detector = V.DETECT_ORB
    # keyframe map growth (LocalMapper._process_new_keyframe, local_mapper.py:116-149; Tracker inserts a keyframe every 20th frame,
    # tracker.py:290) as one batched call: every 20th frame of the batch paired with the next one - ratio-0.8 match, F-RANSAC (4096
    # hypotheses, 3 px), triangulation with the two keyframe poses
    kf = list(range(0, pl.n, 20))
    q = torch.tensor(kf[:-1], dtype=torch.int32, device=dev); t = torch.tensor(kf[1:], dtype=torch.int32, device=dev)
    Pm = np.zeros((len(kf), 3, 4)); Pm[:, :, :3] = K
    for j, f in enumerate(kf):
        Pm[j, :, 3] = K @ np.array([-0.05 * f, 0.0, 0.0])
    dP1 = torch.from_numpy(np.ascontiguousarray(Pm[:-1].reshape(-1, 12))).to(dev); dP2 = torch.from_numpy(np.ascontiguousarray(Pm[1:].reshape(-1, 12))).to(dev)
    pl.io.mode = V.MODE_KEYFRAME; pl.io.ratio = 0.8; pl.io.n_kf_pairs = len(kf) - 1
    pl.io.d_kf_query = q.data_ptr(); pl.io.d_kf_train = t.data_ptr(); pl.io.d_kf_P1 = dP1.data_ptr(); pl.io.d_kf_P2 = dP2.data_ptr()
    elk, stk = timed_calls("keyframe_mode")
    legs["keyframe_mode"] = {"value": round(pl.n * args.steps / elk, 2), "unit": "frames/s", "ms_per_step": round(elk / args.steps * 1e3, 3),
                             "keyframe_pairs": len(kf) - 1, "inliers_per_pair_mean": float(pl.npts[:len(kf) - 1].float().mean().item()),
                             "stage_ms": {k: round(v, 4) for k, v in stk.items()},
                             "note": "extraction of all %d frames + %d keyframe pairs (every 20th frame) matched, F-RANSAC, triangulated" % (pl.n, len(kf) - 1)}
    pl.io.mode = V.MODE_INIT; pl.io.ratio = 0.75; pl.io.n_kf_pairs = 0
    pl.launch()  # (the legs below read the headline configuration's outputs again)
    torch.cuda.synchronize()
    pl.check("restore_headline")
    return legs


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- frames/sec (extract + match + pose) at 640x480, 2000 ORB, on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic frames already resident in HBM:
extract every frame (8-level pyramid, FAST-9, retainBest x2, Harris, IC angle, blur, rBRIEF), match every
consecutive frame pair (brute-force Hamming 2-NN + ratio test), two-view pose + map points per pair
(8-point E RANSAC over 4096 hypotheses, pose recovery, DLT), then the final map-point gather to rank 0.

Workload at N=1: BASELINE.json configs[2] "batch of 256 synthetic 640x480 frames, extract+match pipeline"
plus the two-view stage of configs[3] on every pair.  N>1: frames are independent, so the global frame
sequence is sharded contiguously (weak scaling: 512 frames per rank by default, so that 8 ranks process BASELINE
config 5's 4096 frames); each rank re-extracts the one frame preceding its shard (halo) instead of receiving it, and
the only collective is the RCCL gather of map points (issued asynchronously: it runs beside the next step's kernels and
is drained inside the timed region).

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
PEAK_HBM_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured copy rate

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


def _scene(seed, w, h, rects_per_vga):
    """Seeded canvas: smooth low-contrast shading + many small uniform-grey rectangles (strong, repeatable corners)
    + N(0,1) texture noise baked in."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cw, ch = w // 32 + 2, h // 32 + 2
    cells = rng.uniform(90, 170, size=(ch, cw))
    ys = (np.arange(h) + 0.5) / 32.0
    xs = (np.arange(w) + 0.5) / 32.0
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    img = (cells[y0][:, x0] * (1 - fy) * (1 - fx) + cells[y0][:, x0 + 1] * (1 - fy) * fx +
           cells[y0 + 1][:, x0] * fy * (1 - fx) + cells[y0 + 1][:, x0 + 1] * fy * fx)
    for _ in range(rects_per_vga * w // W):
        rw, rh = rng.integers(5, 22, size=2)
        x = rng.integers(0, w - 1); y = rng.integers(0, h - 1)
        img[y:y + rh, x:x + rw] = rng.uniform(0, 255)
    img = img + rng.normal(0, 1.0, size=img.shape)
    return np.clip(img, 0, 255).astype(np.float32)


def make_frames(torch, device, first, count, seed=20250523):
    """Frames [first, first+count) of the global synthetic sequence: a camera translating along x past a two-depth
    scene (background pans 8 px / frame, foreground patches 16 px / frame, i.e. depths 40 and 20 baselines at
    f = 320) plus per-frame N(0,1) sensor noise.  Non-planar, ~850 ratio-test matches per consecutive pair."""
    span = 2048
    wide = span + 2 * W
    bg = torch.from_numpy(_scene(seed, wide, H, 800)).to(device)
    fg = torch.from_numpy(_scene(seed + 1, wide, H, 800)).to(device)
    mk = torch.from_numpy(_scene(seed + 2, wide, H, 40)).to(device)
    out = torch.empty((count, H, W), dtype=torch.uint8, device=device)
    for i in range(count):
        g = first + i
        xb, xf = (8 * g) % span, (16 * g) % span
        gen = torch.Generator(device=device)
        gen.manual_seed(seed * 1000003 + g)
        fr = torch.where(mk[:, xf:xf + W] > 130.0, fg[:, xf:xf + W], bg[:, xb:xb + W])
        fr = fr + 1.0 * torch.randn((H, W), generator=gen, device=device)
        out[i] = fr.round().clamp_(0, 255).to(torch.uint8)
    return out


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
    core over frames / pairs, on the SAME batch the GPU step processes; a 1-thread run on a 16-frame sample is reported beside it."""
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
    n1 = min(16, n_frames)
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
    ap.add_argument("--batch", type=int, default=0, help="frames per rank per step (default: 256 at N = 1 = BASELINE config 3; 512 at "
                                                        "N > 1, so that 8 ranks process BASELINE config 5's 4096 frames)")
    ap.add_argument("--streams", type=int, default=1, help="concurrent sub-batches (HIP streams) per GPU")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--prewarm-ms", type=float, default=150.0, help="untimed steps for this long ahead of the warm-up steps (clock ramp)")
    ap.add_argument("--sync-gather", type=int, default=0, help="N > 1: 1 = blocking gather after every step instead of the overlapped one")
    ap.add_argument("--sync-steps", type=int, default=0, help="1: the host waits for every step before it enqueues the next one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optin", action="store_true", help="skip the extra timing of the opt-in matrix-core matcher")
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames of the batch the CPU baseline processes (all host cores)")
    ap.add_argument("--no-extras", action="store_true", help="skip the stand-alone kernel timing, H2D-inclusive and single-frame legs")
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
    B = args.batch if args.batch > 0 else (256 if world == 1 else 512)
    first, nb, n_pairs, _ = shard(rank, world, B)   # rank > 0 re-extracts the frame preceding its shard (halo)
    pairs_all = [shard(r, world, B)[2] for r in range(world)]
    frames = make_frames(torch, dev, first, nb)
    K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])  # configs/monocular.yaml:3

    prm = V.orb_params(nfeatures=NFEAT, scale_factor=1.2, nlevels=8, edge_threshold=31, fast_threshold=7,
                       select_order=V.ORDER_LIBSTDCXX)
    pts = torch.zeros((B, CAP, 3), dtype=torch.float32, device=dev)  # rank 0 fills B-1 pairs, the others B
    pts_alt = torch.zeros_like(pts) if world > 1 else pts            # N > 1: steps alternate between two buffers (see step())
    npts = torch.zeros(B, dtype=torch.int32, device=dev)

    class SubBatch:
        """A contiguous run of this rank's pairs [p0, p1) = frames [p0, p1] on its own HIP stream and context.
        Adjacent sub-batches share one frame (re-extracted, like the inter-rank halo); concurrent streams let the
        latency-bound kernels of one sub-batch (selection replay, two-view refit) overlap the VALU-bound ones of
        the other."""

        def __init__(self, p0, p1, src=None, stream=None):
            self.p0, self.p1, n = p0, p1, p1 - p0 + 1
            self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
            self.ctx = V.Context(device=local, max_w=W, max_h=H, max_batch=n)
            self.ctx.set_stream(self.stream.cuda_stream)
            self.kps = torch.zeros((n, CAP, 7), dtype=torch.float32, device=dev)   # 28-byte mo_keypoint records
            self.desc = torch.zeros((n, CAP, 32), dtype=torch.uint8, device=dev)
            self.counts = torch.zeros(n, dtype=torch.int32, device=dev)
            self.midx = torch.zeros((n - 1, CAP, 2), dtype=torch.int32, device=dev)
            self.mdist = torch.zeros((n - 1, CAP, 2), dtype=torch.int32, device=dev)
            self.mpass = torch.zeros((n - 1, CAP), dtype=torch.uint8, device=dev)
            self.pose = torch.zeros((n - 1, 12), dtype=torch.float64, device=dev)
            io = V.BatchIO()
            io.d_gray = (frames if src is None else src)[p0:].data_ptr(); io.w = W; io.h = H; io.batch = n; io.cap = CAP
            io.ratio = 0.75; io.thr_px = 3.0; io.n_hyp = N_HYP; io.seed = 4096
            io.pair_index_base = first + p0  # global pair index of this sub-batch's pair 0
            for i in range(9):
                io.K[i] = float(K.reshape(9)[i])
            io.d_kps = self.kps.data_ptr(); io.d_desc = self.desc.data_ptr(); io.d_counts = self.counts.data_ptr()
            io.d_match_idx = self.midx.data_ptr(); io.d_match_dist = self.mdist.data_ptr()
            io.d_match_pass = self.mpass.data_ptr(); io.d_pose = self.pose.data_ptr()
            io.d_points = pts[p0:].data_ptr(); io.d_n_points = npts[p0:].data_ptr()
            self.io = io

        def launch(self):
            self.ctx._check(self.ctx.lib.mo_dev_frontend_batch(self.ctx.h, C.byref(prm), C.byref(self.io)))

    S = max(1, min(args.streams, n_pairs))
    cuts = [round(j * n_pairs / S) for j in range(S + 1)]
    # An explicit stream becomes this thread's current stream: everything torch enqueues from here on (copies, the gather's
    # stream hand-over) and the library's launches share it.  (torch's DEFAULT stream has the handle 0, which mo_set_stream takes
    # as "use the context's own stream": the library would then run unordered beside torch's work.)
    torch.cuda.synchronize()
    main = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main)
    assert main.cuda_stream != 0
    # one sub-batch: it runs on the current stream itself - a side stream costs two cross-stream event hand-overs per step
    # (wait_stream both ways, ~25 us of idle GPU each), 2 % of a 2.3 ms step
    subs = [SubBatch(cuts[j], cuts[j + 1], stream=main if S == 1 else None) for j in range(S)]
    stage_acc = {}

    def step_local():
        for sb in subs:
            if sb.stream is not main:
                sb.stream.wait_stream(main)
            sb.launch()
        for sb in subs:
            if sb.stream is not main:
                main.wait_stream(sb.stream)

    # N > 1: the gather of step i runs on RCCL's own stream beside the kernels of step i + 1 (rank 0 receives 7 x 12.6 MB per step at
    # N = 8, ~0.3 ms of a 4.2 ms step if the next step waited for it).  Steps alternate between two map-point buffers; a buffer is
    # handed to the pipeline again only after the gather that read it has been waited for (a stream-level wait under RCCL).
    # --sync-gather 1: the plain blocking gather after every step.
    pending = [None, None]
    step_no = [0]

    def step():
        k = step_no[0] & 1
        step_no[0] += 1
        if world == 1 or args.sync_gather:
            step_local()
            if world > 1:  # final map-point gather (the only collective on the path)
                gather_map_points(pts if args.backend == "nccl" else pts.cpu(), n_pairs, dst=0, pairs_per_rank=pairs_all)
            return
        if pending[k] is not None:
            pending[k][1]()          # the gather that read this buffer two steps ago
            pending[k] = None
        buf = pts if k == 0 else pts_alt
        for sb in subs:
            sb.io.d_points = buf[sb.p0:].data_ptr()
        step_local()
        pending[k] = gather_map_points(buf if args.backend == "nccl" else buf.cpu(), n_pairs, dst=0, pairs_per_rank=pairs_all,
                                       async_op=True)

    def drain():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k][1]()
                pending[k] = None

    # Clock ramp: after the idle seconds of frame generation and context creation the first ~100 ms of work run 3 % below the
    # steady-state rate (10 timed steps after 2 warm-up steps: 2.30 ms per step, 40 steps: 2.24 ms).  A fixed stretch of untimed
    # steps ahead of the W warm-up steps puts the timed region on the steady-state clock whatever K and W are.
    t_pw = time.perf_counter()
    while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
        step_local()  # (no collective in here: a time-based loop runs a different number of trips on every rank)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # The K timed steps are enqueued back to back with no host synchronisation in between (the library call only enqueues; the
    # per-stage hipEvents of every step are read AFTER the closing synchronisation from the context's ring of event sets).
    # Reading them inside the loop - as rounds 1 and 2 did - waits for the step's last event and leaves the GPU idle while
    # the host enqueues the next step's ~30 launches: 0.06 ms of every 2.26 ms step.
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if args.sync_steps:
            for sb in subs:
                sb.ctx.sync()
    drain()  # (inside the timed region: every gather has completed before the clock stops)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_hist = 0
    for sb in subs:
        for back in range(min(args.steps, V.TIMING_SLOTS)):  # the most recent steps of the timed region, newest first
            for name, ms in sb.ctx.stage_times(back):
                stage_acc[name] = stage_acc.get(name, 0.0) + ms
        n_hist = min(args.steps, V.TIMING_SLOTS)
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    if rank == 0:
        total_frames = world * B * args.steps
        ms_step = elapsed / args.steps * 1e3
        value = total_frames / elapsed
        cnt = torch.cat([sb.counts for sb in subs]).cpu().numpy()
        npt = npts[:n_pairs].cpu().numpy()
        mpass_mean = float(torch.cat([sb.mpass for sb in subs]).sum(dim=1).float().mean().item())
        per_stage = {k: v / max(n_hist, 1) for k, v in stage_acc.items()}
        n_ext = sum(sb.io.batch for sb in subs)  # frames extracted per step (sub-batches share one frame each)
        units_of = lambda k: n_pairs if k in ("match_knn2_ratio", "two_view") else n_ext
        total_alg = sum(STAGE_BYTES.get(k, 0) * units_of(k) for k in per_stage)
        # Stand-alone kernel times: in the timed region the blur runs on a low-priority stream BESIDE fast_nms + select_harris,
        # so those three spans are stretched by one another.  A few extra steps on a second context with the blur serialised
        # (VSLAM_AMD_SERIAL_BLUR=1; "fast_nms" then spans blur + FAST on one stream) give each kernel's own duration.
        alone = None
        if world == 1 and S == 1 and not args.no_extras:
            os.environ["VSLAM_AMD_SERIAL_BLUR"] = "1"
            try:
                sbs = SubBatch(0, n_pairs)
            finally:
                os.environ.pop("VSLAM_AMD_SERIAL_BLUR", None)
            acc = {}
            for it in range(3 + 10):
                sbs.launch()
                torch.cuda.synchronize()
                if it >= 3:
                    for name, ms in sbs.ctx.stage_times():
                        acc[name] = acc.get(name, 0.0) + ms / 10
            alone = dict(acc)
            alone["fast_nms"] = acc["fast_nms"] - acc["blur"]
        # roofline of the kernel on the critical path with the largest stand-alone time among the image kernels: k_fast
        crit = "fast_nms"
        crit_ms = alone[crit] if alone else per_stage[crit]
        alg_bytes = STAGE_BYTES[crit] * n_ext
        achieved = alg_bytes / (crit_ms * 1e-3) / 1e9
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r02_pmc_per_kernel.json")) as f:
                pmc = json.load(f)
        except Exception:
            pmc = {}
        kf = pmc.get("kernels", {}).get("k_fast", {}) if pmc.get("batch_frames") == B else {}
        traffic = kf.get("hbm_bytes")  # FETCH_SIZE x 2 (guide: 16-byte-per-lane streams count half) + WRITE_SIZE, per launch
        valu = kf.get("valu_insts")    # SQ_INSTS_VALU per launch (wave instructions)
        valu_frac = valu * 2.0 / (crit_ms * 1e-3 * 2.4e9 * 1024) if valu else None  # 2 cycles per wave64 VALU instruction, 1024 SIMDs, 2.4 GHz
        # the same accounting for every stage (stand-alone duration; counters summed over the stage's kernels and launches per step)
        STAGE_KERNELS = {"pyramid": ["k_resize2", "k_resize"], "fast_nms": ["k_fast"], "select_harris": ["k_select"], "blur": ["k_blur"],
                         "angle_rbrief": ["k_describe"], "match_knn2_ratio": ["k_pair_frames", "k_match_lds"],
                         "two_view": ["k_tv_prep", "k_tv_hyp", "k_tv_tasks", "k_tv_score", "k_tv_finish"]}
        per_kernel = None
        if alone and pmc.get("batch_frames") == B:
            # steps per profiled pass = launches of a once-per-step kernel (prewarm + warm-up + timed steps of profiles/collect_r02.sh)
            per_kernel, steps_prof = {}, float(pmc["kernels"].get("k_fast", {}).get("launches", 7.0))
            for st, names in STAGE_KERNELS.items():
                ks = [pmc["kernels"][n] for n in names if n in pmc.get("kernels", {})]
                if not ks or st not in alone:
                    continue
                per_step = lambda key: sum(k.get(key, 0.0) * k.get("launches", steps_prof) / steps_prof for k in ks)
                ms_st, algb = alone[st], STAGE_BYTES.get(st, 0) * units_of(st)
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
                       "frames_per_gpu": B, "streams_per_gpu": S, "n_features": NFEAT, "hypotheses": N_HYP,
                       "keypoints_per_frame_mean": float(cnt.mean()), "matches_per_pair_mean": mpass_mean,
                       "map_points_per_pair_mean": float(npt.mean()),
                       "parallelism": "frame-sharded x%d, RCCL gather of map points" % world},
            "roofline": {"bound": "hbm", "kernel": "k_fast", "achieved": round(achieved, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(achieved / PEAK_HBM_GBPS, 5), "hbm_frac": round(achieved / PEAK_HBM_GBPS, 5),
                         "valu_frac": round(valu_frac, 4) if valu_frac else None, "traffic": traffic,
                         "algorithmic_bytes": alg_bytes, "kernel_ms": round(crit_ms, 4),
                         "kernel_ms_source": "stand-alone (blur serialised)" if alone else "hipEvent span in the timed region",
                         "pipeline_achieved": round(total_alg / (ms_step * 1e-3) / 1e9, 2),
                         "note": "k_fast = the image kernel with the longest stand-alone time (the round-1 review's choice; "
                                 "k_match_lds is longer but touches 0.04 GB per launch: see roofline_per_stage); algorithmic bytes (SURVEY 8d: "
                                 "950 532 B per frame) / its own duration.  valu_frac = SQ_INSTS_VALU (profiles/r02_pmc_per_kernel.json) "
                                 "x 2 cycles / (1024 SIMDs x 2.4 GHz x duration); tools/ubench.hip measures 2.3 cycles only for add/sub/"
                                 "logic/shift-right/f32 add-mul and 4.5 for every other vector instruction at the occupancy these "
                                 "kernels run at, so an integer kernel tops out near 0.5 on this scale (DESIGN.md 4)"},
            "roofline_per_stage": per_kernel,
            "stage_ms_standalone": {k: round(v, 4) for k, v in alone.items()} if alone else None,
            "stage_ms": {k: round(v, 4) for k, v in per_stage.items()},
            "aux_stream_probe": dict(zip(("state", "fork_join_ms"), subs[0].ctx.aux_probe())),  # 1 = blur beside FAST on the aux stream
        }
        # Opt-in variant, timed outside the headline region on the same inputs: the matrix-core matcher
        # (VSLAM_AMD_MATCHER=mfma at context creation; identical results).  Its int8 operations (2 * 256 per descriptor
        # pair) are priced against the dense int8 MFMA peak (2 x the bf16 rate, MI355X_MICROARCH.md).
        if world == 1 and S == 1 and not args.no_optin:
            os.environ["VSLAM_AMD_MATCHER"] = "mfma"
            try:
                # (BENCH_OPTIN_MAIN=1, diagnostic: this context on the headline stream - the layout in which its aux stream did not
                #  run beside it)
                sb2 = SubBatch(0, n_pairs, stream=main if os.environ.get("BENCH_OPTIN_MAIN") else None)
            finally:
                os.environ.pop("VSLAM_AMD_MATCHER", None)
            for _ in range(max(1, args.warmup)):
                sb2.launch()
            torch.cuda.synchronize()
            acc2 = {}
            t1 = time.perf_counter()
            for _ in range(args.steps):
                sb2.launch()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            n2 = min(args.steps, V.TIMING_SLOTS)
            for back in range(n2):
                for name, ms in sb2.ctx.stage_times(back):
                    acc2[name] = acc2.get(name, 0.0) + ms
            m_ms = acc2.get("match_knn2_ratio", 0.0) / n2
            c64 = sb2.counts.cpu().numpy().astype(np.float64)
            ops = float((c64[:-1] * c64[1:]).sum()) * 512.0
            tops = ops / (m_ms * 1e-3) / 1e12 if m_ms > 0 else 0.0
            out["matcher_mfma_optin"] = {"value": round(B * args.steps / el2, 2), "unit": "frames/s", "aux_stream_probe": list(sb2.ctx.aux_probe()),
                                         "ms_per_step": round(el2 / args.steps * 1e3, 3), "match_ms": round(m_ms, 4),
                                         "roofline": {"bound": "mfma", "kernel": "k_match_mfma", "achieved": round(tops, 1),
                                                      "peak": 5000.0, "unit": "TOP/s", "frac": round(tops / 5000.0, 4),
                                                      "dtype": "int8"}}
        if world == 1 and S == 1 and not args.no_extras:
            # (c) the "next" rows of SURVEY 8f on the same frames: the tracking step as one batched call (MO_MODE_TRACK: ratio test,
            #     displacement filter at 2 % of (w + h) / 2, 2 x median distance filter, 8-point E RANSAC at 1 px - tracker.py:214-254)
            #     and one keyframe pair through the fundamental-matrix RANSAC of local_mapper.py:116-149 (host API)
            sbt = subs[0]  # the headline context with its mode switched (a SECOND context on this stream ran 40 % slower here: DESIGN.md 7)
            sel = torch.zeros((n_pairs, CAP, 2), dtype=torch.int32, device=dev)
            seln = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
            sbt.io.mode = V.MODE_TRACK; sbt.io.disp_frac = 0.02; sbt.io.thr_px = 1.0
            sbt.io.d_sel_idx = sel.data_ptr(); sbt.io.d_sel_n = seln.data_ptr()
            t_pw = time.perf_counter()  # (clock ramp after the PCIe-bound legs above, as ahead of the headline region)
            while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
                sbt.launch()
                torch.cuda.synchronize()
            for _ in range(max(1, args.warmup)):
                sbt.launch()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                sbt.launch()
            torch.cuda.synchronize()
            elt = time.perf_counter() - t1
            acct, nt_ = {}, min(args.steps, V.TIMING_SLOTS)
            for back in range(nt_):
                for name, ms in sbt.ctx.stage_times(back):
                    acct[name] = acct.get(name, 0.0) + ms / nt_
            kp0 = sbt.kps[0, :int(sbt.counts[0].item())].cpu().numpy()
            kp1 = sbt.kps[1, :int(sbt.counts[1].item())].cpu().numpy()
            keep = sbt.mpass[0, :len(kp0)].cpu().numpy().astype(bool)
            tr = sbt.midx[0, :len(kp0), 0].cpu().numpy()
            p1 = kp0[keep][:, :2].copy(); p2 = kp1[tr[keep]][:, :2].copy()
            ctxf = V.default_context()
            ctxf.find_fundamental(p1, p2, 3.0)
            tsf = []
            for _ in range(10):
                t = time.perf_counter(); ctxf.find_fundamental(p1, p2, 3.0); tsf.append((time.perf_counter() - t) * 1e3)
            out["next_rows"] = {"track_mode": {"value": round(B * args.steps / elt, 2), "unit": "frames/s", "ms_per_step": round(elt / args.steps * 1e3, 3),
                                               "kept_matches_per_pair_mean": float(seln.float().mean().item()),
                                               "stage_ms": {k: round(v, 4) for k, v in acct.items()}},
                                "find_fundamental_ms": {"value": round(sorted(tsf)[len(tsf) // 2], 3), "correspondences": int(len(p1)),
                                                        "note": "one pair, host API (H2D + 4096-hypothesis F RANSAC + D2H + sync)"}}
            sbt.io.mode = V.MODE_INIT; sbt.io.thr_px = 3.0; sbt.io.d_sel_idx = None; sbt.io.d_sel_n = None
        if world == 1 and S == 1 and not args.no_extras:
            # (a) PCIe-inclusive rate: every step first copies its frames from pinned host memory into HBM on the same stream
            #     (307 200 B per frame over Gen5 x16); never the headline value
            #     Both legs run on the HEADLINE context with its input pointer switched (a second context on this stream ran up to 40 %
            #     slower here - DESIGN.md 7).
            host = torch.empty((nb, H, W), dtype=torch.uint8).pin_memory()
            host.copy_(frames.cpu())
            bufs = [torch.empty_like(frames), torch.empty_like(frames)]
            sbh = subs[0]
            def h2d_step():
                bufs[0].copy_(host, non_blocking=True)  # (current stream = the context's stream)
                sbh.launch()
            sbh.io.d_gray = bufs[0].data_ptr()
            for _ in range(max(1, args.warmup)):
                h2d_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                h2d_step()
            torch.cuda.synchronize()
            el3 = time.perf_counter() - t1
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
                freed[k].record(main)
            torch.cuda.synchronize()
            n_ov = args.steps + 2
            t1 = time.perf_counter()
            enqueue_copy(0)
            for i in range(n_ov):
                k = i & 1
                if i + 1 < n_ov:
                    enqueue_copy(1 - k)
                main.wait_event(copied[k])
                sbh.io.d_gray = bufs[k].data_ptr()
                sbh.launch()
                freed[k].record(main)
            torch.cuda.synchronize()
            el4 = time.perf_counter() - t1
            sbh.io.d_gray = frames.data_ptr()
            out["h2d_inclusive"] = {"value": round(B * args.steps / el3, 2), "unit": "frames/s", "ms_per_step": round(el3 / args.steps * 1e3, 3),
                                    "overlapped_value": round(B * n_ov / el4, 2), "overlapped_ms_per_step": round(el4 / n_ov * 1e3, 3),
                                    "note": "pinned host -> HBM copy of the batch (%.1f MB) inside every step; 'overlapped': double-buffered, "
                                            "the copy of the next batch runs on its own stream beside the compute of the current one"
                                            % (nb * H * W / 1e6)}
            # (b) single-frame latency through the drop-in classes, host arrays in and Python objects out: what the reference's
            #     Tracker would see per call (BASELINE config 2; extract_features(distributed=True) is Tracker's default path)
            from orbslam2.extractor import ORBExtractor
            from orbslam2.matcher import DescriptorMatcher
            f0, f1 = frames[0].cpu().numpy(), frames[1].cpu().numpy()
            ex = ORBExtractor(n_features=NFEAT)
            mt = DescriptorMatcher("bruteforce-hamming", ratio_threshold=0.75)
            def med_ms(fn, n):
                fn()
                ts = []
                for _ in range(n):
                    t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
                return round(sorted(ts)[len(ts) // 2], 3)
            (k0, d0), (k1, d1) = ex.detect_and_compute(f0), ex.detect_and_compute(f1)
            import vslam_amd as V2
            ctx1 = V2.default_context()
            out["single_frame_ms"] = {
                "detect_and_compute": med_ms(lambda: ex.detect_and_compute(f0), 20),
                "detect_and_compute_native_arrays": med_ms(lambda: ctx1.orb_detect_compute(f0, ex.orb.prm), 20),
                "match_2000x2000": med_ms(lambda: mt.match(d0, d1), 20),
                "match_native_arrays": med_ms(lambda: ctx1.match_knn2_ratio(d0, d1, 0.75), 20),
                "extract_features_distributed": med_ms(lambda: ex.extract_features(f0, distributed=True), 10),
                "note": "median wall ms per call, host numpy in / Python objects out (H2D + kernels + D2H + sync); the *_native_arrays "
                        "rows stop at numpy arrays (no KeyPoint / DMatch objects)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames[:args.cpu_frames].cpu().numpy(), args.cpu_frames, K)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for sb in subs:
        sb.ctx.close()


if __name__ == "__main__":
    main()

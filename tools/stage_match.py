"""Times mo_dev_match_pairs alone on random descriptors (profiling helper: run it under rocprofv3).
usage: python3 tools/stage_match.py [frames] [descriptors per frame] [repeats]"""
import importlib, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "visual-slam_amd"))
import torch

G = importlib.import_module("vslam_amd")


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda:0")
    ctx = G.Context(device=0, max_w=64, max_h=64, max_batch=1)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    desc = torch.randint(0, 256, (frames, n, 32), dtype=torch.uint8, device=dev)
    counts = torch.full((frames,), n, dtype=torch.int32, device=dev)
    pairs = frames - 1
    qf = torch.arange(pairs, dtype=torch.int32, device=dev)
    tf = qf + 1
    idx = torch.empty((pairs, n, 2), dtype=torch.int32, device=dev)
    dist = torch.empty_like(idx)
    ok = torch.empty((pairs, n), dtype=torch.uint8, device=dev)

    def run():
        ctx._check(ctx.lib.mo_dev_match_pairs(ctx.h, desc.data_ptr(), counts.data_ptr(), n, qf.data_ptr(), tf.data_ptr(), pairs,
                                              0.75, idx.data_ptr(), dist.data_ptr(), ok.data_ptr()))

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("match %d pairs x %d x %d: %.4f ms  (%.1f G descriptor pairs/s)" % (pairs, n, n, ms, pairs * n * n / ms / 1e6))


if __name__ == "__main__":
    main()

#!/bin/bash
# tools/ab_stream_lanes.sh name ... -- ON the GPU box: tools/stream_rate.py --repeat 5 per build ("cur" = in-tree, else visual-slam_amd/variants/lib<name>.so),
# interleaved over two rounds on the SAME box (the example driver's rate differs by box): examples/run_frames.py --batch 64 / --grid / --batch 128
for round in 1 2; do
for v in "$@"; do
    lib=visual-slam_amd/variants/lib$v.so; [ "$v" = cur ] && lib=visual-slam_amd/libvslam_amd.so
    VSLAM_AMD_LIB=$lib python tools/stream_rate.py --repeat 5 2>/dev/null | grep "over 5 runs" | sed "s/^/$v round $round: /"
done; done

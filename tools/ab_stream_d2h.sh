#!/bin/bash
# tools/ab_stream_d2h.sh name ... -- ON the GPU box, each build in fresh processes: submits longer than 2 ms (tools/stream_first.py), rate per quarter of the first and later
# streams (tools/stream_ramp.py), the example's repeated rate (tools/stream_rate.py --repeat 4)
for round in 1 2; do
for v in "$@"; do
    lib=visual-slam_amd/variants/lib$v.so; [ "$v" = cur ] && lib=visual-slam_amd/libvslam_amd.so
    VSLAM_AMD_LIB=$lib python tools/stream_first.py 2>/dev/null | tail -1 | sed "s/^/$v r$round stalls: /"
    VSLAM_AMD_LIB=$lib python tools/stream_ramp.py 2>/dev/null | head -3 | sed "s/^/$v r$round ramp:   /"
    VSLAM_AMD_LIB=$lib python tools/stream_rate.py --repeat 4 2>/dev/null | grep "over 4 runs" | sed "s/^/$v r$round example: /"
done; done

#!/bin/bash
# tools/ab_single.sh name ... -- ON the GPU box, per build ("cur" = in-tree): the bit-exact single-frame tests, then tools/single_frame_probe.py three times interleaved:
# detect_and_compute wall, the device stage spans, tracked frame on both detectors
O=${AB_OUT:-gpurun_out/ab_single}; mkdir -p $O
for v in "$@"; do
    lib=visual-slam_amd/variants/lib$v.so; [ "$v" = cur ] && lib=visual-slam_amd/libvslam_amd.so
    VSLAM_AMD_LIB=$lib timeout -k 10 400 python -m pytest tests/test_gpu_orb.py tests/test_gpu_frame_api.py tests/test_gpu_front_single.py -x -q -m gpu > $O/tests_$v.log 2>&1
    printf "%-10s parity: %s\n" $v "$(tail -1 $O/tests_$v.log)"
done
for round in 1 2 3; do for v in "$@"; do
    lib=visual-slam_amd/variants/lib$v.so; [ "$v" = cur ] && lib=visual-slam_amd/libvslam_amd.so
    VSLAM_AMD_LIB=$lib python tools/single_frame_probe.py --json $O/probe_$v.json > /dev/null 2>&1
    python3 - $v $O/probe_$v.json <<'PY'
import json, sys
n = json.load(open(sys.argv[2])); d = n["detect_and_compute"]; e = d["device_ms_with_events"]
print("%-10s detect %.4f ms | fast %.4f select %.4f describe %.4f pyramid %.4f | tracker_frame %.4f grid %.4f" % (
    sys.argv[1], d["wall_ms"], e["fast_nms"], e["select_harris"], e["angle_rbrief"], e["pyramid"], n["tracker_frame_class_wall_ms"], n["tracker_frame_grid_class_wall_ms"]), flush=True)
PY
done; done

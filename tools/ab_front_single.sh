#!/bin/bash
# tools/ab_front_single.sh name ... -- ON the GPU box: per build ("cur" = in-tree, else visual-slam_amd/variants/lib<name>.so) the parity tests of the
# single-frame pyramid + blur kernel, then tools/single_frame_probe.py twice: detect_and_compute wall and the pyramid stage's hipEvent span
O=${AB_OUT:-gpurun_out/ab_fs}; mkdir -p $O
for round in 1 2; do
for v in "$@"; do
    lib=visual-slam_amd/variants/lib$v.so; [ "$v" = cur ] && lib=visual-slam_amd/libvslam_amd.so
    if [ $round = 1 ]; then
        VSLAM_AMD_LIB=$lib timeout -k 10 240 python -m pytest tests/test_gpu_front_single.py -x -q > $O/tests_$v.log 2>&1
        printf "%-10s parity: %s\n" $v "$(tail -1 $O/tests_$v.log)"
    fi
    VSLAM_AMD_LIB=$lib python tools/single_frame_probe.py --json $O/probe_$v.json > $O/probe_$v.log 2>&1
    python3 - $v $O/probe_$v.json <<'PY'
import json, sys
n = json.load(open(sys.argv[2])); d = n["detect_and_compute"]
print("%-10s detect_and_compute wall %.4f ms (class %.4f) | pyramid stage %.4f ms | tracker_frame %.4f  initialize %.4f" % (
    sys.argv[1], d["wall_ms"], d["class_wall_ms"], d["device_ms_with_events"]["pyramid"], n["tracker_frame_class_wall_ms"], n["initialize_class_wall_ms"]), flush=True)
PY
done; done

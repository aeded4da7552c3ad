#!/bin/bash
# A/B of k_select workgroup sizes (variants/libsel512.so, libsel1024.so against the in-tree 256): batched step and single-frame probe
O=gpurun_out/r04d; mkdir -p $O
bash tools/ab_lib.sh cur sel512 sel1024 > $O/ab_sel_threads.txt 2>&1
for v in cur sel512 sel1024; do
  lib=visual-slam_amd/variants/lib$v.so; [ "$v" = cur ] && lib=visual-slam_amd/libvslam_amd.so
  VSLAM_AMD_LIB=$lib python tools/single_frame_probe.py --iters 30 --json $O/probe_$v.json > /dev/null 2>&1
  python - <<PY >> $O/ab_sel_threads.txt
import json; d=json.load(open("$O/probe_$v.json"))
print("$v", "detect wall", d["detect_and_compute"]["wall_ms"], "select", d["detect_and_compute"]["device_ms_with_events"].get("select_harris"), "tracker_frame", d["tracker_frame_class_wall_ms"])
PY
done
cat $O/ab_sel_threads.txt

#!/bin/bash
# diagnostic variant of k_gftt_cell: frame 3 prints, for cells 0 / 27 / 63, the s_memtime phase deltas of thread 0 (never shipped)
# use: tools/gftt_stamps.sh && VSLAM_AMD_LIB=visual-slam_amd/variants/libgftt_stamps.so python3 tools/grid_probe.py 2
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
tmp=$(mktemp -d /tmp/abl.XXXX)
mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
rm -rf "$tmp/visual-slam_amd/csrc/_obj"
python3 - "$tmp/visual-slam_amd/csrc/gftt_kernels.hip" <<'PY'
import sys
f = sys.argv[1]
s = open(f).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("    const int x0 = cj * cw, y0 = ci * ch, x1 = x0 + cw, y1 = y0 + ch;\n", "    const int x0 = cj * cw, y0 = ci * ch, x1 = x0 + cw, y1 = y0 + ch;\n    unsigned long long T0 = __builtin_amdgcn_s_memtime();\n")
rep("    // 1. cell maximum (minMaxLoc with the cell mask)\n", "    unsigned long long T1 = __builtin_amdgcn_s_memtime();\n")
rep("    // 2. candidates: above threshold", "    unsigned long long T2 = __builtin_amdgcn_s_memtime();\n    // 2. candidates: above threshold")
rep("    const int n_all = s_n;", "    unsigned long long T3 = __builtin_amdgcn_s_memtime();\n    const int n_all = s_n;")
rep("    auto sort_and_pick = [&](int n) {", "    unsigned long long T4 = 0;\n    auto sort_and_pick = [&](int n) {")
rep("        if (wv == 0) {\n            const float md2", "        T4 = __builtin_amdgcn_s_memtime();\n        if (wv == 0) {\n            const float md2")
rep("    if (wv == 0) {  // accepted corners in acceptance order", "    unsigned long long T5 = __builtin_amdgcn_s_memtime();\n    if (blockIdx.y == 3 && tid == 0 && (cell == 0 || cell == 27 || cell == 63)) printf(\"STAMP cell %d n %d nacc %d: tile %llu max %llu list %llu sort %llu pick %llu total %llu (s_memtime ticks)\\n\", cell, n_all, nacc, T1 - T0, T2 - T1, T3 - T2, T4 - T3, T5 - T4, T5 - T0);\n    if (wv == 0) {  // accepted corners in acceptance order")
open(f, "w").write(s)
PY
make -C "$tmp/visual-slam_amd/csrc" -j8 2>&1 | grep -E "error" -A3 | head
cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libgftt_stamps.so"
rm -rf "$tmp"; echo built stamps

#!/bin/bash
# diagnostic variant of k_select: frame $STAMP_FRAME (default 3; 0 for single-frame calls) prints, per level, the s_memtime phase deltas of
# thread 0 (never shipped) -> visual-slam_amd/variants/libsel_stamps$STAMP_FRAME.so
set -e
F=${STAMP_FRAME:-3}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
tmp=$(mktemp -d /tmp/abl.XXXX)
mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
rm -rf "$tmp/visual-slam_amd/csrc/_obj"
python3 - "$tmp/visual-slam_amd/csrc/orb_kernels.hip" $F <<'PY'
import sys
f = sys.argv[1]
FR = sys.argv[2]
s = open(f).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("    __syncthreads();\n    int N2 = replay::wg_retain_best<NT, uint64_t>(B, N1, lv.quota,", "    __syncthreads();\n    unsigned long long TH = __builtin_amdgcn_s_memtime();\n    int N2 = replay::wg_retain_best<NT, uint64_t>(B, N1, lv.quota,")
rep("    if (N2 > lv.fin_cap) {\n        if (tid == 0) { atomicOr(&flags[0], 1);", "    unsigned long long TR = __builtin_amdgcn_s_memtime();\n    if (blockIdx.x == " + FR + " && tid == 0) printf(\"STAMP2 L=%d N1=%d N2=%d harris_end %llu retain2 %llu\\n\", (int)blockIdx.y, N1, N2, TH, TR - TH);\n    if (N2 > lv.fin_cap) {\n        if (tid == 0) { atomicOr(&flags[0], 1);")
rep("    const int L = level0 + blockIdx.y, frame = blockIdx.x, tid = threadIdx.x;  // dispatch order: all frames of the finest level first\n", "    const int L = level0 + blockIdx.y, frame = blockIdx.x, tid = threadIdx.x;\n    unsigned long long T0 = __builtin_amdgcn_s_memtime();\n")
rep("    // pass 1: retainBest(2 * quota) on the FAST score\n", "    unsigned long long T1 = __builtin_amdgcn_s_memtime();\n")
rep("    // the Harris records (and their rpos / ballots) go behind the surviving FAST records when both fit the window\n", "    unsigned long long T2 = __builtin_amdgcn_s_memtime();\n")
rep("    else select_harris<NT>(P, lv, img, gA, gB, N1, g_rpos, g_bl, fin, fin_cnt_out, flags, &s_ws, s_hw, L);\n}", "    else select_harris<NT>(P, lv, img, gA, gB, N1, g_rpos, g_bl, fin, fin_cnt_out, flags, &s_ws, s_hw, L);\n    unsigned long long T3 = __builtin_amdgcn_s_memtime();\n    if (frame == " + FR + " && tid == 0) printf(\"STAMP L=%d N=%d N1=%d a_lds=%d  start %llu gather %llu retain1 %llu harris+retain2+write %llu total %llu (T2 %llu)\\n\", L, N, N1, (int)a_lds, T0, T1 - T0, T2 - T1, T3 - T2, T3 - T0, T2);\n}")
open(f, "w").write(s)
PY
make -C "$tmp/visual-slam_amd/csrc" -j8 2>&1 | grep -E "error" -A3 | head
cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libsel_stamps$F.so"
rm -rf "$tmp"; echo built stamps

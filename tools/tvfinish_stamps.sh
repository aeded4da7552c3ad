#!/bin/bash
# diagnostic variant of k_tv_finish: pair $STAMP_PAIR (default 3; 0 for the single-pair host calls) prints s_memtime phase deltas of thread 0
# (never shipped) -> visual-slam_amd/variants/libtvf_stamps$STAMP_PAIR.so
set -e
PR=${STAMP_PAIR:-3}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
tmp=$(mktemp -d /tmp/abl.XXXX)
mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
rm -rf "$tmp/visual-slam_amd/csrc/_obj"
python3 - "$tmp/visual-slam_amd/csrc/twoview_kernels.hip" $PR <<'PY'
import sys
f = sys.argv[1]
PR = sys.argv[2]
s = open(f).read()
k0 = s.index("__global__ __launch_bounds__(TVF_BLOCK) void k_tv_finish")
head, body = s[:k0], s[k0:]
def ins(after, text, before=False):
    global body
    assert after in body, after
    body = body.replace(after, (text + after) if before else (after + text), 1)
ins("    const int pair = blockIdx.x, tid = threadIdx.x;\n", "    unsigned long long T[11]; int nlo = 0; for (int q = 0; q < 11; q++) T[q] = 0; T[0] = __builtin_amdgcn_s_memtime();\n")
ins("    // ---- local optimisation: least-squares 8-point refits", "    T[1] = __builtin_amdgcn_s_memtime();\n", before=True)
ins("    int c_prev = -1;\n", "    T[2] = __builtin_amdgcn_s_memtime();\n")
ins("        const int c = block_sum_i(cnt, s_redi);\n", "        nlo++;\n")
ins("        v4d gram = {0.0, 0.0, 0.0, 0.0};\n", "        unsigned long long L0 = __builtin_amdgcn_s_memtime();\n", before=True)
ins("        const int c = block_sum_i(cnt, s_redi);\n", "        unsigned long long L1 = __builtin_amdgcn_s_memtime(); T[8] += L1 - L0;\n", before=True)
ins("        // smallest eigenvector of the normal matrix on wavefront 0", "        unsigned long long L2 = __builtin_amdgcn_s_memtime();\n", before=True)
ins("            const bool conv = wave_smallest_eigvec9(s_N, s_M, s_x, tid);\n", "            T[10] += __builtin_amdgcn_s_memtime() - L2;\n")
ins("        if (s_stop) break;\n", "        unsigned long long L3 = __builtin_amdgcn_s_memtime(); T[9] += L3 - L2;\n", before=True)
ins("    if (a.model) {  // fundamental matrix: mask", "    T[3] = __builtin_amdgcn_s_memtime();\n", before=True)
ins("    // ---- final RANSAC mask + cheirality vote", "    T[4] = __builtin_amdgcn_s_memtime();\n", before=True)
ins("    int g[4];\n", "    T[5] = __builtin_amdgcn_s_memtime();\n", before=True)
ins("    const int win = s_win;\n", "    T[6] = __builtin_amdgcn_s_memtime();\n")
# end of kernel: last closing brace of body up to next kernel
end = body.index("\n}\n", body.index("        if (inl_out) inl_out[o] = 1;"))
body = body[:end] + "\n    __builtin_amdgcn_s_waitcnt(0); T[7] = __builtin_amdgcn_s_memtime();\n    if (pair == " + PR + " && tid == 0) printf(\"STAMP m=%d lo_iters=%d  clear+lookup %llu  first_count %llu  lo_loop %llu  decompose %llu  cheirality %llu  vote %llu  triangulate %llu  total %llu | in lo_loop: gram passes %llu  eigenvector (wavefront) + projection (lane 0) %llu (eigenvector alone %llu)\\n\", m, nlo, T[1]-T[0], T[2]-T[1], T[3]-T[2], T[4]-T[3], T[5]-T[4], T[6]-T[5], T[7]-T[6], T[7]-T[0], T[8], T[9], T[10]);" + body[end:]
open(f, "w").write(head + body)
PY
make -C "$tmp/visual-slam_amd/csrc" -j8 2>&1 | grep -E "error" -A3 | head
cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libtvf_stamps$PR.so"
rm -rf "$tmp"; echo built stamps

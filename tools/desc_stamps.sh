#!/bin/bash
# diagnostic variant of k_describe: a few wavefronts print their s_memtime phase deltas (never shipped)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
tmp=$(mktemp -d /tmp/abl.XXXX)
mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
rm -rf "$tmp/visual-slam_amd/csrc/_obj"
python3 - "$tmp/visual-slam_amd/csrc/orb_kernels.hip" <<'PY'
import sys
f = sys.argv[1]
s = open(f).read()
def ins(after, text):
    global s
    assert after in s, after
    s = s.replace(after, after + text, 1)
ins("    __shared__ int s_lv[MO_MAX_LEVELS][DLV_N];\n", "    unsigned long long T0 = __builtin_amdgcn_s_memtime(), T1 = 0, T2 = 0, T3 = 0, T4 = 0, T5 = 0, T6 = 0;\n")
ins("    const FinalKp fk = fin_all[(size_t)frame * P.fin_stride + lv.fin_off + idx];\n    const int x = fk.x, y = fk.y;\n", "    asm volatile(\"\" :: \"v\"(x)); T1 = __builtin_amdgcn_s_memtime();\n")
ins("    patch_store<31, 9>(s_p, DP_RAW_PITCH, gl, raw);\n", "    __builtin_amdgcn_s_waitcnt(0); T2 = __builtin_amdgcn_s_memtime();\n")
ins("    const float angle = fast_atan2_deg((float)m01, (float)m10);\n", "    asm volatile(\"\" :: \"v\"(angle)); T3 = __builtin_amdgcn_s_memtime();\n")
ins("        patch_store<39, 11>(s_p, DP_BLR_PITCH, gl, blr);\n", "        __builtin_amdgcn_s_waitcnt(0); T4 = __builtin_amdgcn_s_memtime();\n")
a = "        *(uint16_t*)(desc + ((size_t)frame * cap + k) * 32 + 2 * gl) = rbrief_u16(s_p, DP_BLR_PITCH, 19 + offb, 19, angle, gl);\n"
assert a in s
s = s.replace(a, "        const uint16_t dd = rbrief_u16(s_p, DP_BLR_PITCH, 19 + offb, 19, angle, gl); asm volatile(\"\" :: \"v\"((int)dd)); T5 = __builtin_amdgcn_s_memtime();\n        *(uint16_t*)(desc + ((size_t)frame * cap + k) * 32 + 2 * gl) = dd;\n        __builtin_amdgcn_s_waitcnt(0); T6 = __builtin_amdgcn_s_memtime();\n        if ((blockIdx.x == 5 || blockIdx.x == 60) && (blockIdx.y == 3 || blockIdx.y == 100 || blockIdx.y == 200) && threadIdx.x == 0)\n            printf(\"STAMP b(%d,%d) prologue+kp %llu  loads+rawstore %llu  ic+atan %llu  blurwait+store %llu  sincos+brief %llu  descstore %llu  total %llu\\n\", blockIdx.x, blockIdx.y, T1 - T0, T2 - T1, T3 - T2, T4 - T3, T5 - T4, T6 - T5, T6 - T0);\n")
open(f, "w").write(s)
PY
make -C "$tmp/visual-slam_amd/csrc" -j8 2>&1 | grep -E "error" -A3 | head
cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libdesc_stamps.so"
rm -rf "$tmp"; echo built stamps

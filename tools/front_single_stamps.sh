#!/bin/bash
# diagnostic variant of k_front_single: tiles 0 (corner), nx - 1 (right border), an interior one and the last print s_memtime deltas of thread 0
# at every workgroup barrier (never shipped) -> visual-slam_amd/variants/libfs_stamps.so; run: VSLAM_AMD_LIB=... python tools/stamps_single.py
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
tmp=$(mktemp -d /tmp/abl.XXXX)
mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
rm -rf "$tmp/visual-slam_amd/csrc/_obj"
python3 - "$tmp/visual-slam_amd/csrc/front_single.hip" <<'PY'
import sys
f = sys.argv[1]
s = open(f).read()
k0 = s.index("__global__ __launch_bounds__(FS_NT) void k_front_single")
head, body = s[:k0], s[k0:]
def ins(anchor, text, before=False, nth=1):
    global body
    assert body.count(anchor) >= nth, anchor
    pos = -1
    for _ in range(nth): pos = body.index(anchor, pos + 1)
    at = pos if before else pos + len(anchor)
    body = body[:at] + text + body[at:]
ins("    const int tid = threadIdx.x, frame = blockIdx.z, nl = P.nlevels;\n",
    "    unsigned long long T[20]; for (int q = 0; q < 20; q++) T[q] = 0; int ns = 0; T[ns++] = __builtin_amdgcn_s_memtime();\n")
ins("    // ---- levels 1 .. :", "    T[ns++] = __builtin_amdgcn_s_memtime();\n", before=True)                               # level-0 box + tables
ins("        __syncthreads();\n    }\n    if (!want_blur) return;\n", "", before=True)
body = body.replace("        __syncthreads();\n    }\n    if (!want_blur) return;\n", "        __syncthreads();\n        T[ns++] = __builtin_amdgcn_s_memtime();\n    }\n    if (!want_blur) return;\n", 1)
def rep(a, b):
    global body
    assert body.count(a) == 1, a
    body = body.replace(a, b, 1)
rep("    __syncthreads();\n\n    // ---- 7x7 Gaussian", "    __syncthreads();\n    T[ns++] = __builtin_amdgcn_s_memtime();\n\n    // ---- 7x7 Gaussian")   # pads filled
rep("    __syncthreads();\n    for (int L = 0; L < nl; L++) {\n        const FsBox b = box[L];\n        const LevelInfo lv = P.lv[L];\n        const int qw",
    "    __syncthreads();\n    T[ns++] = __builtin_amdgcn_s_memtime();\n    for (int L = 0; L < nl; L++) {\n        const FsBox b = box[L];\n        const LevelInfo lv = P.lv[L];\n        const int qw")
end = body.index("\n}\n", body.index("*(uint32_t*)(out + (size_t)(b.oy0 + r) * lv.bpitch"))
pr = ('\n    __builtin_amdgcn_s_waitcnt(0); T[ns++] = __builtin_amdgcn_s_memtime();\n'
      '    if (tid == 0 && frame == 0 && ((blockIdx.x == 0 && blockIdx.y == 0) || (blockIdx.x == gridDim.x - 1 && blockIdx.y == 0) || (blockIdx.x == 5 && blockIdx.y == 4) || (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1))) {\n'
      '        printf("FS_STAMP tile %d box0 %dx%d own %dx%d | header+level0+tables %llu  levels", (int)(blockIdx.y * gridDim.x + blockIdx.x), box[0].ew, box[0].eh, box[0].ox1 - box[0].ox0, box[0].oy1 - box[0].oy0, T[1] - T[0]);\n'
      '        for (int q = 2; q < 1 + nl; q++) printf(" %llu", T[q] - T[q - 1]);\n'
      '        printf("  pads %llu  blur rows %llu  blur columns %llu  total %llu\\n", T[1 + nl] - T[nl], T[2 + nl] - T[1 + nl], T[3 + nl] - T[2 + nl], T[3 + nl] - T[0]);\n'
      '    }')
body = body[:end] + pr + body[end:]
open(f, "w").write(head + body)
PY
# FS_TWICE=1: the launch is issued twice back to back, so the second set of stamps is the kernel on warm instruction caches
if [ -n "$FS_TWICE" ]; then
python3 - "$tmp/visual-slam_amd/csrc/api.hip" <<'PY'
import sys
f = sys.argv[1]; s = open(f).read()
a = "        if ((rc = orb_launch_front_single(c, d_gray, batch, d_desc != nullptr))) return rc;\n"
assert s.count(a) == 1
open(f, "w").write(s.replace(a, a + a))
PY
fi
make -C "$tmp/visual-slam_amd/csrc" -j8 2>&1 | grep -E "error" -A3 | head
cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libfs_stamps.so"
rm -rf "$tmp"; echo built stamps

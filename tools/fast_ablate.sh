#!/bin/bash
# tools/fast_ablate.sh: timing-only variants of k_fast that stop after a phase (results are garbage; never shipped):
#   s1 = after staging + zero fill, s2a = compass pre-test without scoring, s2 = after scoring, s3 = after the NMS
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
for v in s1 s2a s2 s3; do
  tmp=$(mktemp -d /tmp/abl.XXXX)
  mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
  cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
  rm -rf "$tmp/visual-slam_amd/csrc/_obj"
  f="$tmp/visual-slam_amd/csrc/orb_kernels.hip"
  python3 - "$f" "$v" <<'PY'
import sys
f, v = sys.argv[1], sys.argv[2]
s = open(f).read()
stage_end = "    if (tid == 0) s_ncorner = 0;\n    __syncthreads();\n"
assert stage_end in s
if v == "s1":
    s = s.replace(stage_end, stage_end + "    if (P.nlevels > 0) return;\n", 1)
elif v == "s2a":
    s = s.replace("                score_pair(wq[2 * lane], wq[2 * lane + 1], true, true);", "                asm volatile(\"\" :: \"v\"(wq[2 * lane]));")
    s = s.replace("            score_pair(wq[i0], wq[i1], 2 * lane < qn, 2 * lane + 1 < qn);", "            asm volatile(\"\" :: \"v\"(wq[i0] + wq[i1]));")
    s = s.replace("    // ---- 3. NMS + border filter on the listed corners", "    if (P.nlevels > 0) return;\n    // ---- 3. NMS + border filter on the listed corners", 1)
elif v == "s2":
    s = s.replace("    // ---- 3. NMS + border filter on the listed corners", "    if (P.nlevels > 0) return;\n    // ---- 3. NMS + border filter on the listed corners", 1)
elif v == "s3":
    s = s.replace("    const int nwords = (nitems + 31) >> 5, wpt = (nwords + 255) >> 8;  // wpt <= 2", "    if (P.nlevels > 0) return;\n    const int nwords = (nitems + 31) >> 5, wpt = (nwords + 255) >> 8;  // wpt <= 2", 1)
open(f, "w").write(s)
PY
  make -C "$tmp/visual-slam_amd/csrc" -j8 >/dev/null 2>&1
  cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libfast_$v.so"
  rm -rf "$tmp"; echo built $v
done

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
    for a, b in (("score_one(wq[qd + lane], 0, true);", 'asm volatile("" :: "v"(wq[qd + lane]));'),
                 ("score_one(wq[FAST_STACK - 1 - qb - lane], 0xFF, true);", 'asm volatile("" :: "v"(wq[FAST_STACK - 1 - qb - lane]));'),
                 ("if (qd > 0) score_one(wq[min(lane, qd - 1)], 0, lane < qd);", 'if (qd > 0) asm volatile("" :: "v"(wq[min(lane, qd - 1)]));'),
                 ("if (qb > 0) score_one(wq[FAST_STACK - 1 - min(lane, qb - 1)], 0xFF, lane < qb);", 'if (qb > 0) asm volatile("" :: "v"(wq[FAST_STACK - 1 - min(lane, qb - 1)]));')):
        assert a in s, a
        s = s.replace(a, b)
    a = "    // ---- 3. NMS + border filter on the listed corners"
    assert a in s
    s = s.replace(a, "    if (P.nlevels > 0) return;\n" + a, 1)
elif v == "s2":
    a = "    // ---- 3. NMS + border filter on the listed corners"
    assert a in s
    s = s.replace(a, "    if (P.nlevels > 0) return;\n" + a, 1)
elif v == "s3":
    a = "    const int nwords = (nitems + 31) >> 5, wpt = (nwords + NT - 1) / NT;  // wpt <= 2"
    assert a in s
    s = s.replace(a, "    if (P.nlevels > 0) return;\n" + a, 1)
open(f, "w").write(s)
PY
  make -C "$tmp/visual-slam_amd/csrc" -j8 2>&1 | grep -E "error" -A3 | head -5 || true
  cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libfast_$v.so"
  rm -rf "$tmp"; echo built $v
done

#!/bin/bash
# tools/tv_ablate.sh: timing-only variants of k_tv_hyp (garbage results): noproj = no essential projection, nogen = no 8-point solve
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
for v in noproj nogen; do
  tmp=$(mktemp -d /tmp/abl.XXXX)
  mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
  cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
  rm -rf "$tmp/visual-slam_amd/csrc/_obj"
  f="$tmp/visual-slam_amd/csrc/twoview_kernels.hip"
  python3 - "$f" "$v" <<'PY'
import sys
f, v = sys.argv[1], sys.argv[2]
s = open(f).read()
a = "        valid = eight_point(pts, E) && (a.model ? project_rank2(E) : project_essential(E));"
assert a in s
if v == "noproj":
    s = s.replace(a, "        valid = eight_point(pts, E);")
else:
    s = s.replace(a, "        for (int j = 0; j < 9; j++) E[j] = pts[j] + pts[j + 9]; valid = true;")
open(f, "w").write(s)
PY
  make -C "$tmp/visual-slam_amd/csrc" -j8 >/dev/null 2>&1
  cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libtv_$v.so"
  rm -rf "$tmp"; echo built $v
done

#!/bin/bash
# timing-only variants of k_describe: skel1 = prologue only, skel2 = prologue + keypoint record load + keypoint store (angle 0)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
for v in skel1 skel2; do
  tmp=$(mktemp -d /tmp/abl.XXXX)
  mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
  cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
  rm -rf "$tmp/visual-slam_amd/csrc/_obj"
  f="$tmp/visual-slam_amd/csrc/orb_kernels.hip"
  python3 - "$f" "$v" <<'PY'
import sys
f, v = sys.argv[1], sys.argv[2]
s = open(f).read()
if v == "skel1":
    a = "    if (L < 0 || k >= cap) return;  // uniform within the group; no barriers below\n"
    assert a in s
    s = s.replace(a, a + "    if (P.nlevels > 0) return;\n")
else:
    a = "    const float px = (float)x * lv.scale, py = (float)y * lv.scale;\n"
    assert a in s
    s = s.replace(a, a + "    if (P.nlevels > 0) { mo_keypoint* o2 = kps + (size_t)frame * cap + k; if (gl == 0) { o2->x = px; o2->y = py; o2->size = 31 * lv.scale; o2->angle = 0; o2->response = fk.response; o2->octave = L; o2->class_id = -1; } return; }\n")
open(f, "w").write(s)
PY
  make -C "$tmp/visual-slam_amd/csrc" -j8 >/dev/null 2>&1
  cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libdesc_$v.so"
  rm -rf "$tmp"; echo built $v
done

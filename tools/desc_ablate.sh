#!/bin/bash
# tools/desc_ablate.sh: timing-only variants of k_describe (results are garbage; never shipped)
#   d_noic = intensity-centroid loop skipped, d_nosincos = f32 sincosf instead of the f64 sincos, d_nobrief = rBRIEF tests skipped
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
for v in noic nosincos nobrief; do
  tmp=$(mktemp -d /tmp/abl.XXXX)
  mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
  cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
  rm -rf "$tmp/visual-slam_amd/csrc/_obj"
  f="$tmp/visual-slam_amd/csrc/orb_kernels.hip"
  python3 - "$f" "$v" <<'PY'
import sys
f, v = sys.argv[1], sys.argv[2]
s = open(f).read()
if v == "noic":
    a = "            for (int u = -d; u <= d; u++) {"
    assert a in s
    s = s.replace(a, "            for (int u = -d; u <= -d; u++) {")
elif v == "nosincos":
    a = "    sincos((double)angle, &sd, &cd);  // f64 then rounded to f32, as cv2's (float)cos(angle) / (float)sin(angle)\n    const float a = (float)cd, b = (float)sd;"
    assert a in s
    s = s.replace(a, "    float sf, cf; __sincosf(angle, &sf, &cf); sd = sf; cd = cf;\n    const float a = (float)cd, b = (float)sd;")
elif v == "nobrief":
    a = "    for (int k = 0; k < 16; k++) {\n        const int8_t* pt = &c_pattern[(gl * 16 + k) * 4];"
    assert a in s
    s = s.replace(a, "    for (int k = 0; k < 1; k++) {\n        const int8_t* pt = &c_pattern[(gl * 16 + k) * 4];")
open(f, "w").write(s)
PY
  make -C "$tmp/visual-slam_amd/csrc" -j8 >/dev/null 2>&1
  cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libdesc_$v.so"
  rm -rf "$tmp"; echo built $v
done

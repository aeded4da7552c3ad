#!/bin/bash
# timing-only variants of k_describe (garbage results): noload = window loads replaced by constants, nolds = no LDS patch stores
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
for v in noload nolds; do
  tmp=$(mktemp -d /tmp/abl.XXXX)
  mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
  cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
  rm -rf "$tmp/visual-slam_amd/csrc/_obj"
  f="$tmp/visual-slam_amd/csrc/orb_kernels.hip"
  python3 - "$f" "$v" <<'PY'
import sys
f, v = sys.argv[1], sys.argv[2]
s = open(f).read()
if v == "noload":
    a = "            reg[st] = *(const uint32_t*)(base + (st == PM::SA - 1 ? min(2 * st + ra, ROWS - 1) * pitch + 4 * ca : off));"
    assert a in s
    s = s.replace(a, "            reg[st] = (uint32_t)off * 2654435761u;")
    b = "            reg[PM::SA + st] = *(const uint32_t*)(base + (st == PM::SB - 1 ? min(PM::RB * st + rb, ROWS - 1) * pitch + 4 * cb : off));"
    assert b in s
    s = s.replace(b, "            reg[PM::SA + st] = (uint32_t)off * 40503u;")
else:
    a = "        if (st < PM::SA - 1 || 2 * st + ra < ROWS) *(uint32_t*)(da + st * 2 * dpitch) = reg[st];"
    assert a in s
    s = s.replace(a, "        if (st == 0) *(uint32_t*)(da + st * 2 * dpitch) = reg[st]; else asm volatile(\"\" :: \"v\"(reg[st]));")
    b = "        if (own && (st < PM::SB - 1 || PM::RB * st + rb < ROWS)) *(uint32_t*)(db + st * PM::RB * dpitch) = reg[PM::SA + st];"
    assert b in s
    s = s.replace(b, "        asm volatile(\"\" :: \"v\"(reg[PM::SA + st]));")
open(f, "w").write(s)
PY
  make -C "$tmp/visual-slam_amd/csrc" -j8 >/dev/null 2>&1
  cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libdesc_$v.so"
  rm -rf "$tmp"; echo built $v
done

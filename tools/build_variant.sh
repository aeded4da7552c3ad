#!/bin/bash
# tools/build_variant.sh <git-ref> <name>: build the library as it was at <git-ref> into visual-slam_amd/variants/lib<name>.so
# (git-ignored, travels to the GPU box) for A/B timing against the working tree: VSLAM_AMD_LIB=<path> python bench.py ...
set -e
ref=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/variant.XXXX)
git -C "$root" archive "$ref" visual-slam_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/visual-slam_amd/csrc" -j8 >/dev/null
mkdir -p "$root/visual-slam_amd/variants"
cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/lib$name.so"
rm -rf "$tmp"
echo "built visual-slam_amd/variants/lib$name.so from $ref"

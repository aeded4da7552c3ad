#!/bin/bash
# tools/build_defs.sh <name> "<extra compiler flags>": the working tree's library with extra -D switches -> visual-slam_amd/variants/lib<name>.so
# (git-ignored, travels to the GPU box) for A/B timing: tools/ab_lib.sh <name> ...
set -e
name=$1; defs=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/variant.XXXX)
mkdir -p $tmp/visual-slam_amd $tmp/include
cp -r $root/visual-slam_amd/csrc $tmp/visual-slam_amd/ && rm -rf $tmp/visual-slam_amd/csrc/_obj
cp $root/include/vslam_amd.h $tmp/include/
make -C $tmp/visual-slam_amd/csrc -j8 EXTRA="$defs" > /dev/null
mkdir -p $root/visual-slam_amd/variants
cp $tmp/visual-slam_amd/libvslam_amd.so $root/visual-slam_amd/variants/lib$name.so
rm -rf $tmp
echo "built visual-slam_amd/variants/lib$name.so ($defs)"

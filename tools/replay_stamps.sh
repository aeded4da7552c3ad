#!/bin/bash
# diagnostic variant: every round of wg_ls_nth_element of frame 0 prints (range size, cycles) -> variants/libreplay_stamps.so (never shipped)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/abl.XXXX)
mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
rm -rf "$tmp/visual-slam_amd/csrc/_obj"
python3 - "$tmp/visual-slam_amd/csrc/select_replay.h" <<'PY'
import sys
f = sys.argv[1]
s = open(f).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
# wg loop: stamp per round
rep("""    int depth = (31 - __clz(last - first)) * 2;
    while (last - first > 3) {
        if (last - first <= REPLAY_REG_MAX) {  // the rest on the registers of one wavefront
            if (tid < 64) wave_reg_introselect<T>(a, first, nth, last, depth, tid);
            __syncthreads();
            return;
        }""", """    int depth = (31 - __clz(last - first)) * 2;
    unsigned long long TS = __builtin_amdgcn_s_memtime();
    while (last - first > 3) {
        { unsigned long long TN = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && tid == 0) printf("ROUND lvl %d T%d n %d prev_cycles %llu\\n", (int)blockIdx.y, (int)sizeof(T), last - first, TN - TS); TS = TN; }
        if (last - first <= REPLAY_REG_MAX) {  // the rest on the registers of one wavefront
            if (tid < 64) wave_reg_introselect<T>(a, first, nth, last, depth, tid);
            __syncthreads();
            { unsigned long long TN = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && tid == 0) printf("REGTAIL lvl %d T%d cycles %llu\\n", (int)blockIdx.y, (int)sizeof(T), TN - TS); }
            return;
        }""")
rep("""    const T amb = a[n_points - 1];
    const int tail = n - n_points;
    if (tail < REPLAY_SERIAL_BELOW || tail > 65535) {
        if (tid == 0) ws->cut = partition_ge<T>(a, n_points, n, amb);""", """    const T amb = a[n_points - 1];
    const int tail = n - n_points;
    if (blockIdx.x == 0 && tid == 0) printf("TAILPART lvl %d T%d tail %d at %llu\\n", (int)blockIdx.y, (int)sizeof(T), tail, (unsigned long long)__builtin_amdgcn_s_memtime());
    if (tail < REPLAY_SERIAL_BELOW || tail > 65535) {
        if (tid == 0) ws->cut = partition_ge<T>(a, n_points, n, amb);""")
open(f, "w").write(s)
PY
make -C "$tmp/visual-slam_amd/csrc" -j8 2>&1 | grep -E "error" -A3 | head
mkdir -p "$root/visual-slam_amd/variants"
cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libreplay_stamps.so"
rm -rf "$tmp"; echo built replay stamps

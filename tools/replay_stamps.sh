#!/bin/bash
# diagnostic variant: every round of wg_ls_nth_element of frame 0 stores (range size, cycles) in LDS and k_select prints them ONCE at its
# end (a printf inside the loop costs ~ 1 M cycles and drowns the rounds) -> variants/libreplay_stamps.so (never shipped)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/abl.XXXX)
mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
rm -rf "$tmp/visual-slam_amd/csrc/_obj"
python3 - "$tmp/visual-slam_amd/csrc/select_replay.h" "$tmp/visual-slam_amd/csrc/orb_kernels.hip" <<'PY'
import sys
f, g = sys.argv[1], sys.argv[2]
s = open(f).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("namespace replay {\n", "namespace replay {\n__device__ int g_stamp_n[4]; __device__ unsigned long long g_stamp[4][2][96];\n#define STAMP(kind, val) do { if (blockIdx.x == 0 && tid == 0 && blockIdx.y < 2 && ws->sn < 96) { g_stamp[blockIdx.y * 2 + (sizeof(T) == 8)][0][ws->sn] = ((unsigned long long)(kind) << 32) | (unsigned)(val); g_stamp[blockIdx.y * 2 + (sizeof(T) == 8)][1][ws->sn] = __builtin_amdgcn_s_memtime(); ws->sn++; } } while (0)\n")
rep("    int cut;\n};", "    int cut;\n    int sn;\n};")
rep("""        if (last - first < WG_PARTITION_MIN) {
            if (tid < 64) wave_ls_introselect<T>(a, first, nth, last, depth, rpos, bl, tid);
            __syncthreads();
            return;
        }""", """        if (last - first < WG_PARTITION_MIN) {
            if (tid < 64) wave_ls_introselect<T>(a, first, nth, last, depth, rpos, bl, tid);
            __syncthreads();
            STAMP(3, last - first);
            return;
        }""")
rep("""    int depth = (31 - __clz(last - first)) * 2;
    while (last - first > 3) {
        // below the cooperative partition's range""", """    int depth = (31 - __clz(last - first)) * 2;
    if (tid == 0) ws->sn = 0;
    __syncthreads();
    STAMP(1, last - first);
    while (last - first > 3) {
        // below the cooperative partition's range""")
rep("""        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (tid == 0) ls_insertion_sort<T>(a, first, last);
    __syncthreads();
}""", """        if (cut <= nth) first = cut;
        else last = cut;
        STAMP(2, last - first);
    }
    if (tid == 0) ls_insertion_sort<T>(a, first, last);
    __syncthreads();
}""")
rep("""    const T amb = a[n_points - 1];
    const int tail = n - n_points;
    if (tail < REPLAY_SERIAL_BELOW || tail > 65535) {
        if (tid == 0) ws->cut = partition_ge<T>(a, n_points, n, amb);""", """    const T amb = a[n_points - 1];
    const int tail = n - n_points;
    STAMP(4, tail);
    if (tail < REPLAY_SERIAL_BELOW || tail > 65535) {
        if (tid == 0) ws->cut = partition_ge<T>(a, n_points, n, amb);""")
rep("""    int total_true = 0;
    wg_partition_step<NT, T>(
        a, n_points, n, [amb](T v) { return !R::ge(v, amb); }, [amb](T v) { return R::ge(v, amb); }, rpos, bl, tid, ws,
        &total_true);
    return n_points + total_true;""", """    int total_true = 0;
    wg_partition_step<NT, T>(
        a, n_points, n, [amb](T v) { return !R::ge(v, amb); }, [amb](T v) { return R::ge(v, amb); }, rpos, bl, tid, ws,
        &total_true);
    STAMP(5, total_true);
    if (blockIdx.x == 0 && tid == 0 && blockIdx.y < 2) g_stamp_n[blockIdx.y * 2 + (sizeof(T) == 8)] = ws->sn;
    return n_points + total_true;""")
open(f, "w").write(s)
k = open(g).read()
a = "    else select_harris<NT>(P, lv, img, gA, gB, N1, g_rpos, g_bl, fin, fin_cnt_out, flags, &s_ws, s_hw, L);\n}"
assert a in k
k = k.replace(a, a[:-2] + """
    __syncthreads();
    if (frame == 0 && tid == 0 && blockIdx.y < 2)
        for (int q = 0; q < 2; q++) {
            const int id = blockIdx.y * 2 + q, n = replay::g_stamp_n[id];
            for (int i = 0; i < n && i < 96; i++)
                printf("RSTAMP lvl %d pass %d kind %d val %d dcycles %llu\\n", (int)blockIdx.y, q, (int)(replay::g_stamp[id][0][i] >> 32), (int)(unsigned)replay::g_stamp[id][0][i],
                       i ? replay::g_stamp[id][1][i] - replay::g_stamp[id][1][i - 1] : 0ull);
        }
}""", 1)
open(g, "w").write(k)
PY
make -C "$tmp/visual-slam_amd/csrc" -j8 2>&1 | grep -E "error" -A3 | head
mkdir -p "$root/visual-slam_amd/variants"
cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libreplay_stamps.so"
rm -rf "$tmp"; echo built replay stamps

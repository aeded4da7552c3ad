// track_kernels.hip -- the two match filters of the reference's per-frame tracking step, fused behind the matcher
// (reference src/orbslam2/tracker.py:214-254):
//   matcher.py:109-142  filter_matches_by_geometric_distance(kp_prev, kp_cur, matches, 0.02, (h, w)):
//                       keep  hypot(pt_cur - pt_prev) <= ((w + h) / 2) * 0.02          (Python floats = IEEE double)
//   matcher.py:144-169  filter_matches_by_distance(matches): stable sort by distance, keep distance < 2 * np.median
// The survivors, IN THE REFERENCE'S ORDER (ascending distance, ties in query order), are what Tracker feeds to
// cv2.findEssentialMat(..., RANSAC, 0.999, 1.0) and cv2.recoverPose (tracker.py:242-249): k_track_select writes that list
// and the two-view kernels (twoview_kernels.hip) consume it instead of the ratio-test flags.
//
// One workgroup per frame pair.  Distances are integers 0..256 and a pair has at most 8192 matches, so the sort key
// (distance << 16 | query index) is unique and the stable order is its plain order: every survivor counts the smaller keys
// (LDS broadcast reads; a few hundred survivors per pair) and scatters itself to that rank.  np.median of the sorted list is
// (d[(n-1)/2] + d[n/2]) / 2, so "distance < 2 * median" is the integer test  distance < d[(n-1)/2] + d[n/2]  and, the
// list being sorted, the kept matches are a prefix.
#include "common.h"

#define TS_BLOCK 256

__global__ __launch_bounds__(TS_BLOCK) void k_track_select(const mo_keypoint* __restrict__ kps, const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ qf, const int32_t* __restrict__ tf,
                                                           const int32_t* __restrict__ midx, const int32_t* __restrict__ mdist,
                                                           const uint8_t* __restrict__ mpass, int cap, double max_disp,
                                                           int32_t* __restrict__ sel /* [pairs][cap][2] */,
                                                           int32_t* __restrict__ sel_dist /* [pairs][cap] or null */,
                                                           int32_t* __restrict__ sel_n /* [pairs] */) {
    extern __shared__ uint32_t s_key[];  // [cap] keys in query order, then [cap] keys by rank
    __shared__ int s_w[TS_BLOCK / 64];
    __shared__ int s_base;
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fq = qf ? qf[pair] : pair, ft = tf ? tf[pair] : pair + 1;
    const int nq = min(counts[fq], cap);
    const mo_keypoint* k1 = kps + (size_t)fq * cap;
    const mo_keypoint* k2 = kps + (size_t)ft * cap;
    const int32_t* idx = midx + (size_t)pair * cap * 2;
    const int32_t* dst = mdist + (size_t)pair * cap * 2;
    const uint8_t* pass = mpass + (size_t)pair * cap;
    uint32_t* s_sorted = s_key + cap;
    if (tid == 0) s_base = 0;
    __syncthreads();
    // 1. ratio-test survivors within the displacement limit, compacted in query order
    for (int b0 = 0; b0 < nq; b0 += TS_BLOCK) {
        const int i = b0 + tid;
        bool ok = i < nq && pass[i];
        int d = 0;
        if (ok) {
            const int j = idx[2 * i];
            const double dx = (double)k2[j].x - (double)k1[i].x, dy = (double)k2[j].y - (double)k1[i].y;
            ok = sqrt(dx * dx + dy * dy) <= max_disp;
            d = dst[2 * i];
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) s_w[wv] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int k = 0; k < wv; k++) off += s_w[k];
        if (ok) s_key[off + __popcll(bal & ((1ull << lane) - 1ull))] = ((uint32_t)d << 16) | (uint32_t)i;
        __syncthreads();
        if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    const int n = s_base;
    // 2. rank of every survivor = number of smaller keys (keys are unique) -> sorted order
    for (int e = tid; e < n; e += TS_BLOCK) {
        const uint32_t key = s_key[e];
        int rank = 0;
        for (int j = 0; j < n; j++) rank += s_key[j] < key ? 1 : 0;  // wave-uniform address: one broadcast read per trip
        s_sorted[rank] = key;
    }
    __syncthreads();
    // 3. 2 * median = d[(n-1)/2] + d[n/2]; kept = the prefix with distance below it
    int n_keep = 0;
    if (n > 0) {
        const int thr = (int)(s_sorted[(n - 1) >> 1] >> 16) + (int)(s_sorted[n >> 1] >> 16);
        int cnt = 0;
        for (int e = tid; e < n; e += TS_BLOCK) cnt += (int)(s_sorted[e] >> 16) < thr ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) s_w[wv] = cnt;
        __syncthreads();
        n_keep = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
    int32_t* out = sel + (size_t)pair * cap * 2;
    for (int e = tid; e < n_keep; e += TS_BLOCK) {
        const uint32_t key = s_sorted[e];
        const int i = key & 0xFFFF;
        out[2 * e] = i;
        out[2 * e + 1] = idx[2 * i];
        if (sel_dist) sel_dist[(size_t)pair * cap + e] = (int)(key >> 16);
    }
    if (tid == 0) sel_n[pair] = n_keep;
}

int track_select_launch(mo_ctx* c, const mo_keypoint* d_kps, const int32_t* d_counts, const int32_t* d_qf, const int32_t* d_tf,
                        const int32_t* d_midx, const int32_t* d_mdist, const uint8_t* d_mpass, int cap, int n_pairs, int w, int h,
                        double disp_frac, int32_t* d_sel, int32_t* d_sel_dist, int32_t* d_sel_n) {
    if (n_pairs <= 0) return MO_OK;
    if (cap > 8192) return mo_fail(c, MO_ERR_UNSUPPORTED, "tracking filters support at most 8192 keypoints per frame (two key arrays in 64 KB of LDS)");
    const double max_disp = ((double)(w + h) / 2.0) * disp_frac;  // matcher.py:128 ((width + height) / 2.0) * threshold_percent
    hipLaunchKernelGGL(k_track_select, dim3(n_pairs), dim3(TS_BLOCK), (size_t)cap * 2 * sizeof(uint32_t), c->stream, d_kps, d_counts,
                       d_qf, d_tf, d_midx, d_mdist, d_mpass, cap, max_disp, d_sel, d_sel_dist, d_sel_n);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// track_kernels.hip -- the two match filters of the reference's per-frame tracking step, fused behind the matcher
// (reference src/orbslam2/tracker.py:214-254):
//   matcher.py:109-142  filter_matches_by_geometric_distance(kp_prev, kp_cur, matches, 0.02, (h, w)):
//                       keep  hypot(pt_cur - pt_prev) <= ((w + h) / 2) * 0.02          (Python floats = IEEE double)
//   matcher.py:144-169  filter_matches_by_distance(matches): stable sort by distance, keep distance < 2 * np.median
// The survivors, IN THE REFERENCE'S ORDER (ascending distance, ties in query order), are what Tracker feeds to
// cv2.findEssentialMat(..., RANSAC, 0.999, 1.0) and cv2.recoverPose (tracker.py:242-249): k_track_select writes that list
// and the two-view kernels (twoview_kernels.hip) consume it instead of the ratio-test flags.
//
// One workgroup per frame pair.  Distances are integers 0..256, so the stable sort is a counting sort: the ratio-test survivors inside
// the displacement gate are compacted in query order (key = distance << 23 | query index) while a 257-bin histogram is taken; its
// exclusive prefix gives every distance its first rank, the median pair d[(n-1)/2], d[n/2] (np.median of the sorted list is their mean,
// so "distance < 2 * median" is the integer test distance < d[(n-1)/2] + d[n/2]) and - the list being sorted - the number of kept matches,
// prefix[threshold].  Only the kept matches are then placed: wavefront w walks the compacted list 64 keys at a time and scatters those of
// ITS distances (d & 3 == w) behind the running count of their bin - ties keep the query order, which is the order Python's stable
// sorted() leaves them in.  Any number of keypoints per frame: the two key arrays live in LDS up to 6 000 records, in an HBM scratch
// slot beyond (rounds 2 - 3 ranked every survivor against all others out of LDS: at most 8 192 keypoints, 23 us for one pair).
#include "common.h"

#define TS_BLOCK 256
#define TS_KEY_BITS 23
#define TS_UNROLL 4
#define TS_LDS_CAP 6000   // records whose two key arrays fit the default 48 KB of dynamic LDS

__device__ __forceinline__ void ts_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ __launch_bounds__(TS_BLOCK) void k_track_select(const mo_keypoint* __restrict__ kps, const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ qf, const int32_t* __restrict__ tf,
                                                           const int32_t* __restrict__ midx, const int32_t* __restrict__ mdist,
                                                           const uint8_t* __restrict__ mpass, int cap, double max_disp,
                                                           uint32_t* __restrict__ gkeys /* [pairs][2 * cap] or null: keys in LDS */,
                                                           int32_t* __restrict__ sel /* [pairs][cap][2] */,
                                                           int32_t* __restrict__ sel_dist /* [pairs][cap] or null */,
                                                           int32_t* __restrict__ sel_n /* [pairs] */) {
    extern __shared__ uint32_t s_dyn[];  // [cap] keys in query order, then [cap] kept keys by rank (when gkeys is null)
    __shared__ int s_hist[320];          // survivors per distance (257 bins, zero-padded), then their exclusive prefix
    __shared__ int s_bin[260];           // running placement count per distance
    __shared__ int s_w4[TS_UNROLL][TS_BLOCK / 64];
    __shared__ int s_med[2];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fq = qf ? qf[pair] : pair, ft = tf ? tf[pair] : pair + 1;
    const int nq = min(counts[fq], cap);
    const mo_keypoint* k1 = kps + (size_t)fq * cap;
    const mo_keypoint* k2 = kps + (size_t)ft * cap;
    const int32_t* idx = midx + (size_t)pair * cap * 2;
    const int32_t* dst = mdist + (size_t)pair * cap * 2;
    const uint8_t* pass = mpass + (size_t)pair * cap;
    uint32_t* keys = gkeys ? gkeys + (size_t)pair * 2 * cap : s_dyn;
    uint32_t* sorted = keys + cap;
    for (int b = tid; b < 320; b += TS_BLOCK) s_hist[b] = 0;
    __syncthreads();
    // 1. ratio-test survivors within the displacement limit, compacted in query order; histogram of their distances.  TS_UNROLL x 256
    //    queries per trip: the (pass, knn index, distance, both keypoints) loads of all of a thread's queries are in flight together -
    //    one pair of 2000 queries waited for eight dependent load chains in a row before (22 us for this kernel alone).
    int n_run = 0;
    for (int b0 = 0; b0 < nq; b0 += TS_BLOCK * TS_UNROLL) {
        bool ok[TS_UNROLL];
        int d[TS_UNROLL], j[TS_UNROLL];
        float x1[TS_UNROLL], y1[TS_UNROLL];
#pragma unroll
        for (int u = 0; u < TS_UNROLL; u++) {
            const int i = b0 + u * TS_BLOCK + tid;
            ok[u] = i < nq && pass[i];
            j[u] = i < nq ? idx[2 * i] : 0;
            d[u] = i < nq ? dst[2 * i] : 0;
            x1[u] = i < nq ? k1[i].x : 0.f;
            y1[u] = i < nq ? k1[i].y : 0.f;
        }
#pragma unroll
        for (int u = 0; u < TS_UNROLL; u++) {
            if (ok[u]) {
                const mo_keypoint* kb = k2 + max(j[u], 0);
                const double dx = (double)kb->x - (double)x1[u], dy = (double)kb->y - (double)y1[u];
                ok[u] = sqrt(dx * dx + dy * dy) <= max_disp;
                d[u] = min(max(d[u], 0), 256);
            }
        }
        unsigned long long bal[TS_UNROLL];
#pragma unroll
        for (int u = 0; u < TS_UNROLL; u++) {
            bal[u] = __ballot(ok[u]);
            if (lane == 0) s_w4[u][wv] = __popcll(bal[u]);
        }
        __syncthreads();
        int off = n_run;  // survivors of the earlier trips, then of the earlier sub-rows and wavefronts of this one (query order)
#pragma unroll
        for (int u = 0; u < TS_UNROLL; u++) {
            int mine = off;
            for (int k = 0; k < TS_BLOCK / 64; k++) { if (k < wv) mine += s_w4[u][k]; off += s_w4[u][k]; }
            if (ok[u]) {
                keys[mine + __popcll(bal[u] & ((1ull << lane) - 1ull))] = ((uint32_t)d[u] << TS_KEY_BITS) | (uint32_t)(b0 + u * TS_BLOCK + tid);
                atomicAdd(&s_hist[d[u]], 1);
            }
        }
        n_run = off;
        __syncthreads();  // (s_w4 is rewritten by the next trip)
    }
    const int n = n_run;
    // 2. exclusive prefix of the histogram (one wavefront, five bins per lane)
    if (wv == 0) {
        int v[5], sum = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) { v[k] = s_hist[lane * 5 + k]; sum += v[k]; }
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
        int run = inc - sum;
#pragma unroll
        for (int k = 0; k < 5; k++) { s_hist[lane * 5 + k] = run; run += v[k]; }
    }
    __syncthreads();
    // 3. 2 * median = d[(n-1)/2] + d[n/2]: the bins these two ranks fall into; kept = the distances below that sum = a prefix
    int n_keep = 0, thr = 0;
    if (n > 0) {
        const int r0 = (n - 1) >> 1, r1 = n >> 1;
        for (int b = tid; b < 257; b += TS_BLOCK) {
            const int lo = s_hist[b], hi = s_hist[b + 1];   // (bin 257.. hold n)
            if (lo <= r0 && r0 < hi) s_med[0] = b;
            if (lo <= r1 && r1 < hi) s_med[1] = b;
        }
        __syncthreads();
        thr = s_med[0] + s_med[1];
        n_keep = s_hist[min(thr, 257)];
    }
    for (int b = tid; b < 257; b += TS_BLOCK) s_bin[b] = s_hist[b];
    __syncthreads();
    // 4. stable placement of the kept matches: wavefront w owns the distances with d & 3 == w and walks the compacted list 64 keys at a
    //    time.  A lane finds the lanes of its chunk that hold the SAME distance with nine ballots (one per bit of d: no loop over the
    //    distinct distances, which cost an LDS round trip each - 11 us of this kernel for one pair); its rank among them plus the bin's
    //    running count is its slot, and the lowest lane of every group advances that count.
    if (n_keep > 0) {
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int c0 = 0; c0 < n; c0 += 64) {  // (workgroup-uniform trip count)
            const int e = c0 + lane;
            const uint32_t key = e < n ? keys[e] : 0xFFFFFFFFu;
            const int d = (int)(key >> TS_KEY_BITS);
            const bool mine = e < n && (d & 3) == wv && d < thr;
            unsigned long long same = __ballot(mine);
#pragma unroll
            for (int b = 2; b < 9; b++) {  // (bits 0 - 1 are the wavefront's: equal for all of its keys)
                const unsigned long long hb = __ballot((d >> b) & 1);
                same &= ((d >> b) & 1) ? hb : ~hb;
            }
            if (mine) {
                const int base = s_bin[d];
                sorted[base + __popcll(same & lt)] = key;
                if ((same & lt) == 0) s_bin[d] = base + __popcll(same);  // (the group's lowest lane; the other lanes read the old count above)
            }
            ts_wave_sync();
        }
    }
    __syncthreads();
    int32_t* out = sel + (size_t)pair * cap * 2;
    for (int e = tid; e < n_keep; e += TS_BLOCK) {
        const uint32_t key = sorted[e];
        const int i = key & ((1u << TS_KEY_BITS) - 1u);
        out[2 * e] = i;
        out[2 * e + 1] = idx[2 * i];
        if (sel_dist) sel_dist[(size_t)pair * cap + e] = (int)(key >> TS_KEY_BITS);
    }
    if (tid == 0) sel_n[pair] = n_keep;
}

int track_select_launch(mo_ctx* c, const mo_keypoint* d_kps, const int32_t* d_counts, const int32_t* d_qf, const int32_t* d_tf,
                        const int32_t* d_midx, const int32_t* d_mdist, const uint8_t* d_mpass, int cap, int n_pairs, int w, int h,
                        double disp_frac, int32_t* d_sel, int32_t* d_sel_dist, int32_t* d_sel_n) {
    if (n_pairs <= 0) return MO_OK;
    if (cap >= (1 << TS_KEY_BITS)) return mo_fail(c, MO_ERR_UNSUPPORTED, "tracking filters support at most 8 388 607 keypoints per frame");
    const double max_disp = ((double)(w + h) / 2.0) * disp_frac;  // matcher.py:128 ((width + height) / 2.0) * threshold_percent
    uint32_t* gkeys = nullptr;
    size_t lds = (size_t)cap * 2 * sizeof(uint32_t);
    if (cap > TS_LDS_CAP) {  // frames beyond 6 000 keypoints: the key arrays in an HBM scratch slot per pair
        int rc = mo_reserve(c, c->d_track_keys, c->track_keys_bytes, (size_t)n_pairs * cap * 2 * sizeof(uint32_t));
        if (rc) return rc;
        gkeys = c->d_track_keys;
        lds = 0;
    }
    hipLaunchKernelGGL(k_track_select, dim3(n_pairs), dim3(TS_BLOCK), lds, c->stream, d_kps, d_counts, d_qf, d_tf, d_midx, d_mdist, d_mpass, cap,
                       max_disp, gkeys, d_sel, d_sel_dist, d_sel_n);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

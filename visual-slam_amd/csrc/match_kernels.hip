// match_kernels.hip -- brute-force Hamming 2-NN + fused Lowe ratio test (gfx950).
// Replaces cv2.BFMatcher(NORM_HAMMING).knnMatch(d1, d2, k=2) and the Python ratio loop of the reference
// (src/orbslam2/matcher.py:70,73-81).
//
// Integer bit work, no MFMA: each lane owns one query descriptor in 8 VGPRs; the train descriptors are
// walked by every wave in the same order, so their 32 bytes arrive through wave-uniform (scalar) loads and
// the inner loop is 8 x (v_xor_b32 + v_bcnt_u32_b32) + a packed (distance << 20 | trainIdx) key folded
// into the running best / second-best with v_med3_u32 + v_min_u32.  Smallest key == smallest distance,
// ties towards the lower trainIdx, exactly batch_distance.cpp's strict '<' insertion order.
#include <climits>

#include "common.h"

#define MQ_PER_BLOCK 256
#define KEY_NONE 0xFFFFFFFFu

__global__ __launch_bounds__(MQ_PER_BLOCK) void k_match(const uint8_t* __restrict__ qbase, const uint8_t* __restrict__ tbase,
                                                        size_t q_stride, size_t t_stride,
                                                        const int32_t* __restrict__ counts, const int32_t* __restrict__ qf,
                                                        const int32_t* __restrict__ tf, int nq_fixed, int nt_fixed,
                                                        int out_stride, double ratio, int32_t* __restrict__ oidx,
                                                        int32_t* __restrict__ odist, uint8_t* __restrict__ opass) {
    const int pair = blockIdx.y;
    const int qfr = qf ? qf[pair] : pair, tfr = tf ? tf[pair] : pair;
    const int nq = counts ? min(counts[qfr], out_stride) : nq_fixed;
    const int nt = counts ? min(counts[tfr], out_stride) : nt_fixed;
    const int qi = blockIdx.x * MQ_PER_BLOCK + threadIdx.x;
    if (blockIdx.x * MQ_PER_BLOCK >= nq) return;
    const uint32_t* q = (const uint32_t*)(qbase + (size_t)qfr * q_stride);
    const uint32_t* t = (const uint32_t*)(tbase + (size_t)tfr * t_stride);
    uint32_t a[8];
    const int qclamped = min(qi, nq - 1);
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = q[(size_t)qclamped * 8 + k];
    uint32_t k0 = KEY_NONE, k1 = KEY_NONE;
    for (int j = 0; j < nt; j++) {
        const uint32_t* b = t + (size_t)j * 8;  // wave-uniform address -> scalar loads
        uint32_t d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) d += __popc(a[k] ^ b[k]);
        uint32_t key = (d << 20) | (uint32_t)j;
        // new second-best = median(k0, k1, key) given k0 <= k1; new best = min(k0, key)
        uint32_t hi = max(k0, key);
        k1 = min(k1, hi);
        k0 = min(k0, key);
    }
    if (qi >= nq) return;
    int i0 = k0 == KEY_NONE ? -1 : (int)(k0 & 0xFFFFFu), i1 = k1 == KEY_NONE ? -1 : (int)(k1 & 0xFFFFFu);
    int d0 = k0 == KEY_NONE ? INT_MAX : (int)(k0 >> 20), d1 = k1 == KEY_NONE ? INT_MAX : (int)(k1 >> 20);
    size_t o = (size_t)pair * out_stride + qi;
    oidx[2 * o] = i0; oidx[2 * o + 1] = i1;
    odist[2 * o] = d0; odist[2 * o + 1] = d1;
    // matcher.py:73-81: len(match) >= 2 -> m.distance < ratio * n.distance (Python floats = IEEE double);
    // one neighbour only -> kept; ratio <= 0 encodes ratio_test=False
    uint8_t pass;
    if (i0 < 0) pass = 0;
    else if (i1 < 0 || !(ratio > 0.0)) pass = 1;
    else pass = ((double)d0 < ratio * (double)d1) ? 1 : 0;
    opass[o] = pass;
}

int match_launch_pairs(mo_ctx* c, const uint8_t* d_q, const uint8_t* d_t, size_t q_stride, size_t t_stride,
                       const int32_t* d_counts, const int32_t* d_qf, const int32_t* d_tf, int nq_fixed, int nt_fixed,
                       int n_pairs, int out_stride, double ratio, int32_t* d_idx, int32_t* d_dist, uint8_t* d_pass) {
    if (n_pairs <= 0) return MO_OK;
    int nq_max = d_counts ? out_stride : nq_fixed;
    if (nq_max <= 0) return MO_OK;
    if ((d_counts ? out_stride : nt_fixed) >= (1 << 20)) return mo_fail(c, MO_ERR_UNSUPPORTED, "more than 2^20-1 train descriptors");
    dim3 grid((nq_max + MQ_PER_BLOCK - 1) / MQ_PER_BLOCK, n_pairs);
    hipLaunchKernelGGL(k_match, grid, dim3(MQ_PER_BLOCK), 0, c->stream, d_q, d_t, q_stride, t_stride, d_counts, d_qf, d_tf,
                       nq_fixed, nt_fixed, out_stride, ratio, d_idx, d_dist, d_pass);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

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

#define MQ_THREADS 256
#define MQ_Q 2            // queries per lane (4 leaves too few workgroups at 2000 queries per pair: measured slower)
#define MQ_PER_BLOCK (MQ_THREADS * MQ_Q)
#define KEY_NONE 0xFFFFFFFFu

__device__ __forceinline__ void emit_match(uint32_t k0, uint32_t k1, double ratio, size_t o, int32_t* oidx, int32_t* odist,
                                           uint8_t* opass) {
    int i0 = k0 == KEY_NONE ? -1 : (int)(k0 & 0xFFFFFu), i1 = k1 == KEY_NONE ? -1 : (int)(k1 & 0xFFFFFu);
    int d0 = k0 == KEY_NONE ? INT_MAX : (int)(k0 >> 20), d1 = k1 == KEY_NONE ? INT_MAX : (int)(k1 >> 20);
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

__global__ __launch_bounds__(MQ_THREADS) void k_match(const uint8_t* __restrict__ qbase, const uint8_t* __restrict__ tbase,
                                                      size_t q_stride, size_t t_stride, const int32_t* __restrict__ counts,
                                                      const int32_t* __restrict__ qf, const int32_t* __restrict__ tf,
                                                      int nq_fixed, int nt_fixed, int out_stride, double ratio,
                                                      int32_t* __restrict__ oidx, int32_t* __restrict__ odist,
                                                      uint8_t* __restrict__ opass) {
    const int pair = blockIdx.y;
    const int qfr = qf ? qf[pair] : pair, tfr = tf ? tf[pair] : pair;
    const int nq = counts ? min(counts[qfr], out_stride) : nq_fixed;
    const int nt = counts ? min(counts[tfr], out_stride) : nt_fixed;
    if (blockIdx.x * MQ_PER_BLOCK >= nq) return;
    const uint32_t* q = (const uint32_t*)(qbase + (size_t)qfr * q_stride);
    const uint32_t* t = (const uint32_t*)(tbase + (size_t)tfr * t_stride);
    uint32_t a[MQ_Q][8], k0[MQ_Q], k1[MQ_Q];
#pragma unroll
    for (int m = 0; m < MQ_Q; m++) {
        const int qi = min(blockIdx.x * MQ_PER_BLOCK + m * MQ_THREADS + (int)threadIdx.x, nq - 1);
#pragma unroll
        for (int k = 0; k < 8; k++) a[m][k] = q[(size_t)qi * 8 + k];
        k0[m] = KEY_NONE; k1[m] = KEY_NONE;
    }
#pragma unroll 4
    for (int j = 0; j < nt; j++) {  // 4 train descriptors (scalar loads) in flight per trip
        const uint32_t* b = t + (size_t)j * 8;  // wave-uniform address -> scalar loads
        uint32_t d[MQ_Q];
#pragma unroll
        for (int m = 0; m < MQ_Q; m++) d[m] = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t bk = b[k];
#pragma unroll
            for (int m = 0; m < MQ_Q; m++) d[m] += __popc(a[m][k] ^ bk);
        }
#pragma unroll
        for (int m = 0; m < MQ_Q; m++) {
            const uint32_t key = (d[m] << 20) | (uint32_t)j;
            // new second-best = min(k1, max(k0, key)) given k0 <= k1; new best = min(k0, key)
            k1[m] = min(k1[m], max(k0[m], key));
            k0[m] = min(k0[m], key);
        }
    }
#pragma unroll
    for (int m = 0; m < MQ_Q; m++) {
        const int qi = blockIdx.x * MQ_PER_BLOCK + m * MQ_THREADS + threadIdx.x;
        if (qi < nq) emit_match(k0[m], k1[m], ratio, (size_t)pair * out_stride + qi, oidx, odist, opass);
    }
}

int match_launch_pairs(mo_ctx* c, const uint8_t* d_q, const uint8_t* d_t, size_t q_stride, size_t t_stride,
                       const int32_t* d_counts, const int32_t* d_qf, const int32_t* d_tf, int nq_fixed, int nt_fixed,
                       int n_pairs, int out_stride, double ratio, int32_t* d_idx, int32_t* d_dist, uint8_t* d_pass) {
    if (n_pairs <= 0) return MO_OK;
    int nq_max = d_counts ? out_stride : nq_fixed;
    if (nq_max <= 0) return MO_OK;
    if ((d_counts ? out_stride : nt_fixed) >= (1 << 20)) return mo_fail(c, MO_ERR_UNSUPPORTED, "more than 2^20-1 train descriptors");
    dim3 grid((nq_max + MQ_PER_BLOCK - 1) / MQ_PER_BLOCK, n_pairs);
    hipLaunchKernelGGL(k_match, grid, dim3(MQ_THREADS), 0, c->stream, d_q, d_t, q_stride, t_stride, d_counts, d_qf, d_tf,
                       nq_fixed, nt_fixed, out_stride, ratio, d_idx, d_dist, d_pass);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

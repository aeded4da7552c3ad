// match_kernels.hip -- brute-force Hamming 2-NN + fused Lowe ratio test (gfx950).
// Replaces cv2.BFMatcher(NORM_HAMMING).knnMatch(d1, d2, k=2) and the Python ratio loop of the reference
// (src/orbslam2/matcher.py:70,73-81).
//
// Default path (north_star: integer bit work, no MFMA): each lane owns one query descriptor in 8 VGPRs; the train descriptors are
// walked by every wave in the same order, so their 32 bytes arrive through wave-uniform (scalar) loads and
// the inner loop is 8 x (v_xor_b32 + v_bcnt_u32_b32) + a packed (distance << 20 | trainIdx) key folded
// into the running best / second-best with v_med3_u32 + v_min_u32.  Smallest key == smallest distance,
// ties towards the lower trainIdx, exactly batch_distance.cpp's strict '<' insertion order.
#include <algorithm>
#include <climits>

#include "common.h"

#define KEY_NONE 0xFFFFFFFFu

// one output row: i0/i1 = trainIdx of the best / second-best neighbour (-1: none), d0/d1 their distances
__device__ __forceinline__ void emit_match(int i0, int d0, int i1, int d1, double ratio, size_t o, int32_t* oidx, int32_t* odist,
                                           uint8_t* opass) {
    oidx[2 * o] = i0; oidx[2 * o + 1] = i1;
    odist[2 * o] = d0; odist[2 * o + 1] = d1;
    // matcher.py:73-81: len(match) >= 2 -> m.distance < ratio * n.distance (Python floats = IEEE double);
    // one neighbour only -> kept; a NEGATIVE ratio encodes ratio_test=False (0.0 is a valid threshold: like the reference's
    // strict '<' it lets nothing with two neighbours through)
    uint8_t pass;
    if (i0 < 0) pass = 0;
    else if (i1 < 0 || ratio < 0.0) pass = 1;
    else pass = ((double)d0 < ratio * (double)d1) ? 1 : 0;
    opass[o] = pass;
}

__device__ __forceinline__ void emit_match_key(uint32_t k0, uint32_t k1, double ratio, size_t o, int32_t* oidx, int32_t* odist,
                                               uint8_t* opass) {
    int i0 = k0 == KEY_NONE ? -1 : (int)(k0 & 0xFFFFFu), i1 = k1 == KEY_NONE ? -1 : (int)(k1 & 0xFFFFFu);
    int d0 = k0 == KEY_NONE ? INT_MAX : (int)(k0 >> 20), d1 = k1 == KEY_NONE ? INT_MAX : (int)(k1 >> 20);
    emit_match(i0, d0, i1, d1, ratio, o, oidx, odist, opass);
}

__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {  // popcount(x) + acc in one instruction, kept as a chain
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {  // the compiler only forms it for constant clamps
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Default path since round 2: the same XOR + popcount + packed-key fold, but the train descriptors are staged through LDS.
// A workgroup copies tiles of ML_TILE train descriptors (4 KB, one 16-byte load per thread, register-staged and
// double-buffered so the next tile's global load is in flight during the current tile's compute; one barrier per tile)
// and every wavefront reads them back with wave-uniform (broadcast) ds_read_b128: two LDS instructions per train descriptor
// next to 38 vector instructions, in-order returns (counted lgkmcnt waits, reads issued several descriptors ahead).
// (Round 1's kernel took them through scalar loads instead: those return out of order, every trip waited for all of them
// (lgkmcnt(0)) with nothing in flight, and it sat at 0.42 of the vector issue rate (profiles/r01_sq_counters); it lived on as the path
// for descriptor arrays that are not 16-byte aligned until round 4, which asks for the alignment instead.)
#define ML_THREADS 256
#define ML_Q 2
#define ML_PER_BLOCK (ML_THREADS * ML_Q)
#define ML_SPLIT_BLOCKS 32  // launches with at most this many workgroups slice the train set ...
#define ML_SPLIT_MAX 32     // ... into at most this many slices
#define ML_TILE 128  // train descriptors per tile: 256 threads x 16 B

__global__ __launch_bounds__(ML_THREADS) void k_match_lds(const uint8_t* __restrict__ qbase, const uint8_t* __restrict__ tbase,
                                                          size_t q_stride, size_t t_stride, const int32_t* __restrict__ counts,
                                                          const int32_t* __restrict__ qf, const int32_t* __restrict__ tf,
                                                          int nq_fixed, int nt_fixed, int out_stride, double ratio,
                                                          int32_t* __restrict__ oidx, int32_t* __restrict__ odist,
                                                          uint8_t* __restrict__ opass, uint2* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) uint4 s_t[2][ML_TILE * 2 + 4];  // + one look-ahead group past the last descriptor
    const int pair = blockIdx.y, tid = threadIdx.x;
    const int qfr = qf ? qf[pair] : pair, tfr = tf ? tf[pair] : pair;
    const int nq = counts ? min(counts[qfr], out_stride) : nq_fixed;
    const int nt = counts ? min(counts[tfr], out_stride) : nt_fixed;
    if (blockIdx.x * ML_PER_BLOCK >= nq) return;  // uniform over the workgroup
    const uint4* q = (const uint4*)(qbase + (size_t)qfr * q_stride);
    const uint4* t = (const uint4*)(tbase + (size_t)tfr * t_stride);
    uint32_t a[ML_Q][8], k0[ML_Q], k1[ML_Q];
#pragma unroll
    for (int m = 0; m < ML_Q; m++) {
        const int qi = min(blockIdx.x * ML_PER_BLOCK + m * ML_THREADS + tid, nq - 1);
        const uint4 lo = q[(size_t)qi * 2], hi = q[(size_t)qi * 2 + 1];
        a[m][0] = lo.x; a[m][1] = lo.y; a[m][2] = lo.z; a[m][3] = lo.w;
        a[m][4] = hi.x; a[m][5] = hi.y; a[m][6] = hi.z; a[m][7] = hi.w;
        k0[m] = KEY_NONE; k1[m] = KEY_NONE;
    }
    // gridDim.z > 1 (few pairs, see match_launch_pairs): this workgroup folds only its slice of the train tiles and leaves its
    // two best keys in `part`; k_match_merge folds the slices - keys carry the global train index, so the result is the same
    const int ntile_all = (nt + ML_TILE - 1) / ML_TILE, per_z = (ntile_all + gridDim.z - 1) / gridDim.z;
    const int tile_lo = blockIdx.z * per_z, ntile = min(ntile_all, tile_lo + per_z);
    const int n16 = nt * 2;  // 16-byte pieces of the train set
    if (tile_lo < ntile) {
        const int i = tile_lo * ML_TILE * 2 + tid;
        s_t[tile_lo & 1][tid] = i < n16 ? t[i] : make_uint4(0, 0, 0, 0);
    }
    // the query registers are complete HERE: otherwise their vmcnt wait lands inside the tile loop and, counting in issue
    // order, would also wait for the next tile's prefetch on every trip
#pragma unroll
    for (int m = 0; m < ML_Q; m++)
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("" : "+v"(a[m][k]));
    __syncthreads();
    for (int tile = tile_lo; tile < ntile; tile++) {
        const bool more = tile + 1 < ntile;
        uint4 nxt = make_uint4(0, 0, 0, 0);
        if (more) {
            const int i = (tile + 1) * ML_TILE * 2 + tid;
            if (i < n16) nxt = t[i];
        }
        const uint4* s = s_t[tile & 1];
        const int j0 = tile * ML_TILE, n = min(ML_TILE, nt - j0);
        auto fold = [&](const uint4& b0, const uint4& b1, int j) {  // j: wave-uniform train index
            const uint32_t b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            uint32_t d[ML_Q];
#pragma unroll
            for (int m = 0; m < ML_Q; m++) d[m] = 0;
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int m = 0; m < ML_Q; m++) d[m] = bcnt_acc(a[m][k] ^ b[k], d[m]);
#pragma unroll
            for (int m = 0; m < ML_Q; m++) {
                const uint32_t key = (d[m] << 20) | (uint32_t)j;
                k1[m] = med3_u32(k0[m], k1[m], key);  // k0 <= k1: new second-best = median, new best = minimum
                k0[m] = min(k0[m], key);
            }
        };
        // groups of two descriptors with one group of look-ahead: the broadcast reads of group g+1 are in flight while
        // group g is folded (reads past the last descriptor stay inside the padded buffer and are never folded).  Two groups
        // per trip in two register sets that swap roles: rotating one set cost four 64-bit moves per group, 12 % of the loop.
        uint4 c0 = s[0], c1 = s[1], c2 = s[2], c3 = s[3];
        int j = 0;
        for (; j + 4 <= n; j += 4) {
            const uint4 e0 = s[2 * j + 4], e1 = s[2 * j + 5], e2 = s[2 * j + 6], e3 = s[2 * j + 7];
            fold(c0, c1, j0 + j);
            fold(c2, c3, j0 + j + 1);
            c0 = s[2 * j + 8]; c1 = s[2 * j + 9]; c2 = s[2 * j + 10]; c3 = s[2 * j + 11];
            fold(e0, e1, j0 + j + 2);
            fold(e2, e3, j0 + j + 3);
        }
        if (j + 2 <= n) {
            const uint4 e0 = s[2 * j + 4], e1 = s[2 * j + 5];
            fold(c0, c1, j0 + j);
            fold(c2, c3, j0 + j + 1);
            c0 = e0; c1 = e1;
            j += 2;
        }
        if (j < n) fold(c0, c1, j0 + j);
        if (more) s_t[(tile + 1) & 1][tid] = nxt;
        __syncthreads();  // the buffer written here was last read before the previous barrier
    }
#pragma unroll
    for (int m = 0; m < ML_Q; m++) {
        const int qi = blockIdx.x * ML_PER_BLOCK + m * ML_THREADS + tid;
        if (qi < nq && part) part[((size_t)pair * gridDim.z + blockIdx.z) * out_stride + qi] = make_uint2(k0[m], k1[m]);
        if (qi < nq && !part) emit_match_key(k0[m], k1[m], ratio, (size_t)pair * out_stride + qi, oidx, odist, opass);
    }
}

// Folds the per-slice key pairs of a split k_match_lds launch (one thread per query).
__global__ __launch_bounds__(256) void k_match_merge(const uint2* __restrict__ part, int n_split, const int32_t* __restrict__ counts,
                                                     const int32_t* __restrict__ qf, int nq_fixed, int out_stride, double ratio,
                                                     int32_t* __restrict__ oidx, int32_t* __restrict__ odist,
                                                     uint8_t* __restrict__ opass) {
    const int pair = blockIdx.y, qi = blockIdx.x * 256 + threadIdx.x;
    const int nq = counts ? min(counts[qf ? qf[pair] : pair], out_stride) : nq_fixed;
    if (qi >= nq) return;
    uint32_t k0 = KEY_NONE, k1 = KEY_NONE;
    for (int z = 0; z < n_split; z++) {
        const uint2 k = part[((size_t)pair * n_split + z) * out_stride + qi];
        k1 = med3_u32(k0, k1, k.x); k0 = min(k0, k.x);
        k1 = med3_u32(k0, k1, k.y); k0 = min(k0, k.y);
    }
    emit_match_key(k0, k1, ratio, (size_t)pair * out_stride + qi, oidx, odist, opass);
}

// ---------------------------------------------------------------------------------------------------------------
// Opt-in matrix-core path (environment VSLAM_AMD_MATCHER=mfma when the context is created; same results bit for bit).
// The 2000 x 2000 x 256-bit distance table of a pair is a GEMM over +-1 vectors: with every
// descriptor bit expanded to the int8 value +127 (set) or -127 (clear), a.b = 16129 * (256 - 2 * hamming).  The
// accumulator of v_mfma_i32_32x32x32_i8 is started at C = 31 - (train row within the tile), so each of its 16 registers
// ends as  M - (32258 * hamming + row)  with M = 16129 * 256 + 31: ONE signed integer ordered exactly like
// (distance, trainIdx) reversed.  The epilogue therefore only keeps the two largest of a lane's 16 registers
// (v_max / v_med3), turns them into ascending keys 32258 * hamming + trainIdx and merges them into the running pair;
// no per-element key has to be built.  The MFMA result layout puts the query on the lane (column) and 16 train rows in
// the registers, so the scan is lane-local until the two half-wavefronts of a column are merged at the very end.
//
// Train rows are expanded from bits to bytes by the workgroup itself (one dword of bits per thread and tile) into a
// double-buffered LDS tile (row pitch 272 B: the 16-byte fragment reads of 16 lanes hit 64 distinct banks); the query
// fragments are expanded once into registers.  The k order inside a fragment is the same for both operands (byte
// position = bit index), which is all a dot product needs.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define MM_THREADS 256
#define MM_QPB 256        // queries per workgroup: two 32-column MFMA blocks per wavefront
#define MM_TT 32          // train rows per tile
#define MM_PITCH 272
#define MM_C 32258        // ascending key = MM_C * hamming + trainIdx
#define MM_M (16129 * 256 + 31)
#define MM_MAX_TRAIN 32258
#define MM_NONE 0x7FFFFFFF

__device__ __forceinline__ uint32_t expand4(uint32_t nib) {  // 4 bits -> 4 bytes: set -> +127 (0x7F), clear -> -127 (0x81)
    return 0x81818181u - ((nib * 0x00408102u) & 0x02020202u);
}
__device__ __forceinline__ v4i expand16(uint32_t hw) {
    v4i r;
    r.x = (int)expand4(hw & 15u); r.y = (int)expand4((hw >> 4) & 15u);
    r.z = (int)expand4((hw >> 8) & 15u); r.w = (int)expand4((hw >> 12) & 15u);
    return r;
}

__device__ __forceinline__ int med3_i32(int a, int b, int c) {  // the compiler only forms v_med3_i32 for constant clamps
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 32 train bits -> 32 bytes of the LDS tile through the byte table (4 lookups instead of 32 vector instructions)
__device__ __forceinline__ void stage_train(uint8_t* dst, uint32_t w, const uint2* lut) {
    const uint2 e0 = lut[w & 255u], e1 = lut[(w >> 8) & 255u], e2 = lut[(w >> 16) & 255u], e3 = lut[w >> 24];
    *(uint4*)dst = make_uint4(e0.x, e0.y, e1.x, e1.y);
    *(uint4*)(dst + 16) = make_uint4(e2.x, e2.y, e3.x, e3.y);
}

// two largest of the 16 accumulator registers -> ascending keys -> merged into the running (k0 <= k1)
__device__ __forceinline__ void fold_tile(const v16i& acc, int mtb, int& k0, int& k1) {
    int m0 = max(acc[0], acc[1]), m1 = min(acc[0], acc[1]);
#pragma unroll
    for (int k = 2; k < 16; k++) {  // m0 >= m1: the second largest of {m0, m1, x} is their median
        m1 = med3_i32(m0, m1, acc[k]);
        m0 = max(m0, acc[k]);
    }
    const int v0 = mtb - m0, v1 = mtb - m1;  // v0 <= v1
    k1 = min(min(k1, v1), max(k0, v0));
    k0 = min(k0, v0);
}

__global__ __launch_bounds__(MM_THREADS) void k_match_mfma(const uint8_t* __restrict__ qbase, const uint8_t* __restrict__ tbase,
                                                           size_t q_stride, size_t t_stride, const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ qf, const int32_t* __restrict__ tf,
                                                           int nq_fixed, int nt_fixed, int out_stride, double ratio,
                                                           int32_t* __restrict__ oidx, int32_t* __restrict__ odist,
                                                           uint8_t* __restrict__ opass) {
    __shared__ __attribute__((aligned(16))) uint8_t s_t[2][MM_TT * MM_PITCH];
    __shared__ uint2 s_lut[256];  // 8 bits -> 8 bytes of +-127
    const int pair = blockIdx.y;
    const int qfr = qf ? qf[pair] : pair, tfr = tf ? tf[pair] : pair;
    const int nq = counts ? min(counts[qfr], out_stride) : nq_fixed;
    const int nt = counts ? min(counts[tfr], out_stride) : nt_fixed;
    if (blockIdx.x * MM_QPB >= nq) return;  // uniform over the workgroup
    s_lut[threadIdx.x] = make_uint2(expand4(threadIdx.x & 15u), expand4(threadIdx.x >> 4));
    __syncthreads();
    const uint32_t* q = (const uint32_t*)(qbase + (size_t)qfr * q_stride);
    const uint32_t* t = (const uint32_t*)(tbase + (size_t)tfr * t_stride);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, hh = lane >> 5;
    const int qblock = blockIdx.x * MM_QPB + wv * 64;

    v4i bq[2][8];  // query fragments: lane (col, hh), k-step s <- bits 32 s + 16 hh .. + 15 of query col
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int qi = min(qblock + b * 32 + col, nq - 1);
#pragma unroll
        for (int s = 0; s < 8; s++) bq[b][s] = expand16((q[(size_t)qi * 8 + s] >> (16 * hh)) & 0xFFFFu);
    }
    v16i cinit;
#pragma unroll
    for (int k = 0; k < 16; k++) cinit[k] = 31 - ((k & 3) + 8 * (k >> 2) + 4 * hh);
    int k0[2] = {MM_NONE, MM_NONE}, k1[2] = {MM_NONE, MM_NONE};

    const int ntiles = (nt + MM_TT - 1) / MM_TT;
    const int trow = tid >> 3, td = tid & 7;  // staging: one dword of train bits per thread and tile
    if (ntiles > 0) {
        {
            const uint32_t w = trow < nt ? t[(size_t)trow * 8 + td] : 0u;
            stage_train(s_t[0] + trow * MM_PITCH + td * 32, w, s_lut);
        }
        __syncthreads();
    }
    for (int j = 0; j < ntiles; j++) {
        uint32_t nw = 0;
        const bool more = j + 1 < ntiles;
        if (more) {
            const int r = (j + 1) * MM_TT + trow;
            nw = r < nt ? t[(size_t)r * 8 + td] : 0u;
        }
        const uint8_t* frag = s_t[j & 1] + col * MM_PITCH + hh * 16;
        v16i acc0 = cinit, acc1 = cinit;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const v4i a = *(const v4i*)(frag + s * 32);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[1][s], acc1, 0, 0, 0);
        }
        if (j == ntiles - 1 && (nt & (MM_TT - 1))) {  // rows past the last train descriptor never win
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (j * MM_TT + (k & 3) + 8 * (k >> 2) + 4 * hh >= nt) { acc0[k] = -(1 << 30); acc1[k] = -(1 << 30); }
        }
        const int mtb = MM_M + j * MM_TT;
        fold_tile(acc0, mtb, k0[0], k1[0]);
        fold_tile(acc1, mtb, k0[1], k1[1]);
        if (more) {
            stage_train(s_t[(j + 1) & 1] + trow * MM_PITCH + td * 32, nw, s_lut);
        }
        __syncthreads();
    }
    // the two half-wavefronts of a column hold disjoint train rows: merge, then half hh writes query block hh
    int r0 = MM_NONE, r1 = MM_NONE;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int o0 = __shfl_xor(k0[b], 32, 64), o1 = __shfl_xor(k1[b], 32, 64);
        const int n1 = min(min(k1[b], o1), max(k0[b], o0)), n0 = min(k0[b], o0);
        if (b == hh) { r0 = n0; r1 = n1; }
    }
    const int qi = qblock + hh * 32 + col;
    if (qi < nq) {
        const bool h0 = r0 < MM_C * 257, h1 = r1 < MM_C * 257;
        const int d0 = r0 / MM_C, d1 = r1 / MM_C;
        emit_match(h0 ? r0 - d0 * MM_C : -1, h0 ? d0 : INT_MAX, h1 ? r1 - d1 * MM_C : -1, h1 ? d1 : INT_MAX, ratio,
                   (size_t)pair * out_stride + qi, oidx, odist, opass);
    }
}

int match_launch_pairs(mo_ctx* c, const uint8_t* d_q, const uint8_t* d_t, size_t q_stride, size_t t_stride,
                       const int32_t* d_counts, const int32_t* d_qf, const int32_t* d_tf, int nq_fixed, int nt_fixed,
                       int n_pairs, int out_stride, double ratio, int32_t* d_idx, int32_t* d_dist, uint8_t* d_pass) {
    if (n_pairs <= 0) return MO_OK;
    int nq_max = d_counts ? out_stride : nq_fixed;
    if (nq_max <= 0) return MO_OK;
    if ((d_counts ? out_stride : nt_fixed) >= (1 << 20)) return mo_fail(c, MO_ERR_UNSUPPORTED, "more than 2^20-1 train descriptors");
    const int nt_max = d_counts ? out_stride : nt_fixed;
    if (c->match_mode == 1 && nt_max <= MM_MAX_TRAIN) {
        dim3 grid((nq_max + MM_QPB - 1) / MM_QPB, n_pairs);
        hipLaunchKernelGGL(k_match_mfma, grid, dim3(MM_THREADS), 0, c->stream, d_q, d_t, q_stride, t_stride, d_counts, d_qf,
                           d_tf, nq_fixed, nt_fixed, out_stride, ratio, d_idx, d_dist, d_pass);
        HIPCHK(c, hipGetLastError());
        return MO_OK;
    }
    // descriptor rows are read as 16-byte pieces: the host entry points stage into their own (aligned) buffers, device callers hand in
    // hipMalloc'd / torch arrays (256-byte aligned) with rows of cap x 32 bytes
    if (((((size_t)d_q) | ((size_t)d_t) | q_stride | t_stride) & 15) != 0)
        return mo_fail(c, MO_ERR_ARG, "descriptor arrays must be 16-byte aligned (base pointers and row strides)");
    dim3 grid((nq_max + ML_PER_BLOCK - 1) / ML_PER_BLOCK, n_pairs);
    // a handful of workgroups (the single-pair calls of the host API) would leave most of the 256 CUs idle: slice the train
    // tiles over gridDim.z and merge the per-slice keys
    const int ntile = (nt_max + ML_TILE - 1) / ML_TILE, blocks = (int)grid.x * n_pairs;
    const int n_split = blocks <= ML_SPLIT_BLOCKS ? std::min(std::min(ntile, ML_SPLIT_MAX), 256 / blocks) : 1;
    if (n_split > 1) {
        const size_t need = (size_t)n_pairs * n_split * out_stride * sizeof(uint2);
        if (int rc = mo_reserve(c, c->d_match_part, c->match_part_bytes, need)) return rc;
        grid.z = n_split;
        hipLaunchKernelGGL(k_match_lds, grid, dim3(ML_THREADS), 0, c->stream, d_q, d_t, q_stride, t_stride, d_counts, d_qf,
                           d_tf, nq_fixed, nt_fixed, out_stride, ratio, d_idx, d_dist, d_pass, c->d_match_part);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(k_match_merge, dim3((nq_max + 255) / 256, n_pairs), dim3(256), 0, c->stream,
                           c->d_match_part, n_split, d_counts, d_qf, nq_fixed, out_stride, ratio, d_idx, d_dist,
                           d_pass);
        HIPCHK(c, hipGetLastError());
        return MO_OK;
    }
    hipLaunchKernelGGL(k_match_lds, grid, dim3(ML_THREADS), 0, c->stream, d_q, d_t, q_stride, t_stride, d_counts, d_qf, d_tf,
                       nq_fixed, nt_fixed, out_stride, ratio, d_idx, d_dist, d_pass, (uint2*)nullptr);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

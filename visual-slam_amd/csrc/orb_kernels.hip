// orb_kernels.hip -- hand-written gfx950 kernels of the ORB extractor (replaces cv2.ORB.detectAndCompute /
// compute behind src/orbslam2/extractor.py:61-65,79-83 of the reference).
//
// Pipeline over a batch of frames (one launch per stage, frames in blockIdx.y):
//   gray      BGR -> Y, fixed-point 15-bit                                   (only for 3-channel input)
//   resize    level L from level L-1, INTER_LINEAR_EXACT fixed point          (nlevels-1 dependent launches)
//   fast      FAST-9/16 score + 3x3 NMS + border filter, full-width strips,   raster-ordered candidate slots
//             pixel tiles and the score band staged in LDS; compass pre-test -> polarity stacks -> one-sided 16-bit score
//             of the survivors -> NMS on the listed corners -> bitmap -> ordered compaction
//   select    per (frame, level): retainBest(2q) on FAST score -> Harris -> retainBest(q), exact
//             standard-library permutation replay (select_replay.h), record arrays in LDS
//   blur      7x7 Gaussian, 8-bit quantised taps, separable in LDS, REFLECT_101
//   describe  a quarter wavefront (16 lanes) per keypoint: both windows fetched into registers up front, a group-private
//             LDS patch, intensity-centroid angle (2 disc rows per lane + 16-lane reduction), 256 rotated rBRIEF tests
//             (16 per lane); compute() with caller keypoints: one wavefront per keypoint, 4 tests per lane
// All arithmetic is integer or non-contracted float32 (compile with -ffp-contract=off) so results are
// bit-identical to the CPU oracle.
#include <cstring>

#include "common.h"
#include "select_replay.h"

__constant__ __attribute__((aligned(16))) int8_t c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};

#define WAVE 64

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// XCD affinity (speed only, any mapping is correct): consecutive workgroup ids go round-robin to the 8 XCDs, each with its own
// L2.  Workgroup `lin` of a launch of `batch` frames x `per` workgroups works on frame (lin mod 8) + 8 * ((lin / 8) / per), so that
// all workgroups of a frame share one id residue; the last batch mod 8 frames (a sharded rank extracts a halo frame beyond its
// multiple of 8) keep the plain frame-major mapping.  inv_per: the host's reciprocal of `per` (0 encodes per == 1); exact while
// dividend * divisor < 2^32.
__device__ __forceinline__ void xcd_map(uint32_t lin, uint32_t per, uint32_t inv_per, uint32_t batch, int& frame, int& item) {
    const uint32_t b8 = batch & ~7u, cut = per * b8;
    const bool head = lin < cut;
    const uint32_t n = head ? lin >> 3 : lin - cut;
    const uint32_t q = inv_per ? __umulhi(n, inv_per) : n;  // n / per
    item = (int)(n - q * per);
    frame = (int)(head ? (lin & 7u) + 8u * q : b8 + q);
}

__device__ __forceinline__ const uint8_t* level_ptr(const Plan& P, int L, const uint8_t* gray, const uint8_t* pyr,
                                                    int frame) {
    return L == 0 ? gray + (size_t)frame * P.w * P.h : pyr + (size_t)frame * P.pyr_stride + P.lv[L].off;
}

// ------------------------------------------------------------------ gray ----------------------------
__global__ void k_gray(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, size_t npx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
    gray[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
}

int orb_launch_gray(mo_ctx* c, const uint8_t* d_bgr, int w, int h, int batch, uint8_t* d_gray) {
    size_t npx = (size_t)w * h * batch;
    hipLaunchKernelGGL(k_gray, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, c->stream, d_bgr, d_gray, npx);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// Single-frame host calls: the image sits in PINNED host memory (the context's staging buffer) and this kernel is the upload - it
// reads the mapped host pages over PCIe and writes the dense gray frame in HBM (BGR input is converted on the way, so a colour frame
// never exists in HBM), and clears the call's flag word.  One launch instead of a copy (+ a fill) (+ k_gray); a runtime copy of a
// PAGEABLE array is staged by the host thread and took 36 us of a 0.3 ms detect_and_compute call.
__global__ __launch_bounds__(256) void k_ingest(const uint8_t* __restrict__ src, uint8_t* __restrict__ gray, int npx, int ch, int* flags_clear) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 16;  // 16 pixels per thread
    if (blockIdx.x == 0 && threadIdx.x == 0 && flags_clear) { flags_clear[0] = 0; flags_clear[1] = 0; flags_clear[2] = 0; flags_clear[3] = 0; }
    if (i >= npx) return;
    if (i + 16 <= npx) {
        if (ch == 1) {
            *(uint4*)(gray + i) = *(const uint4*)(src + i);
        } else {
            const uint4* s4 = (const uint4*)(src + (size_t)3 * i);   // 48 bytes, 16-byte aligned (i is a multiple of 16)
            const uint4 a = s4[0], b = s4[1], c = s4[2];
            const uint32_t wsrc[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int byte = 3 * (4 * q + k);
                    const uint32_t bb = (wsrc[byte >> 2] >> (8 * (byte & 3))) & 255u, gg = (wsrc[(byte + 1) >> 2] >> (8 * ((byte + 1) & 3))) & 255u,
                                   rr = (wsrc[(byte + 2) >> 2] >> (8 * ((byte + 2) & 3))) & 255u;
                    v |= ((bb * 3735u + gg * 19235u + rr * 9798u + (1u << 14)) >> 15) << (8 * k);
                }
                o[q] = v;
            }
            *(uint4*)(gray + i) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    for (int j = i; j < npx; j++) {  // tail of a frame whose pixel count is not a multiple of 16
        if (ch == 1) gray[j] = src[j];
        else gray[j] = (uint8_t)(((int)src[3 * (size_t)j] * 3735 + (int)src[3 * (size_t)j + 1] * 19235 + (int)src[3 * (size_t)j + 2] * 9798 + (1 << 14)) >> 15);
    }
}

int orb_launch_ingest(mo_ctx* c, const uint8_t* src_mapped, int w, int h, int ch, uint8_t* d_gray, int* flags_clear) {
    const int npx = w * h;
    hipLaunchKernelGGL(k_ingest, dim3((unsigned)((npx + 4095) / 4096)), dim3(256), 0, c->stream, src_mapped, d_gray, npx, ch, flags_clear);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// ------------------------------------------------------------------ resize --------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}

#define RS_TW 64
#define RS_TH 64
#define RS_SRC_ROWS 136   // source rows a 64-row output tile can touch (scale <= 2) + 1
#define RS_SRC_PITCH 144  // source columns a 64-column output tile can touch (scale <= 2) + alignment lead-in

// INTER_LINEAR_EXACT level L from level L-1: one workgroup = 64x32 output tile.  The source window is staged in LDS
// with aligned dword loads, the per-column / per-row fixed-point coefficients (built on the host in cv2's double
// arithmetic) are cached in LDS, each thread produces 4 adjacent pixels and stores one dword.
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, size_t src_fstride, int spitch, int sw, int sh,
                                                uint8_t* __restrict__ dst, size_t dst_fstride, int dpitch, int dw, int dh,
                                                const int* __restrict__ xofs, const int* __restrict__ xc1,
                                                const int* __restrict__ yofs, const int* __restrict__ yc1,
                                                uint32_t inv_per, uint32_t inv_gx, int org) {
    __shared__ __attribute__((aligned(16))) uint8_t s_src[RS_SRC_ROWS * RS_SRC_PITCH];
    __shared__ int s_xo[RS_TW], s_xc[RS_TW], s_yo[RS_TH], s_yc[RS_TH];
    const int tid = threadIdx.x;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {  // XCD affinity (speed only): all tiles of a frame on one XCD (the level just written is in its L2)
        int rem;
        xcd_map(bx + gridDim.x * (by + gridDim.y * bz), gridDim.x * gridDim.y, inv_per, gridDim.z, bz, rem);
        by = (int)(inv_gx ? __umulhi((uint32_t)rem, inv_gx) : (uint32_t)rem);
        bx = rem - by * (int)gridDim.x;
    }
    const int tx0 = org + bx * RS_TW, ty0 = org + by * RS_TH;  // org: margin of the level nothing reads (orb_launch_pyramid)
    const uint8_t* s = src + (size_t)bz * src_fstride;
    // source window of the tile from four wave-uniform (scalar) table reads, so that the coefficient tables and the
    // source pixels are fetched in the same round trip and one barrier covers both
    const int sx0 = xofs[tx0], sx1 = min(xofs[min(tx0 + RS_TW - 1, dw - 1)] + 1, sw - 1);
    const int sy0 = yofs[ty0], sy1 = min(yofs[min(ty0 + RS_TH - 1, dh - 1)] + 1, sh - 1);
    const int sxa = sx0 & ~3;
    const int ncols = sx1 - sxa + 1, nrows = sy1 - sy0 + 1;
    if (ncols > RS_SRC_PITCH || nrows > RS_SRC_ROWS) return;  // scale factor > 2: not supported by this tile shape (host checks)
    if (tid < RS_TW) {
        int x = min(tx0 + tid, dw - 1);
        s_xo[tid] = xofs[x];
        s_xc[tid] = xc1[x];
    } else if (tid < RS_TW + RS_TH) {
        int y = min(ty0 + tid - RS_TW, dh - 1);
        s_yo[tid - RS_TW] = yofs[y];
        s_yc[tid - RS_TW] = yc1[y];
    }
    if ((spitch & 3) == 0 && (((size_t)s) & 3) == 0) {
        const int ndw = (ncols + 3) >> 2;
        // i / ndw by a 20-bit reciprocal: exact while i * ndw < 2^20 (i < 136 * 36); a generic integer division per dword was
        // two thirds of this kernel's vector instructions
        const uint32_t inv = 0x100000u / (uint32_t)ndw + 1u;
        const uint8_t* s0 = s + (size_t)sy0 * spitch + sxa;
        for (int i = tid; i < nrows * ndw; i += 256) {
            const uint32_t r = ((uint32_t)i * inv) >> 20, c4 = (uint32_t)i - r * (uint32_t)ndw;
            ((uint32_t*)(s_src + r * RS_SRC_PITCH))[c4] = *(const uint32_t*)(s0 + r * (uint32_t)spitch + 4u * c4);
        }
    } else {
        for (int i = tid; i < nrows * ncols; i += 256) {
            int r = i / ncols, cc = i - r * ncols;
            s_src[r * RS_SRC_PITCH + cc] = s[(size_t)(sy0 + r) * spitch + sxa + cc];
        }
    }
    __syncthreads();
    const int c0 = (tid & 15) * 4, x = tx0 + c0;
    if (x >= dw) return;
    int ox[4], ox1[4];
    uint32_t mx1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ox[k] = s_xo[c0 + k] - sxa;
        ox1[k] = min(s_xo[c0 + k] + 1, sw - 1) - sxa;
        mx1[k] = s_xc[c0 + k];
    }
#pragma unroll
    for (int half = 0; half < RS_TH / 16; half++) {  // each thread: 4 adjacent pixels of rows ry and ry + 16
        const int ry = (tid >> 4) + 16 * half, y = ty0 + ry;
        if (y >= dh) break;
        const int oy = s_yo[ry] - sy0, oy1 = min(s_yo[ry] + 1, sh - 1) - sy0;
        const uint32_t my1 = s_yc[ry], my0 = 256 - my1;
        const uint8_t* r0 = s_src + oy * RS_SRC_PITCH;
        const uint8_t* r1 = s_src + oy1 * RS_SRC_PITCH;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t mx0 = 256 - mx1[k];
            uint32_t h0 = mx0 * r0[ox[k]] + mx1[k] * r0[ox1[k]];
            uint32_t h1 = mx0 * r1[ox[k]] + mx1[k] * r1[ox1[k]];
            uint32_t v = my0 * h0 + my1 * h1;
            packed |= ((v + 32768u) >> 16) << (8 * k);
        }
        // dpitch is a multiple of 16 >= dw: the <= 3 bytes past dw land in row padding
        *(uint32_t*)(dst + (size_t)bz * dst_fstride + (size_t)y * dpitch + x) = packed;
    }
}


// Default resize kernel (late round 2): the same INTER_LINEAR_EXACT arithmetic in two separable passes.  k_resize above
// gathers four source BYTES from an LDS window per output pixel and interpolates the two source rows of every output row
// separately (each source row twice over): its LDS pipe was 62 % busy, 45 % of that bank conflicts, next to a vector unit
// at 58 % (profiles/r02_pmc_per_kernel.json).  Here
//   pass 1: every source row of the tile is interpolated ONCE along x, straight from three aligned global dwords per
//           thread (4 adjacent output columns need <= 8 consecutive source bytes at scale <= 2): two v_alignbyte put the
//           window at byte 0, one v_perm per column picks the (left, right) pair as u16x2, one v_dot2_u32_u16 against
//           (256 - c, c) gives the row interpolant (< 2^16); four of them -> one 8-byte LDS store
//   pass 2: an output row reads the interpolants of its two source rows (two ds_read_b64 per 4 pixels), pairs them with
//           v_perm and finishes with one v_dot2 per pixel whose accumulator carries the rounding constant.
// Tables: one packed dword per output column / row (ResizeTab).  Needs 4-byte aligned source rows; k_resize stays as the
// path for sources that are not (dense caller images whose width is not a multiple of 4).
#define RS2_ROWS 136  // source rows a 64-row output tile can touch (scale <= 2) + 1
#define RS2_UNR 5
__global__ __launch_bounds__(256) void k_resize2(const uint8_t* __restrict__ src, size_t src_fstride, int spitch, int sw, int sh,
                                                 uint8_t* __restrict__ dst, size_t dst_fstride, int dpitch, int dw, int dh,
                                                 const uint32_t* __restrict__ xpk, const uint32_t* __restrict__ ypk,
                                                 uint32_t inv_per, uint32_t inv_gx, int org) {
    __shared__ __attribute__((aligned(16))) uint2 s_h[RS2_ROWS][16];  // [source row of the tile][column group]: 4 x u16
    const int tid = threadIdx.x;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {  // XCD affinity (speed only): all tiles of a frame on one XCD (the level just written is in its L2)
        int rem;
        xcd_map(bx + gridDim.x * (by + gridDim.y * bz), gridDim.x * gridDim.y, inv_per, gridDim.z, bz, rem);
        by = (int)(inv_gx ? __umulhi((uint32_t)rem, inv_gx) : (uint32_t)rem);
        bx = rem - by * (int)gridDim.x;
    }
    const int tx0 = org + bx * RS_TW, ty0 = org + by * RS_TH;  // org: margin of the level nothing reads (orb_launch_pyramid)
    // source rows of the tile from two wave-uniform table reads
    const uint32_t ey0 = ypk[ty0], ey1 = ypk[min(ty0 + RS_TH - 1, dh - 1)];
    const int sy0 = (int)(ey0 & 0x7FFFu), sy1 = (int)(ey1 & 0x7FFFu) + (int)((ey1 >> 15) & 1u);
    const int nrows = sy1 - sy0 + 1;
    if (nrows > RS2_ROWS) return;  // scale factor > 2: not supported by this tile shape (host checks)
    const int cg = tid & 15, rr = tid >> 4;
    const uint4 xe = *(const uint4*)(xpk + tx0 + 4 * cg);  // the tables are padded to a multiple of 64 entries
    uint32_t ye[RS_TH / 16];
#pragma unroll
    for (int half = 0; half < RS_TH / 16; half++) ye[half] = ypk[ty0 + rr + 16 * half];
    const uint32_t xk[4] = {xe.x, xe.y, xe.z, xe.w};
    const int ox0 = (int)(xk[0] & 0x7FFFu), ab = ox0 & ~3;
    const uint32_t sh8 = (uint32_t)(ox0 & 3);
    uint32_t sel[4], coef[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t a = (xk[k] & 0x7FFFu) - (uint32_t)ox0, b = a + ((xk[k] >> 15) & 1u), c1 = xk[k] >> 16;
        sel[k] = a | (b << 16) | 0x0C000C00u;       // {byte a, 0, byte b, 0} of the realigned 8-byte window
        coef[k] = (256u - c1) | (c1 << 16);
    }
    // dwords past the end of the row are never needed (every byte used lies left of sw <= spitch): clamp them into the row
    const int o1 = min(ab + 4, spitch - 4), o2 = min(ab + 8, spitch - 4);
    const uint8_t* s0 = src + (size_t)bz * src_fstride + (size_t)sy0 * spitch;
    // RS2_UNR rows per thread and trip with ALL their loads issued before the first is used: at scale 1.2 a tile touches <= 78
    // source rows = 5 rows per thread, so the kernel makes one global round trip instead of five dependent ones (its
    // workgroups live a few microseconds; the chain of round trips, not the instruction count, is what bounds them)
    for (int r0 = rr; r0 < nrows; r0 += 16 * RS2_UNR) {
        uint32_t w[RS2_UNR][3];
#pragma unroll
        for (int i = 0; i < RS2_UNR; i++) {
            const uint8_t* row = s0 + (uint32_t)(min(r0 + 16 * i, nrows - 1) * spitch);  // rows past the window re-read its last row
            w[i][0] = *(const uint32_t*)(row + ab); w[i][1] = *(const uint32_t*)(row + o1); w[i][2] = *(const uint32_t*)(row + o2);
        }
#pragma unroll
        for (int i = 0; i < RS2_UNR; i++) {
            const int r = r0 + 16 * i;
            if (r >= nrows) break;
            const uint32_t u0 = __builtin_amdgcn_alignbyte(w[i][1], w[i][0], sh8), u1 = __builtin_amdgcn_alignbyte(w[i][2], w[i][1], sh8);
            uint32_t hv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) hv[k] = udot2(__builtin_amdgcn_perm(u1, u0, sel[k]), coef[k], 0u);
            s_h[r][cg] = make_uint2(hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16));
        }
    }
    __syncthreads();
    const int x = tx0 + 4 * cg;
    if (x >= dw) return;
#pragma unroll
    for (int half = 0; half < RS_TH / 16; half++) {
        const int y = ty0 + rr + 16 * half;
        if (y >= dh) break;
        const uint32_t e = ye[half], c1 = e >> 16, cy = (256u - c1) | (c1 << 16);
        const int oy = (int)(e & 0x7FFFu) - sy0, oy1 = oy + (int)((e >> 15) & 1u);
        const uint2 a = s_h[oy][cg], b = s_h[oy1][cg];
        const uint32_t v0 = udot2(__builtin_amdgcn_perm(b.x, a.x, 0x05040100u), cy, 32768u);
        const uint32_t v1 = udot2(__builtin_amdgcn_perm(b.x, a.x, 0x07060302u), cy, 32768u);
        const uint32_t v2 = udot2(__builtin_amdgcn_perm(b.y, a.y, 0x05040100u), cy, 32768u);
        const uint32_t v3 = udot2(__builtin_amdgcn_perm(b.y, a.y, 0x07060302u), cy, 32768u);
        // (v + 32768) >> 16 < 256 is byte 2 of every sum
        const uint32_t packed = __builtin_amdgcn_perm(v1, v0, 0x0C0C0602u) | (__builtin_amdgcn_perm(v3, v2, 0x0C0C0602u) << 16);
        // dpitch is a multiple of 16 >= dw: the <= 3 bytes past dw land in row padding
        *(uint32_t*)(dst + (size_t)bz * dst_fstride + (size_t)y * dpitch + x) = packed;
    }
}

// margin: only [margin, w - margin) x [margin, h - margin) of levels 1.. is produced (a multiple of 4).  The pipeline passes
// the blur's margin - 4 (8 at edge_threshold 31): FAST stages from column 16 / row 27, the Harris and orientation windows stay
// 16 px inside, the blur tiling starts at 12 and reads from 8, and the next level's [8, ..) only needs this level's [9, ..).
// compute() with caller keypoints and the probes pass 0.
int orb_launch_pyramid(mo_ctx* c, const uint8_t* d_gray, int batch, int nlevels, int margin) {
    const Plan& P = c->plan;
    for (int L = 1; L < nlevels; L++) {
        const LevelInfo& s = P.lv[L - 1];
        const LevelInfo& d = P.lv[L];
        const uint8_t* src = L == 1 ? d_gray : c->d_pyr + s.off;
        size_t sfs = L == 1 ? (size_t)P.w * P.h : (size_t)P.pyr_stride;
        const ResizeTab& t = c->rtab[L];
        const int org = (d.w > 2 * margin + 8 && d.h > 2 * margin + 8) ? margin : 0;
        dim3 grid((d.w - 2 * org + RS_TW - 1) / RS_TW, (d.h - 2 * org + RS_TH - 1) / RS_TH, batch);
        const uint32_t per = grid.x * grid.y, inv_per = per > 1 ? 0xFFFFFFFFu / per + 1u : 0u, inv_gx = grid.x > 1 ? 0xFFFFFFFFu / grid.x + 1u : 0u;
        const bool al4 = ((((size_t)src) | sfs | (size_t)s.pitch) & 3) == 0 && s.pitch >= 12;
        if (al4 && t.two_pass_ok)
            hipLaunchKernelGGL(k_resize2, grid, dim3(256), 0, c->stream, src, sfs, s.pitch, s.w, s.h, c->d_pyr + d.off,
                               (size_t)P.pyr_stride, d.pitch, d.w, d.h, t.xpk, t.ypk, inv_per, inv_gx, org);
        else
            hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, c->stream, src, sfs, s.pitch, s.w, s.h, c->d_pyr + d.off,
                               (size_t)P.pyr_stride, d.pitch, d.w, d.h, t.xofs, t.xc1, t.yofs, t.yc1, inv_per, inv_gx, org);
    }
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// ------------------------------------------------------------------ blur ----------------------------
#define BT_W 64
#define BT_QW (BT_W / 4)      // quads (4 adjacent outputs) per tile row
#define BT_PP (256 / BT_QW)   // row pairs one pass of the workgroup covers
#define BT_H 58   // output rows per tile: 58 + 6 halo rows = 32 row pairs, two passes of 16 (round 1: 26 rows; half the workgroups,
                  // 10 % instead of 23 % halo rows)
#define BT_PW (BT_W + 16)  // LDS pixel-tile pitch: 4 (aligned lead-in) + BT_W + 3 halo, rounded to a multiple of 16
#define BT_ROWS (BT_H + 6)


// {sat_u8(a >> 16), sat_u8(b >> 16)} in bits 0-7 and 8-15 (a, b < 2^31): five slow-class instructions pack four outputs instead
// of a shift, a min and a shift-or each
__device__ __forceinline__ uint32_t hi16_pair_sat_u8(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(__builtin_amdgcn_perm(b, a, 0x07060302u)));
    return d;
}

// 7x7 Gaussian (8-bit quantised taps, sum 257), separable through LDS: u8 tile -> u16 row sums -> u8 output, all in
// integer dot instructions.  Row pass: the 7 taps of one output are two v_dot4_u32_u8 on byte windows cut out of three
// aligned dwords with v_alignbyte.  The u16 row sums of tile rows 2p and 2p+1 are stored interleaved in one dword per
// column, so the column pass is four v_dot2_u32_u16 per output (tap pairs shifted by one row for odd output rows).
// Interior tiles are staged with aligned dword loads; tiles touching the level border index with REFLECT_101.
__global__ __launch_bounds__(256) void k_blur(Plan P, const uint32_t* __restrict__ tile_tab, uint32_t inv_per, int org,
                                              const uint8_t* __restrict__ gray, const uint8_t* __restrict__ pyr,
                                              uint8_t* __restrict__ blur) {
    __shared__ __attribute__((aligned(16))) uint8_t s_px[BT_ROWS * BT_PW];
    __shared__ __attribute__((aligned(16))) uint32_t s_row[(BT_ROWS / 2) * BT_W];  // [row pair][column] = lo: even row, hi: odd
    int tile = blockIdx.x, frame = blockIdx.y;
    xcd_map(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, inv_per, gridDim.y, frame, tile);  // XCD affinity (speed only)
    // one wave-uniform table read instead of a level search and two divisions per workgroup (the scalar prologue was as long
    // as the vector body of these short-lived workgroups)
    const uint32_t te = tile_tab[tile];
    const int L = te & 0xFF;
    const LevelInfo lv = P.lv[L];
    const int tx0 = org + (int)((te >> 8) & 0xFFF) * BT_W, ty0 = org + (int)(te >> 20) * BT_H;  // org: margin the tiling skips
    const uint8_t* img = level_ptr(P, L, gray, pyr, frame);
    const int tid = threadIdx.x;
    // s_px column c holds level column tx0 - 4 + c  (c = 1 .. BT_W + 6 are used).  Rows are reflected per tile row
    // (REFLECT_101); a dword whose four columns lie inside the level is one aligned load, the few that touch the left or
    // right border are assembled from reflected bytes -- border tiles cost about the same as interior ones.
    if ((lv.pitch & 3) == 0 && (((size_t)img) & 3) == 0) {
        // BT_QW interior dwords per row by BT_PP rows per pass, then the two halo dwords of every row: no division.  Tiles that
        // lie inside the level with their halo (wave-uniform test) skip the reflections and the bounds checks.
        auto stage = [&](int r, int c4) {
            const int y = reflect101(ty0 + r - 3, lv.h), x0 = tx0 - 4 + 4 * c4;
            const uint8_t* row = img + (uint32_t)(y * lv.pitch);
            uint32_t v;
            if (x0 >= 0 && x0 + 3 < lv.w) v = *(const uint32_t*)(row + x0);
            else {
                v = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) v |= (uint32_t)row[reflect101(x0 + b, lv.w)] << (8 * b);
            }
            ((uint32_t*)(s_px + r * BT_PW))[c4] = v;
        };
        if (ty0 >= 3 && ty0 + BT_ROWS - 3 <= lv.h && tx0 >= 4 && tx0 + BT_W + 4 <= lv.w) {
            const uint8_t* base = img + (uint32_t)((ty0 - 3) * lv.pitch + tx0 - 4);
#pragma unroll
            for (int r = tid / BT_QW; r < BT_ROWS; r += BT_PP)
                ((uint32_t*)(s_px + r * BT_PW))[1 + tid % BT_QW] = *(const uint32_t*)(base + (uint32_t)(r * lv.pitch) + 4 + 4 * (tid % BT_QW));
            if (tid < 2 * BT_ROWS)
                ((uint32_t*)(s_px + (tid >> 1) * BT_PW))[(tid & 1) * (BT_QW + 1)] =
                    *(const uint32_t*)(base + (uint32_t)((tid >> 1) * lv.pitch) + (tid & 1) * (BT_W + 4));
        } else {
            for (int r = tid / BT_QW; r < BT_ROWS; r += BT_PP) stage(r, 1 + tid % BT_QW);
            for (int i = tid; i < 2 * BT_ROWS; i += 256) stage(i >> 1, (i & 1) * (BT_QW + 1));
        }
    } else {
        for (int i = tid; i < BT_ROWS * (BT_W + 6); i += 256) {
            int r = i / (BT_W + 6), cidx = i - r * (BT_W + 6);
            int y = reflect101(ty0 + r - 3, lv.h), x = reflect101(tx0 + cidx - 3, lv.w);
            s_px[r * BT_PW + cidx + 1] = img[(size_t)y * lv.pitch + x];
        }
    }
    __syncthreads();
    const uint32_t g0 = P.gk[0], g1 = P.gk[1], g2 = P.gk[2], g3 = P.gk[3];
#pragma unroll
    for (int h2 = 0; h2 < BT_ROWS / 2 / BT_PP; h2++) {   // row pass: one task = 4 adjacent outputs of tile rows 2p and 2p+1 (BT_PP row pairs x BT_QW quads per pass)
        const uint32_t ta = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24), tb = g2 | (g1 << 8) | (g0 << 16);
        const int p = tid / BT_QW + BT_PP * h2, q = tid % BT_QW;
        uint32_t o[2][4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t* pw = (const uint32_t*)(s_px + (2 * p + h) * BT_PW + 4 * q);
            const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2];
            // output column 4q+k is centred on s_px column 4q+k+4: taps over bytes k+1 .. k+7 of (w0, w1, w2)
            o[h][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), ta, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), tb, 0, false), false);
            o[h][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), ta, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), tb, 0, false), false);
            o[h][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), ta, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), tb, 0, false), false);
            o[h][3] = __builtin_amdgcn_udot4(w1, ta, __builtin_amdgcn_udot4(w2, tb, 0, false), false);
        }
        uint4 packed;  // each sum <= 257 * 255 = 65535
        packed.x = o[0][0] | (o[1][0] << 16);
        packed.y = o[0][1] | (o[1][1] << 16);
        packed.z = o[0][2] | (o[1][2] << 16);
        packed.w = o[0][3] | (o[1][3] << 16);
        *(uint4*)(&s_row[p * BT_W + 4 * q]) = packed;
    }
    __syncthreads();
#pragma unroll
    for (int h2 = 0; h2 < BT_ROWS / 2 / BT_PP; h2++) {   // column pass: each thread 4 adjacent columns of output rows 2p and 2p+1, from row pairs p .. p+3
        const int p = tid / BT_QW + BT_PP * h2, c0 = (tid % BT_QW) * 4;
        const int x = tx0 + c0, y = ty0 + 2 * p;
        if (p < BT_H / 2 && y < lv.h && x < lv.w) {
            // even output row 2p   : rows 2p .. 2p+6   = pairs (g0,g1) (g2,g3) (g2,g1) (g0, 0)
            // odd  output row 2p+1 : rows 2p+1 .. 2p+7 = pairs ( 0,g0) (g1,g2) (g3,g2) (g1,g0)
            const uint32_t e0 = g0 | (g1 << 16), e1 = g2 | (g3 << 16), e2 = g2 | (g1 << 16), e3 = g0;
            const uint32_t d0 = g0 << 16, d1 = g1 | (g2 << 16), d2 = g3 | (g2 << 16), d3 = g1 | (g0 << 16);
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = *(const uint4*)(&s_row[(p + j) * BT_W + c0]);
            uint32_t se[4], so[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t a0 = k == 0 ? v[0].x : k == 1 ? v[0].y : k == 2 ? v[0].z : v[0].w;
                const uint32_t a1 = k == 0 ? v[1].x : k == 1 ? v[1].y : k == 2 ? v[1].z : v[1].w;
                const uint32_t a2 = k == 0 ? v[2].x : k == 1 ? v[2].y : k == 2 ? v[2].z : v[2].w;
                const uint32_t a3 = k == 0 ? v[3].x : k == 1 ? v[3].y : k == 2 ? v[3].z : v[3].w;
                se[k] = udot2(a0, e0, udot2(a1, e1, udot2(a2, e2, udot2(a3, e3, 1u << 15))));
                so[k] = udot2(a0, d0, udot2(a1, d1, udot2(a2, d2, udot2(a3, d3, 1u << 15))));
            }
            // (sum >> 16) is at most 257: the high halves of two sums -> one dword (v_perm), both saturated to bytes at once
            const uint32_t pe = hi16_pair_sat_u8(se[0], se[1]) | (hi16_pair_sat_u8(se[2], se[3]) << 16);
            const uint32_t po = hi16_pair_sat_u8(so[0], so[1]) | (hi16_pair_sat_u8(so[2], so[3]) << 16);
            uint8_t* out = blur + (size_t)frame * P.blur_stride + lv.boff + (size_t)y * lv.bpitch + x;
            *(uint32_t*)out = pe;  // bpitch is a multiple of 16 >= w: the <= 3 bytes past w land in row padding
            if (y + 1 < lv.h) *(uint32_t*)(out + lv.bpitch) = po;
        }
    }
}

// margin: the tiling covers [margin, w - margin) x [margin, h - margin) of every level.  The detector's keypoints lie at least
// edge_threshold from the border and rBRIEF samples within 19 px of them, so the pipeline passes (edge_threshold - 19) & ~3
// (12 at the default 31: 255 instead of 286 tiles per 640x480 frame); compute() with caller keypoints and the level probes
// pass 0 (a caller's keypoint on a coarse octave may sample anywhere).
int orb_launch_blur(mo_ctx* c, const uint8_t* d_gray, int batch, int nlevels, int margin) {
    const Plan& P = c->plan;
    if (nlevels < 1 || nlevels > P.nlevels) return mo_fail(c, MO_ERR_ARG, "blur: level count outside the plan");
    const int slot = margin > 0 ? 1 : 0;
    if (slot && c->tile_margin != margin && c->d_tile_tab[1]) { hipFree(c->d_tile_tab[1]); c->d_tile_tab[1] = nullptr; }
    if (!c->d_tile_tab[slot]) {  // (re)built with the plan: free_plan_buffers drops it
        std::vector<uint32_t> tab;
        for (int L = 0; L < P.nlevels; L++) {
            const int cw = std::max(P.lv[L].w - 2 * margin, 1), chh = std::max(P.lv[L].h - 2 * margin, 1);
            const int tx = (cw + BT_W - 1) / BT_W, ty = (chh + BT_H - 1) / BT_H;
            c->tile_cum[slot][L] = (int)tab.size();
            for (int y = 0; y < ty; y++)
                for (int x = 0; x < tx; x++) tab.push_back((uint32_t)L | ((uint32_t)x << 8) | ((uint32_t)y << 20));
        }
        c->tile_cum[slot][P.nlevels] = (int)tab.size();
        HIPCHK(c, hipMalloc((void**)&c->d_tile_tab[slot], tab.size() * sizeof(uint32_t)));
        HIPCHK(c, hipMemcpy(c->d_tile_tab[slot], tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (slot) c->tile_margin = margin;
    }
    const uint32_t per = (uint32_t)c->tile_cum[slot][nlevels], inv_per = per > 1 ? 0xFFFFFFFFu / per + 1u : 0u;  // first nlevels levels
    hipLaunchKernelGGL(k_blur, dim3(per, batch), dim3(256), 0, c->stream, P, c->d_tile_tab[slot], inv_per, margin, d_gray, c->d_pyr,
                       c->d_blur);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// ------------------------------------------------------------------ FAST ----------------------------
__device__ __forceinline__ bool arc9(uint32_t m) {  // 16-bit circular mask has 9 contiguous ones
    uint32_t m32 = m | (m << 16);
    uint32_t x = m32 & (m32 >> 1);
    x &= x >> 2;
    x &= x >> 4;
    x &= m32 >> 8;
    return (x & 0xFFFFu) != 0;
}

// FAST corner score (cornerScore<16>): for a corner, max over the 16 arcs of 9 of min(d) over the arc, minus 1
// (d = centre - circle pixel for a dark corner, circle - centre for a bright one).
__device__ __forceinline__ int arc_score(const int d[16]) {
    int m2[16], m4[16], best = -1000;
#pragma unroll
    for (int i = 0; i < 16; i++) m2[i] = min(d[i], d[(i + 1) & 15]);
#pragma unroll
    for (int i = 0; i < 16; i++) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int m8 = min(m4[i], m4[(i + 4) & 15]);
        int m9 = min(m8, d[(i + 8) & 15]);
        best = max(best, m9);
    }
    return best - 1;
}

// differences centre - circle pixel, OpenCV circle order
#define FAST_LOAD_D(d, v, p, TW)                                                                              \
    d[0] = v - p[3 * TW];   d[1] = v - p[3 * TW + 1];   d[2] = v - p[2 * TW + 2];   d[3] = v - p[TW + 3];        \
    d[4] = v - p[3];        d[5] = v - p[-TW + 3];      d[6] = v - p[-2 * TW + 2];  d[7] = v - p[-3 * TW + 1];   \
    d[8] = v - p[-3 * TW];  d[9] = v - p[-3 * TW - 1];  d[10] = v - p[-2 * TW - 2]; d[11] = v - p[-TW - 3];      \
    d[12] = v - p[-3];      d[13] = v - p[TW - 3];      d[14] = v - p[2 * TW - 2];  d[15] = v - p[3 * TW - 1];

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

// One workgroup = one full-width strip of <= 8 rows of one level of one frame.
//  1. the strip's pixels (+4 rows / +3 columns of halo) are staged in LDS with aligned dword loads
//  2. FAST-9 score of every pixel of the strip and its 1-px ring -> u8 score band in LDS (0 = no corner);
//     work is dealt to the 4 wavefronts in (row, 64-column) units
//  3. 3x3 non-max suppression + border filter + raster-ordered compaction.  Only pixels with a non-zero score (about 7 %
//     on textured frames) can be kept, so phase 2b also lists them (bounded LDS list); after the barrier one lane per
//     LISTED corner compares it with its 8 neighbours and sets its bit in a row-major bitmap of the strip, and the
//     compaction walks the bitmap words (popcount -> block scan -> emit in raster order).  A strip with more corners than
//     the list holds (noise at threshold 0) falls back to a dense scan of the score band that fills the same bitmap.
// 16-bit VOP2 forms (values in the low half of a VGPR): on gfx950 v_sub_u16 / v_min_i16 / v_max_i16 issue at the full
// vector rate while v_min_i32 / v_max_i32 and every packed (VOP3P) form take twice as long (tools/ubench.hip)
__device__ __forceinline__ int sub16(int a, int b) { int r; asm("v_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int min16(int a, int b) { int r; asm("v_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int max16(int a, int b) { int r; asm("v_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// lane mask of (short)a > (short)b: the compare writes the wave-wide mask directly (inactive lanes read as 0)
__device__ __forceinline__ unsigned long long ballot_gt16(int a, int b) {
    unsigned long long m;
    asm("v_cmp_gt_i16 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}

// ds_write_b16 of val at LDS byte address addr by exactly the lanes of mask (a wave mask the caller already holds in scalar
// registers): EXEC is narrowed to the mask around the store instead of rebuilding a per-lane predicate from it (two v_and, a
// 64-bit compare and a branch per push otherwise).  A store with EXEC = 0 is a no-op.
__device__ __forceinline__ void lds_store_u16_masked(unsigned long long mask, uint32_t addr, uint32_t val) {
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "s"(mask), "v"(addr), "v"(val) : "memory", "scc");
}

#define FAST_CORNER_CAP 896   // listed corners per strip (typical: 300); more -> dense fallback
#define FAST_KEEP_WORDS 512   // strip_rows * bw <= 16384 candidate positions (mo_build_plan)
#define FAST_STACK 384        // u16 entries per wavefront: two stacks of <= 191

template <int TW, int NT>  // TW: LDS tile pitch, a compile-time constant so the 16 circle reads use immediate offsets; NT: threads
__global__ __launch_bounds__(NT) void k_fast(Plan P, const uint32_t* __restrict__ strip_tab, uint32_t inv_per, int strip0,
                                              const uint8_t* __restrict__ gray, const uint8_t* __restrict__ pyr,
                                              uint32_t* __restrict__ cand, int* __restrict__ strip_cnt, int score_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int strip = blockIdx.x, frame = blockIdx.y;
    xcd_map(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, inv_per, gridDim.y, frame, strip);  // XCD affinity (speed only)
    const uint32_t se = strip_tab[strip0 + strip];  // one wave-uniform table read instead of a level search (strip0: first strip of this launch's levels)
    const int L = se & 0xFF;
    const LevelInfo lv = P.lv[L];
    strip = (int)(se >> 8);
    if (strip >= lv.nstrips) return;
    const int R = lv.strip_rows;
    const int y0 = lv.by0 + strip * R;
    const int rows = min(R, lv.by0 + lv.bh - y0);
    const int SW = lv.bw + 2;           // scored columns: bx0-1 .. bx0+bw
    const int xs0 = lv.bx0 - 1;
    uint8_t* s_score = smem;            // (R+2) rows x SW
    uint8_t* s_tile = smem + score_bytes;
    constexpr int NW = NT / 64;
    __shared__ int s_wsum[NW];
    // per-wavefront stacks of the pixels that pass the compass pre-test (row << 12 | column): the darker-arc candidates grow
    // up from [0], the brighter-arc candidates down from [FAST_STACK - 1]; each holds <= 63 + 128 entries
    __shared__ __attribute__((aligned(16))) uint16_t s_wq[NW][FAST_STACK];
    __shared__ uint16_t s_corner[FAST_CORNER_CAP];  // band positions (row * SW + column) of the pixels with a non-zero score
    __shared__ int s_ncorner;
    // bit (rr * bw + xx) set <=> border-region pixel survives the 3x3 NMS; overlays the stacks, which are dead by then
    uint32_t* const s_keep = (uint32_t*)&s_wq[0][0];
    static_assert(sizeof(uint16_t) * NW * FAST_STACK >= 4 * FAST_KEEP_WORDS, "bitmap overlays the stacks");
    const uint8_t* img = level_ptr(P, L, gray, pyr, frame);
    const int t = P.fast_threshold;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: loop control stays on the scalar unit

    // ---- 1. stage pixels: rows y0-4 .. y0+rows+3, columns gx0 .. gx0+tw_used (gx0 = xs0-3 rounded down to 16, or to 4
    //         when the level is not 16-byte aligned)
    const bool al16 = (lv.pitch & 15) == 0 && (((size_t)img) & 15) == 0;
    const bool al4 = (lv.pitch & 3) == 0 && (((size_t)img) & 3) == 0;
    const int gx0 = al16 ? (xs0 - 3) & ~15 : (xs0 - 3) & ~3, lead = xs0 - 3 - gx0;
    const int th = rows + 8, gy0 = y0 - 4;
    if (al16) {  // 16 bytes per lane; pitch >= the rounded row end because pitch is a multiple of 16
        const int tw16 = (SW + 6 + lead + 15) >> 4;
        const uint32_t inv = 0xFFFFFFFFu / (uint32_t)tw16 + 1u;  // i / tw16 == mulhi(i, inv) while i * tw16 < 2^32
        for (int i = tid; i < th * tw16; i += NT) {
            const int r = tw16 > 1 ? (int)__umulhi((uint32_t)i, inv) : i, c16 = i - r * tw16;
            *(uint4*)(s_tile + r * TW + 16 * c16) = *(const uint4*)(img + (size_t)(gy0 + r) * lv.pitch + gx0 + 16 * c16);
        }
    } else if (al4) {
        const int tw4 = (SW + 6 + lead + 3) >> 2;
        for (int r = wv; r < th; r += NW) {
            const uint32_t* src = (const uint32_t*)(img + (size_t)(gy0 + r) * lv.pitch + gx0);
            uint32_t* dst = (uint32_t*)(s_tile + r * TW);
            for (int c4 = lane; c4 < tw4; c4 += 64) dst[c4] = src[c4];
        }
    } else {
        const int tw_used = (SW + 6 + lead + 3) & ~3;
        for (int r = wv; r < th; r += NW) {
            const uint8_t* src = img + (size_t)(gy0 + r) * lv.pitch + gx0;
            for (int cc = lane; cc < tw_used; cc += 64) s_tile[r * TW + cc] = (gx0 + cc < lv.w) ? src[cc] : 0;
        }
    }
    for (int i = tid; i < ((rows + 2) * SW + 3) >> 2; i += NT) ((uint32_t*)s_score)[i] = 0;  // score 0 unless phase 2b says otherwise
    if (tid == 0) s_ncorner = 0;
    __syncthreads();

    // ---- 2. scores.
    // 2a (every pixel, 5 LDS reads): a 9-arc of the 16-circle always contains two ADJACENT compass pixels (positions 0, 4, 8, 12),
    //     so a pixel without two adjacent compass differences above t AND without two below -t cannot be a corner: score 0.  Which of
    //     the two holds also fixes the ONLY polarity the pixel can be a corner with (two 9-arcs of a 16-circle overlap, so a
    //     pixel cannot have both a darker and a brighter arc): survivors are appended to a per-wavefront LDS stack with that
    //     polarity (ballot + prefix popcount); the rare pixel that passes both tests is pushed once per polarity.
    // 2b (survivors only, 64 at a time off the top of the stack): one-sided score L = max over the 16 arcs of min(d' over
    //     the 9-arc), d' = +-(centre - circle pixel) by polarity; corner <=> L > t, cornerScore = L - 1 (the other polarity
    //     cannot exceed -L).  One pixel per lane, 16-bit VOP2 min/max/sub: measured on MI355X (tools/ubench.hip) these issue
    //     at 2.3 cycles per wavefront, twice the rate of the packed (v_pk_*) and 32-bit min/max forms.
    typedef const volatile __attribute__((address_space(3))) uint8_t lds_cvu8;  // volatile: byte reads stay separate
    const int t_list = max(t, 1);  // a pixel is listed for phase 3 when its stored score (L - 1 for L > t) is non-zero
    // stack entry = row << xbits | column: 16-row strips exist only for bw <= 1024 (strip_rows * bw <= 16384), narrower strips
    // have <= 10 scored rows and bw + 2 <= 4096
    const int xbits = R > 8 ? 11 : 12;
    const uint32_t xmask = (1u << xbits) - 1u;
    auto score_one = [&](uint32_t e, int flip, bool own) {  // own: not a filler lane
        const int r = e >> xbits, x = e & xmask;
        const int pos = r * SW + x;
        // flip (wave-uniform) = 0xFF: brighter-arc polarity, bytes complemented: (255 - v) - (255 - p) = p - v
        lds_cvu8* p = (lds_cvu8*)&s_tile[(r + 3) * TW + x + 3 + lead];
        // d = centre - circle pixel for BOTH polarities: the brighter-arc score max over arcs of min9(-d) = -(min over arcs of max9(d)) runs
        // the same network with min and max exchanged (flip is a constant at every call site) and negates the result, instead of
        // complementing all 17 pixels.  (Written as plain 16-bit C the compiler fuses pairs into v_min3 / v_max3_i16, which issue at
        // half rate: 0.785 against 0.735 ms per 256 frames, profiles/r03_ab_fast16.txt - hence the VOP2 forms by inline asm.)
        const int v = (int)p[0];
        int d[16];
#define FAST_D(k, o) d[k] = sub16(v, (int)p[o]);
        FAST_D(0, 3 * TW)        FAST_D(1, 3 * TW + 1)    FAST_D(2, 2 * TW + 2)    FAST_D(3, TW + 3)
        FAST_D(4, 3)             FAST_D(5, -TW + 3)       FAST_D(6, -2 * TW + 2)   FAST_D(7, -3 * TW + 1)
        FAST_D(8, -3 * TW)       FAST_D(9, -3 * TW - 1)   FAST_D(10, -2 * TW - 2)  FAST_D(11, -TW - 3)
        FAST_D(12, -3)           FAST_D(13, TW - 3)       FAST_D(14, 2 * TW - 2)   FAST_D(15, 3 * TW - 1)
#undef FAST_D
        auto lo = [flip](int a, int b2) { return flip ? max16(a, b2) : min16(a, b2); };
        auto hi = [flip](int a, int b2) { return flip ? min16(a, b2) : max16(a, b2); };
        int mn3[16];
#pragma unroll
        for (int i = 0; i < 16; i++) mn3[i] = lo(d[i], lo(d[(i + 1) & 15], d[(i + 2) & 15]));
        int Ls = lo(mn3[0], lo(mn3[3], mn3[6]));
#pragma unroll
        for (int i = 1; i < 16; i++) Ls = hi(Ls, lo(mn3[i], lo(mn3[(i + 3) & 15], mn3[(i + 6) & 15])));
        const int L = flip ? -(int)(short)Ls : (int)(short)Ls;
        const int b = (int)(short)L;
        if (own && b > t) s_score[pos] = (uint8_t)(b - 1);  // the band is zero-filled; the other polarity of the pixel writes nothing
        // list the corners for phase 3 (order is irrelevant): one LDS atomic per wavefront call
        const bool c = own && b > t_list;
        const unsigned long long m = __ballot(c);
        if (m) {  // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_ncorner, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            const int o = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (c && o < FAST_CORNER_CAP) s_corner[o] = (uint16_t)pos;
        }
    };
    {
        uint16_t* wq = s_wq[wv];
        const int nxc = (SW + 63) >> 6;
        int qd = 0, qb = 0;  // wave-uniform stack fills (darker-arc / brighter-arc candidates)
        // 64-pixel chunks of the scored rows, numbered row-major; each trip takes two of them (chunk c and c + 4).
        // r*, j*, qd, qb are wave-uniform (scalar registers).
        int ra = 0, ja = wv, rb = 0, jb = wv + NW;
        while (ja >= nxc) { ja -= nxc; ra++; }
        while (jb >= nxc) { jb -= nxc; rb++; }
        const int tv = t, ntv = -t;
        // branch-free: lanes past the last scored column read a clamped (valid) address and are masked out of the result;
        // the window base sits 3 rows and 3 columns before the pixel so that all five reads use non-negative immediates.
        // largest of the four adjacent-pair minima > t <=> two adjacent differences above t; smallest of the pair maxima < -t likewise
        auto chunk = [&](int r, int j) {
            const int x = (j << 6) + lane;
            const int rem = SW - (j << 6);  // scored columns left from this chunk on (wave-uniform, >= 1)
            const unsigned long long valid = rem >= 64 ? ~0ull : (1ull << rem) - 1ull;
            const uint8_t* w = &s_tile[r * TW + lead + min(x, SW - 1)];
            const int v = w[3 * TW + 3];
            const int d0 = sub16(v, w[6 * TW + 3]), d4 = sub16(v, w[3 * TW + 6]), d8 = sub16(v, w[3]), d12 = sub16(v, w[3 * TW]);
#ifdef FAST_COMPASS_ANY2
            const int mn_a = min16(d0, d4), mx_a = max16(d0, d4), mn_b = min16(d8, d12), mx_b = max16(d8, d12);
            const int second_hi = max16(max16(min16(mx_a, mx_b), mn_a), mn_b);
            const int second_lo = min16(min16(max16(mn_a, mn_b), mx_a), mx_b);
#else
            // two ADJACENT compass pixels (a 9-arc holds two or three consecutive multiples of 4): 14 instead of 10 min / max, but 0.62
            // instead of 0.67 survivors per pixel on the SURVEY-8d texture
            const int second_hi = max16(max16(min16(d0, d4), min16(d4, d8)), max16(min16(d8, d12), min16(d12, d0)));
            const int second_lo = min16(min16(max16(d0, d4), max16(d4, d8)), min16(max16(d8, d12), max16(d12, d0)));
#endif
            const unsigned long long md = ballot_gt16(second_hi, tv) & valid;   // centre above >= 2 compass pixels by more than t
            const unsigned long long mb = ballot_gt16(ntv, second_lo) & valid;  // centre below >= 2 compass pixels by more than t
            const uint32_t e = (uint32_t)((r << xbits) | x);
            const uint32_t wq_lds = (uint32_t)(uintptr_t)wq;  // LDS byte address of this wavefront's stacks
            lds_store_u16_masked(md, wq_lds + 2u * (uint32_t)(qd + __builtin_amdgcn_mbcnt_hi((unsigned)(md >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)md, 0u))), e);
            qd += __popcll(md);
            lds_store_u16_masked(mb, wq_lds + 2u * (uint32_t)(FAST_STACK - 1 - qb - (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u))), e);
            qb += __popcll(mb);
        };
        while (ra < rows + 2) {
            chunk(ra, ja);
            if (rb < rows + 2) chunk(rb, jb);  // wave-uniform
            if (qd >= 64 || qb >= 64) {
                replay::wave_sync();  // stack writes of this wavefront are visible to all its lanes
                while (qd >= 64) { qd -= 64; score_one(wq[qd + lane], 0, true); }
                while (qb >= 64) { qb -= 64; score_one(wq[FAST_STACK - 1 - qb - lane], 0xFF, true); }
                replay::wave_sync();  // the entries just read may be overwritten by the next pushes
            }
            ja += 2 * NW;
            while (ja >= nxc) { ja -= nxc; ra++; }
            jb += 2 * NW;
            while (jb >= nxc) { jb -= nxc; rb++; }
        }
        replay::wave_sync();
        // wave-uniform tails; lanes past a stack re-read its last entry
        if (qd > 0) score_one(wq[min(lane, qd - 1)], 0, lane < qd);
        if (qb > 0) score_one(wq[FAST_STACK - 1 - min(lane, qb - 1)], 0xFF, lane < qb);
    }
    __syncthreads();
    for (int i = tid; i < FAST_KEEP_WORDS; i += NT) s_keep[i] = 0;  // the stacks are dead: their space becomes the keep bitmap
    __syncthreads();

    // ---- 3. NMS + border filter on the listed corners -> bitmap -> raster-ordered compaction
    const int nitems = rows * lv.bw;
    {
        const int nc = s_ncorner;  // block-uniform
        const uint32_t inv_sw = 0xFFFFFFFFu / (uint32_t)SW + 1u;  // p / SW == mulhi(p, inv_sw) while p * SW < 2^32 (p < 65536, SW <= 16386)
        auto nms_at = [&](int p) {  // band position p = r * SW + x: keep <=> inside the border region and strictly above its 8 neighbours
            const int r = (int)__umulhi((uint32_t)p, inv_sw), x = p - r * SW;
            if (r < 1 || r > rows || x < 1 || x > lv.bw) return;
            const uint8_t* c = &s_score[p];
            const int mid = c[0];
            const int nb = max(max(max((int)c[-SW - 1], (int)c[-SW]), max((int)c[-SW + 1], (int)c[-1])),
                               max(max((int)c[1], (int)c[SW - 1]), max((int)c[SW], (int)c[SW + 1])));
            if (mid > nb) {
                const int i = (r - 1) * lv.bw + (x - 1);
                atomicOr(&s_keep[i >> 5], 1u << (i & 31));
            }
        };
        if (nc <= FAST_CORNER_CAP) {
            for (int k = tid; k < nc; k += NT) nms_at(s_corner[k]);
        } else {  // the list overflowed: dense scan of the score band, one dword (4 positions) per lane and trip
            const int nband = (rows + 2) * SW;
            for (int k = tid; 4 * k < nband; k += NT) {
                uint32_t w = ((const uint32_t*)s_score)[k];
#pragma unroll
                for (int bq = 0; bq < 4; bq++)
                    if (((w >> (8 * bq)) & 0xFFu) && 4 * k + bq < nband) nms_at(4 * k + bq);
            }
        }
    }
    __syncthreads();
    const int nwords = (nitems + 31) >> 5, wpt = (nwords + NT - 1) / NT;  // wpt <= 2
    uint32_t kw[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < 2; q++)
        if (q < wpt && tid * wpt + q < nwords) kw[q] = s_keep[tid * wpt + q];
    const int cnt = __popc(kw[0]) + __popc(kw[1]);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wv; k++) base += s_wsum[k];
    int total = 0;
    for (int k = 0; k < NW; k++) total += s_wsum[k];
    int pos = base + incl - cnt;
    uint32_t* out = cand + (size_t)frame * P.cand_stride + lv.cand_off + (size_t)strip * lv.strip_cap;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        uint32_t bits = kw[q];
        while (bits) {
            const int j = __ffs((int)bits) - 1;
            bits &= bits - 1;
            const int i = (tid * wpt + q) * 32 + j;
            const int rr = lv.bw > 1 ? (int)__umulhi((uint32_t)i, lv.inv_bw) : i, xx = i - rr * lv.bw;  // exact while i * bw < 2^32
            const int sc = s_score[(rr + 1) * SW + xx + 1];
            if (pos < lv.strip_cap) out[pos] = ((uint32_t)sc << 24) | ((uint32_t)(y0 + rr) << 12) | (uint32_t)(lv.bx0 + xx);
            pos++;
        }
    }
    if (tid == 0) strip_cnt[(size_t)frame * P.strips_per_frame + lv.strip_base + strip] = min(total, lv.strip_cap);
}

template <int TW, int NT> static int launch_fast_tw(mo_ctx* c, const uint8_t* d_gray, int batch, size_t score_bytes, int max_rows,
                                                    int strip0, int nstrips) {
    const Plan& P = c->plan;
    size_t lds = score_bytes + (size_t)(max_rows + 8) * TW + 16;
    if (lds > 128 * 1024) return mo_fail(c, MO_ERR_UNSUPPORTED, "level too wide for the FAST strip kernel");
    const unsigned bit = (TW == 704 ? 1u : TW == 1344 ? 2u : TW == 2112 ? 4u : TW == 608 ? 32u : 8u) << (NT == 512 ? 8 : 0);
    if (!(c->lds_attr_done & bit)) {
        HIPCHK(c, hipFuncSetAttribute((const void*)k_fast<TW, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        c->lds_attr_done |= bit;
    }
    if (!c->d_strip_tab) {  // (re)built with the plan: free_plan_buffers drops it
        std::vector<uint32_t> tab((size_t)P.strips_per_frame, 0xFFFFFF00u);  // (no level has that many strips: the kernel returns)
        for (int L = 0; L < P.nlevels; L++)
            for (int st = 0; st < P.lv[L].nstrips; st++) tab[(size_t)P.lv[L].strip_base + st] = (uint32_t)L | ((uint32_t)st << 8);
        HIPCHK(c, hipMalloc((void**)&c->d_strip_tab, tab.size() * sizeof(uint32_t)));
        HIPCHK(c, hipMemcpy(c->d_strip_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->n_strip_tab = (int)tab.size();
    }
    const uint32_t per = (uint32_t)nstrips, inv_per = per > 1 ? 0xFFFFFFFFu / per + 1u : 0u;
    hipLaunchKernelGGL((k_fast<TW, NT>), dim3(nstrips, batch), dim3(NT), lds, c->stream, P, c->d_strip_tab, inv_per, strip0, d_gray,
                       c->d_pyr, c->d_cand, c->d_strip_cnt, (int)score_bytes);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// levels [level_lo, level_hi) of every frame (the strip table is level-major: a level range is a strip range)
int orb_launch_fast(mo_ctx* c, const uint8_t* d_gray, int batch, int level_lo, int level_hi) {
    const Plan& P = c->plan;
    level_lo = std::max(level_lo, 0); level_hi = std::min(level_hi, P.nlevels);
    if (level_lo >= level_hi) return MO_OK;
    const int strip0 = P.lv[level_lo].strip_base;
    const int nstrips = (level_hi < P.nlevels ? P.lv[level_hi].strip_base : P.strips_per_frame) - strip0;
    if (nstrips < 1) return MO_OK;
    size_t score_bytes = 0;
    int tw_need = 0, max_rows = 1;
    for (int L = 0; L < P.nlevels; L++) {
        const LevelInfo& v = P.lv[L];
        score_bytes = std::max(score_bytes, (((size_t)(v.strip_rows + 2) * (v.bw + 2) + 15) & ~(size_t)15));
        tw_need = std::max(tw_need, (v.bw + 2 + 6 + 15 + 15) & ~15);
        max_rows = std::max(max_rows, v.strip_rows);
    }
    if (P.strips_per_frame < 1) return MO_OK;
    // (the kernel is written for any MO_STRIP_ROWS / thread count; 16-row strips with 512-thread workgroups - half the workgroups,
    // 12 % instead of 25 % ring rows - measured 3 % SLOWER than 8 rows x 256 threads on MI355X: eight wavefronts per barrier)
    if (tw_need <= 608) return launch_fast_tw<608, 256>(c, d_gray, batch, score_bytes, max_rows, strip0, nstrips);  // 640-wide frames: 8 workgroups per CU
    if (tw_need <= 704) return launch_fast_tw<704, 256>(c, d_gray, batch, score_bytes, max_rows, strip0, nstrips);
    if (tw_need <= 1344) return launch_fast_tw<1344, 256>(c, d_gray, batch, score_bytes, max_rows, strip0, nstrips);
    if (tw_need <= 2112) return launch_fast_tw<2112, 256>(c, d_gray, batch, score_bytes, max_rows, strip0, nstrips);
    return launch_fast_tw<4160, 256>(c, d_gray, batch, score_bytes, max_rows, strip0, nstrips);
}

// ------------------------------------------------------------------ select --------------------------
#ifndef SEL_BUF_KB
#define SEL_BUF_KB 46
#endif
#define SEL_BUF_BYTES (SEL_BUF_KB * 1024)  // LDS record window (u32 FAST records, then u64 Harris records overlaid): holds the ~9000 candidates a
                                   // dense 640x480 level 0 produces (5.125 B each with the replay's side arrays; SURVEY 8d texture: 6600 -
                                   // 7000); 3 workgroups per CU with the 3.4 KB of Harris windows and the replay scratch
// (SEL_MAXSTRIPS, the strips of one level - 8K frames: 540 at 8 rows -, is in common.h: mo_build_plan keeps a level below it)

// Harris response of a 7x7 block on the raw level (orb.cpp HarrisResponses): int sums, float32 formula
__device__ float harris_response(const uint8_t* img, int pitch, int x0, int y0) {
    int a = 0, b = 0, cc = 0;
    for (int i = -3; i <= 3; i++) {
        const uint8_t* pm = img + (size_t)(y0 + i - 1) * pitch + x0;
        const uint8_t* p0 = pm + pitch;
        const uint8_t* pp = p0 + pitch;
#pragma unroll
        for (int j = -3; j <= 3; j++) {
            int Ix = ((int)p0[j + 1] - (int)p0[j - 1]) * 2 + ((int)pm[j + 1] - (int)pm[j - 1]) +
                     ((int)pp[j + 1] - (int)pp[j - 1]);
            int Iy = ((int)pp[j] - (int)pm[j]) * 2 + ((int)pp[j - 1] - (int)pm[j - 1]) + ((int)pp[j + 1] - (int)pm[j + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            cc += Ix * Iy;
        }
    }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    float fa = (float)a, fb = (float)b, fc = (float)cc;
    float t1 = fa * fb;
    float t2 = fc * fc;
    float s = fa + fb;
    float t3 = 0.04f * s;
    float t4 = t3 * s;
    return ((t1 - t2) - t4) * scale_sq_sq;
}

// Record window layout (LDS, SEL_BUF_BYTES): [records | rpos u16 x n/2 | ballots u64 x (n/64 + 1)]; the same layout is
// used inside the level's HBM scratch slot when a level has more candidates than the window holds.
__device__ __forceinline__ size_t sel_need_bytes(int n, int rec_bytes) {
    return (((size_t)n * rec_bytes + 7) & ~(size_t)7) + ((((size_t)n / 2 + 1) * 2 + 7) & ~(size_t)7) + ((size_t)n / 64 + 8) * 8;
}

#define HG 8                       // lanes per keypoint in the Harris phase
// threads of a k_select workgroup (the replay's partition passes, the Harris groups and the strip gather are dealt over its wavefronts):
// 256 in the batched mode (3 workgroups per CU; 512 / 1024 threads: 0.192 -> 0.240 / 0.418 ms per 256 frames), 1024 for one or two
// frames, where the chip is empty and the call waits for the finest levels' workgroups (91 -> 73 us; profiles/r04_ab_select_threads.txt)
#define SEL_THREADS 256
#define SEL_THREADS_LATENCY 1024

// Harris response of one keypoint by a group of HG = 8 lanes (same integer sums as harris_response, hence the same float):
// the 9x9 window is fetched as 9 rows x 3 aligned dwords (27 loads dealt over the 8 lanes: 12 cache-line accesses per keypoint;
// one lane per keypoint reading single bytes touched 81 lines with 64 different keypoints per load instruction and made this
// phase the longest of the selection), parked in a group-private LDS tile, and lane g < 7 takes column g - 3 of the 7x7 block
// with the separable forms Ix = d[r-1] + 2 d[r] + d[r+1], d = right - left, and Iy = s[r+1] - s[r-1], s = left + 2 centre + right.
__device__ __forceinline__ void harris_fetch(const uint8_t* img, int pitch, int x0, int y0, int g, uint32_t (&reg)[4]) {
    const uint8_t* base = img + (size_t)(y0 - 4) * pitch + ((x0 - 4) & ~3);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = min(g + HG * k, 26), r = q / 3, dw = q - 3 * r;  // (dwords 27 .. 31 of the deal re-read the last one)
        reg[k] = *(const uint32_t*)(base + r * pitch + 4 * dw);
    }
}
__device__ __forceinline__ float harris_group(const uint32_t (&reg)[4], int x0, uint32_t* tile /* [27] */, int g) {
    const int off = (x0 - 4) & 3;  // bytes off .. off + 8 of the 12 fetched per row
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (g + HG * k < 27) tile[g + HG * k] = reg[k];
    replay::wave_sync();  // the group's lanes sit in one wavefront: its LDS writes are visible to all of them
    int a = 0, b = 0, c = 0;
    if (g < 7) {
        const uint8_t* col = (const uint8_t*)tile + off + 1 + g;  // centre column of this lane: window column 4 + (g - 3)
        int d[9], sm[9];
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const int l = col[12 * r - 1], m = col[12 * r], rt = col[12 * r + 1];
            d[r] = rt - l;
            sm[r] = l + 2 * m + rt;
        }
#pragma unroll
        for (int r = 1; r < 8; r++) {
            const int Ix = d[r - 1] + 2 * d[r] + d[r + 1], Iy = sm[r + 1] - sm[r - 1];
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    }
#pragma unroll
    for (int o = 1; o < HG; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
    replay::wave_sync();  // the tile may be overwritten by the group's next keypoint
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    float fa = (float)a, fb = (float)b, fc = (float)c;
    float t1 = fa * fb;
    float t2 = fc * fc;
    float sx = fa + fb;
    float t3 = 0.04f * sx;
    float t4 = t3 * sx;
    return ((t1 - t2) - t4) * scale_sq_sq;
}

// phase 2 of the selection: Harris on the pass-1 survivors (in their pass-1 order), retainBest(quota) replay, write-out;
// all by the whole workgroup
template <int NT, class PA, class PB>
__device__ __forceinline__ void select_harris(const Plan& P, const LevelInfo& lv, const uint8_t* img, PA A, PB B, int N1,
                                              uint16_t* rpos, unsigned long long* bl, FinalKp* fin, int* fin_cnt_out,
                                              int* flags, replay::WgScratch<NT>* ws, uint32_t (*s_hw)[27] /* [NT / HG] group-private windows */, int level) {
    const int tid = threadIdx.x;
    constexpr int HG_GROUPS = NT / HG;
    if ((lv.pitch & 3) == 0 && (((size_t)img) & 3) == 0) {  // block-uniform; a keypoint sits >= edge_threshold >= 4 columns inside its row
        const int grp = tid / HG, g = tid % HG;
        // HB keypoints per group and trip: all their window loads are issued before the first one is reduced - the phase is a
        // chain of global-load latencies (1.5 us per keypoint when taken one at a time), not of work
#ifndef SEL_HB
#define SEL_HB 4
#endif
        constexpr int HB = SEL_HB;
        for (int i0 = 0; i0 < N1; i0 += HG_GROUPS * HB) {  // block-uniform trip count (the shuffles want whole wavefronts)
            uint32_t e[HB], reg[HB][4];
#pragma unroll
            for (int u = 0; u < HB; u++) {
                e[u] = A[min(i0 + u * HG_GROUPS + grp, N1 - 1)];
                harris_fetch(img, lv.pitch, e[u] & 0xFFF, (e[u] >> 12) & 0xFFF, g, reg[u]);
            }
#pragma unroll
            for (int u = 0; u < HB; u++) {
                const int i = i0 + u * HG_GROUPS + grp;
                const float r = harris_group(reg[u], e[u] & 0xFFF, s_hw[grp], g);
                if (g == 0 && i < N1) B[i] = ((uint64_t)__float_as_uint(r) << 32) | (e[u] & 0xFFFFFFu);
            }
        }
    } else {
        for (int i = tid; i < N1; i += NT) {
            uint32_t e = A[i];
            int x = e & 0xFFF, y = (e >> 12) & 0xFFF;
            float r = harris_response(img, lv.pitch, x, y);
            B[i] = ((uint64_t)__float_as_uint(r) << 32) | (e & 0xFFFFFFu);
        }
    }
    __syncthreads();
    int N2 = replay::wg_retain_best<NT, uint64_t>(B, N1, lv.quota, P.select_order, rpos, bl, tid, ws);
    if (N2 > lv.fin_cap) {
        if (tid == 0) { atomicOr(&flags[0], 1); atomicOr(&flags[1], 1 << level); }  // (word 1 names the level: only its slot grows)
        N2 = lv.fin_cap;
    }
    for (int i = tid; i < N2; i += NT) {
        uint64_t e = B[i];
        FinalKp k;
        k.x = (uint16_t)(e & 0xFFF);
        k.y = (uint16_t)((e >> 12) & 0xFFF);
        k.response = __uint_as_float((uint32_t)(e >> 32));
        fin[i] = k;
    }
    if (tid == 0) *fin_cnt_out = N2;
}

// One workgroup (NT / 64 wavefronts) per (frame, level).  Gather, Harris, write-out and the partition passes of the two
// retainBest replays use all wavefronts (workgroup-parallel pairing partition, select_replay.h).
template <int NT>
__global__ __launch_bounds__(NT) void k_select(Plan P, const uint8_t* __restrict__ gray, const uint8_t* __restrict__ pyr,
                                                        const uint32_t* __restrict__ cand, const int* __restrict__ strip_cnt,
                                                        uint64_t* __restrict__ scratch, size_t scratch_stride,
                                                        FinalKp* __restrict__ fin_all, int* __restrict__ fin_cnt, int* flags,
                                                        int level0, int buf_bytes, int* __restrict__ desc_todo) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_buf[];  // buf_bytes (per launch: coarse levels get less) + the strip prefix table
    int* const s_pref = (int*)(s_buf + buf_bytes);  // [max strips of a level + 1] (sized by the launch: 1 KB at 640 x 480; a static 4 KB
                                                    // table for 4K frames cost every launch a workgroup per CU: 0.218 -> 0.303 ms)
    __shared__ replay::WgScratch<NT> s_ws;
    __shared__ uint32_t s_hw[NT / HG][27];          // Harris windows, one per 8-lane group
    const int L = level0 + blockIdx.y, frame = blockIdx.x, tid = threadIdx.x;  // dispatch order: all frames of the finest level first
    if (desc_todo && frame == 0 && blockIdx.y == 0 && tid == 0) desc_todo[0] = 0;  // k_describe_tiles' list of left-over tiles (this call's)
    const LevelInfo lv = P.lv[L];
    int* fin_cnt_out = &fin_cnt[(size_t)frame * MO_MAX_LEVELS + L];
    if (lv.nstrips == 0) {
        if (tid == 0) *fin_cnt_out = 0;
        return;
    }
    const int* cnts = strip_cnt + (size_t)frame * P.strips_per_frame + lv.strip_base;
    for (int s = tid; s < lv.nstrips; s += NT) s_pref[s + 1] = cnts[s];
    __syncthreads();
    if (tid == 0) {
        s_pref[0] = 0;
        for (int s = 0; s < lv.nstrips; s++) s_pref[s + 1] += s_pref[s];
    }
    __syncthreads();
    const int N = s_pref[lv.nstrips];
    // HBM scratch slot of this (frame, level): [B records u64 x cap | A records u32 x cap | rpos | ballots]
    uint64_t* scr = scratch + (size_t)frame * scratch_stride + lv.scr_off;
    uint64_t* gB = scr;
    uint32_t* gA = (uint32_t*)(scr + lv.cand_cap);
    uint16_t* g_rpos = (uint16_t*)(gA + lv.cand_cap + (lv.cand_cap & 1));
    unsigned long long* g_bl = (unsigned long long*)(g_rpos + (((size_t)lv.cand_cap / 2 + 4) & ~(size_t)3));
    const bool a_lds = sel_need_bytes(N, 4) <= (size_t)buf_bytes;
    uint32_t* s_A = (uint32_t*)s_buf;
    uint16_t* a_rpos = (uint16_t*)(s_buf + (((size_t)N * 4 + 7) & ~(size_t)7));
    unsigned long long* a_bl = (unsigned long long*)((uint8_t*)a_rpos + ((((size_t)N / 2 + 1) * 2 + 7) & ~(size_t)7));
    // gather the strip lists in raster order: one wavefront per strip, strips dealt round-robin to the 4 wavefronts
    const uint32_t* src = cand + (size_t)frame * P.cand_stride + lv.cand_off;
    {
        // four strips per wavefront and trip: their first 64 records each are fetched together (a strip list is a dependent
        // round trip, and a level has up to 60 of them: 15 per wavefront one after the other were 12 us of the level-0 chain)
        const int lane = tid & 63, wv = tid >> 6;
        constexpr int NW = NT / 64, GU = 4;
        for (int s0 = wv; s0 < lv.nstrips; s0 += NW * GU) {  // wave-uniform
            uint32_t v[GU];
            int b[GU], n[GU];
#pragma unroll
            for (int u = 0; u < GU; u++) {
                const int st = min(s0 + u * NW, lv.nstrips - 1);
                b[u] = s_pref[st];
                n[u] = s0 + u * NW < lv.nstrips ? s_pref[st + 1] - b[u] : 0;
                v[u] = lane < n[u] ? src[(size_t)st * lv.strip_cap + lane] : 0u;
            }
#pragma unroll
            for (int u = 0; u < GU; u++) {
                if (lane < n[u]) { if (a_lds) s_A[b[u] + lane] = v[u]; else gA[b[u] + lane] = v[u]; }
                if (n[u] > WAVE) {  // wave-uniform: the rest of a long strip list
                    const uint32_t* e = src + (size_t)(s0 + u * NW) * lv.strip_cap;
                    for (int i = WAVE + lane; i < n[u]; i += WAVE) { if (a_lds) s_A[b[u] + i] = e[i]; else gA[b[u] + i] = e[i]; }
                }
            }
        }
    }
    __syncthreads();
    // pass 1: retainBest(2 * quota) on the FAST score
    const int N1 = a_lds ? replay::wg_retain_best<NT, uint32_t>(s_A, N, 2 * lv.quota, P.select_order, a_rpos, a_bl, tid, &s_ws)
                         : replay::wg_retain_best<NT, uint32_t>(gA, N, 2 * lv.quota, P.select_order, g_rpos, g_bl, tid, &s_ws);
    // the Harris records (and their rpos / ballots) go behind the surviving FAST records when both fit the window
    const size_t b_off = a_lds ? (((size_t)N1 * 4 + 15) & ~(size_t)15) : 0;
    const bool b_lds = b_off + sel_need_bytes(N1, 8) <= (size_t)buf_bytes;
    uint64_t* s_B = (uint64_t*)(s_buf + b_off);
    uint16_t* b_rpos = (uint16_t*)((uint8_t*)s_B + (size_t)N1 * 8);
    unsigned long long* b_bl = (unsigned long long*)((uint8_t*)b_rpos + ((((size_t)N1 / 2 + 1) * 2 + 7) & ~(size_t)7));
    const uint8_t* img = level_ptr(P, L, gray, pyr, frame);
    FinalKp* fin = fin_all + (size_t)frame * P.fin_stride + lv.fin_off;
    if (a_lds && b_lds) select_harris<NT>(P, lv, img, s_A, s_B, N1, b_rpos, b_bl, fin, fin_cnt_out, flags, &s_ws, s_hw, L);
    else if (a_lds) select_harris<NT>(P, lv, img, s_A, gB, N1, g_rpos, g_bl, fin, fin_cnt_out, flags, &s_ws, s_hw, L);
    else if (b_lds) select_harris<NT>(P, lv, img, gA, s_B, N1, b_rpos, b_bl, fin, fin_cnt_out, flags, &s_ws, s_hw, L);
    else select_harris<NT>(P, lv, img, gA, gB, N1, g_rpos, g_bl, fin, fin_cnt_out, flags, &s_ws, s_hw, L);
}

int orb_launch_select(mo_ctx* c, const uint8_t* d_gray, int batch, int level_lo, int level_hi) {
    const Plan& P = c->plan;
    level_lo = std::max(level_lo, 0); level_hi = std::min(level_hi, P.nlevels);
    if (level_lo >= level_hi) return MO_OK;
    for (int L = 0; L < P.nlevels; L++)
        if (P.lv[L].nstrips > SEL_MAXSTRIPS) return mo_fail(c, MO_ERR_UNSUPPORTED, "too many strips per level");
    int max_strips = 1;
    for (int L = 0; L < P.nlevels; L++) max_strips = std::max(max_strips, P.lv[L].nstrips);
    const size_t pref_bytes = (((size_t)max_strips + 1) * sizeof(int) + 15) & ~(size_t)15;
    const size_t attr_bytes = SEL_BUF_BYTES + ((((size_t)SEL_MAXSTRIPS + 1) * sizeof(int) + 15) & ~(size_t)15);
    if (!(c->lds_attr_done & 16u)) {
        HIPCHK(c, hipFuncSetAttribute((const void*)k_select<SEL_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attr_bytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_select<SEL_THREADS_LATENCY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attr_bytes));
        c->lds_attr_done |= 16u;
    }
    // One launch, every workgroup with the full LDS record window (3 resident per CU at 256 threads), levels in dispatch order from fine
    // to coarse: the replays are latency-bound chains whose length grows with the candidate count, so the long level-0 tasks start first
    // and the short coarse-level tasks fill the slots that free up (longest-task-first packing).  A level that outgrows the window falls
    // back to its HBM scratch slot (same code, slower).  Round 2 re-measured the alternatives on MI355X: a second launch for the coarse
    // levels with a smaller window (8 - 24 KB, more workgroups per CU): 0.23 - 0.25 ms against 0.20 ms; selecting the finest level on the
    // auxiliary stream beside FAST of the others: no gain (the coarse levels alone take 0.19 ms: the kernel is bound by the sum of the
    // replays, not by the finest level).  One or two frames (the single-frame host calls): 16 wavefronts per workgroup.
    const dim3 grid(batch, level_hi - level_lo);
    if (batch <= 2)
        hipLaunchKernelGGL(k_select<SEL_THREADS_LATENCY>, grid, dim3(SEL_THREADS_LATENCY), SEL_BUF_BYTES + pref_bytes, c->stream, P, d_gray, c->d_pyr, c->d_cand,
                           c->d_strip_cnt, c->d_scratch, c->scratch_stride, c->d_fin, c->d_fin_cnt, c->flags_cur, level_lo, SEL_BUF_BYTES, c->d_dtodo);
    else
        hipLaunchKernelGGL(k_select<SEL_THREADS>, grid, dim3(SEL_THREADS), SEL_BUF_BYTES + pref_bytes, c->stream, P, d_gray, c->d_pyr, c->d_cand,
                           c->d_strip_cnt, c->d_scratch, c->scratch_stride, c->d_fin, c->d_fin_cnt, c->flags_cur, level_lo, SEL_BUF_BYTES, c->d_dtodo);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// probe for the parity tests: retainBest on a bare float response array (u64 record path, HBM-resident records)
__global__ __launch_bounds__(SEL_THREADS) void k_retain_probe(const float* resp, int n, int n_points, int order, uint64_t* rec,
                                                              uint16_t* rpos, unsigned long long* bl, int32_t* out, int* nout) {
    __shared__ replay::WgScratch<SEL_THREADS> s_ws;
    for (int i = threadIdx.x; i < n; i += SEL_THREADS) rec[i] = ((uint64_t)__float_as_uint(resp[i]) << 32) | (uint32_t)i;
    __syncthreads();
    int keep = replay::wg_retain_best<SEL_THREADS, uint64_t>(rec, n, n_points, order, rpos, bl, threadIdx.x, &s_ws);
    __syncthreads();
    for (int i = threadIdx.x; i < keep; i += SEL_THREADS) out[i] = (int32_t)(rec[i] & 0xFFFFFFFFu);
    if (threadIdx.x == 0) *nout = keep;
}

int orb_launch_retain_probe(mo_ctx* c, const float* d_resp, int n, int n_points, int order, int32_t* d_order, int* d_nout) {
    size_t nn = (size_t)std::max(n, 1);
    size_t rec_b = nn * sizeof(uint64_t), rpos_b = ((nn / 2 + 4) * 2 + 7) & ~(size_t)7, bl_b = (nn / 64 + 8) * 8;
    int rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, rec_b + rpos_b + bl_b);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)c->d_tmp;
    hipLaunchKernelGGL(k_retain_probe, dim3(1), dim3(SEL_THREADS), 0, c->stream, d_resp, n, n_points, order, (uint64_t*)b,
                       (uint16_t*)(b + rec_b), (unsigned long long*)(b + rec_b + rpos_b), d_order, d_nout);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// ------------------------------------------------------------------ describe ------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 256 rotated binary tests around (cx, cy): lane l does tests 4l..4l+3; even lanes store one byte.
// BOUNDS: sample positions may leave the level (compute() with caller keypoints on coarse octaves);
// OpenCV then reads its REFLECT_101 apron, which holds UNBLURRED pixels.
template <bool BOUNDS>
__device__ __forceinline__ void rbrief_wave(const uint8_t* blur, int bpitch, const uint8_t* raw, int rpitch, int lw, int lh,
                                            int cx, int cy, float angle_deg, uint8_t* desc, int lane) {
    float angle = angle_deg;
    angle *= (float)(3.14159265358979323846 / 180.f);
    double sd, cd;
    sincos((double)angle, &sd, &cd);  // f64 then rounded to f32, as cv2's (float)cos(angle) / (float)sin(angle)
    float a = (float)cd, b = (float)sd;
    unsigned nib = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int8_t* pt = &c_pattern[(lane * 4 + k) * 4];
        float fx0 = (float)pt[0], fy0 = (float)pt[1], fx1 = (float)pt[2], fy1 = (float)pt[3];
        int ix0 = __float2int_rn(fx0 * a - fy0 * b), iy0 = __float2int_rn(fx0 * b + fy0 * a);
        int ix1 = __float2int_rn(fx1 * a - fy1 * b), iy1 = __float2int_rn(fx1 * b + fy1 * a);
        int x0 = cx + ix0, y0 = cy + iy0, x1 = cx + ix1, y1 = cy + iy1;
        int t0, t1;
        if (BOUNDS) {
            t0 = (x0 >= 0 && x0 < lw && y0 >= 0 && y0 < lh) ? blur[(size_t)y0 * bpitch + x0]
                                                           : raw[(size_t)reflect101(y0, lh) * rpitch + reflect101(x0, lw)];
            t1 = (x1 >= 0 && x1 < lw && y1 >= 0 && y1 < lh) ? blur[(size_t)y1 * bpitch + x1]
                                                           : raw[(size_t)reflect101(y1, lh) * rpitch + reflect101(x1, lw)];
        } else {
            t0 = blur[(size_t)y0 * bpitch + x0];
            t1 = blur[(size_t)y1 * bpitch + x1];
        }
        nib |= (t0 < t1 ? 1u : 0u) << k;
    }
    unsigned hi = __shfl_down(nib, 1, 64);
    if ((lane & 1) == 0) desc[lane >> 1] = (uint8_t)(nib | (hi << 4));
}

#define DG 16                 // lanes per keypoint (a quarter of a wavefront)

__device__ __forceinline__ int group_sum(int v) {  // sum over the DG lanes of a keypoint group
#pragma unroll
    for (int o = DG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 256 rotated binary tests out of an LDS tile, 16 per lane of a 16-lane group (lane gl produces descriptor bytes 2 gl, 2 gl + 1).  pat: the pattern as floats in LDS, test-major interleaved so that the 16 lanes
// of a group read 16 consecutive float4 rows per test (conflict-free ds_read_b128, the four groups of a wavefront read the same rows:
// broadcast) - no int8 -> float conversion per keypoint (64 half-rate v_cvt per lane before).  lds_base: LDS byte address of the
// sample (cx, cy) of the blurred tile; the sample address is ONE v_mad_i32_i24 per point: the rounding bias 0x4B400000 of jx and its
// low 24 bits 0x400000 of jy (x pitch) are folded into the base.  A test bit is the sign of t0 - t1 shifted in by v_alignbit.
typedef const __attribute__((address_space(3))) uint8_t lds_cu8;
__device__ __forceinline__ uint16_t rbrief_u16_lds(uint32_t lds_base, int ppitch, float a, float b, const float4* pat, int gl) {
    const uint32_t base = lds_base - 0x4B400000u - 0x400000u * (uint32_t)ppitch;
    uint32_t val = 0;
#ifndef RB_UNROLL
#define RB_UNROLL 16
#endif
#pragma unroll 1
    for (int k0 = 16 - RB_UNROLL; k0 >= 0; k0 -= RB_UNROLL) {  // RB_UNROLL tests per trip (16: fully unrolled; the register peak of the kernel lies elsewhere)
#pragma unroll
        for (int kk = RB_UNROLL - 1; kk >= 0; kk--) {  // test 0 is shifted in last: bit 0
            const float4 pt = pat[(k0 + kk) * DG + gl];  // (x0, y0, x1, y1) of test 16 gl + k
            const int jx0 = __float_as_int((pt.x * a - pt.y * b) + 12582912.f), jy0 = __float_as_int((pt.x * b + pt.y * a) + 12582912.f);
            const int jx1 = __float_as_int((pt.z * a - pt.w * b) + 12582912.f), jy1 = __float_as_int((pt.z * b + pt.w * a) + 12582912.f);
            const uint32_t a0 = (uint32_t)__mul24(jy0, ppitch) + (base + (uint32_t)jx0), a1 = (uint32_t)__mul24(jy1, ppitch) + (base + (uint32_t)jx1);
            const int t0 = *(lds_cu8*)(uintptr_t)a0, t1 = *(lds_cu8*)(uintptr_t)a1;
            val = __builtin_amdgcn_alignbit(val, (uint32_t)(t0 - t1), 31);  // (val << 1) | (t0 < t1)
        }
    }
    return (uint16_t)val;
}

// ------------------------------------------------------------------ describe, tile form (round 3) -------------------
// The per-keypoint kernel of rounds 1 - 2 (k_describe, removed; `git show 8789e68:visual-slam_amd/csrc/orb_kernels.hip`) gathered two private windows per keypoint from global memory (31 x 36 B raw + 39 x 44 B blurred in
// 48 dword loads per lane: ~130 cache-line requests per keypoint, 1.45 GB through the vector L1 per 256-frame launch, a wavefront
// spent 8 - 21 k of its 30 k cycles waiting for them) and parks them in LDS through registers (93 VGPRs, 5 wavefronts per SIMD).
// Here a workgroup owns a 128 x 64 TILE of one level's border region: it picks the level's final keypoints that fall into the tile
// (the level's list is <= a few KB, read coalesced), loads the tile + halo of the raw level (15 px: intensity centroid) and of the
// blurred level (19 px: rBRIEF) ONCE with coalesced row loads, and describes its keypoints out of LDS, 16 lanes each:
//   * line requests per frame: 214 tiles x ~390 instead of 2000 keypoints x ~130 (levels 2.. are dense: 20 - 60 keypoints per tile)
//   * no window registers: the per-row intensity-centroid weights (constant per lane) live in registers instead of being re-read
//     from LDS for every keypoint, ~70 VGPRs
//   * tile pitches of 25 / 27 dwords (odd): the 16 lanes of a group read 16 different rows conflict-free in the centroid phase
// Results are those of that kernel bit for bit (same integer sums, same float expressions, same sample addresses).
#ifndef DT_W
#define DT_W 128   // measured on MI355X (profiles/r03_ab_describe.txt): 64 x 64 tiles of 128 threads 0.334 ms, 128 x 64 tiles of 256 threads 0.273 ms
#endif
#define DT_H 64
#define DT_RAW_P ((DT_W + 30 + 7 + 7) & ~7)  // raw tile pitch: DT_W + 30 columns + <= 7 alignment lead-in, 8-byte pieces (104: 26 dwords, the 16
                                             // rows the lanes of a group read in the centroid phase fall into 16 different banks)
#define DT_RAW_ROWS (DT_H + 30)
#define DT_BLR_P ((DT_W + 38 + 7 + 7) & ~7)  // blurred tile pitch: DT_W + 38 columns + <= 7 lead-in (112)
#define DT_BLR_ROWS (DT_H + 38)
#ifndef DT_NT
#define DT_NT 256                    // 16 keypoint groups of 16 lanes
#ifndef DT_SPLIT_LATENCY
#define DT_SPLIT_LATENCY 4           // workgroups per tile in calls on one or two frames (a power of two)
#endif
#endif
#ifdef DT_WAVES
#define DT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(DT_WAVES, 8)))
#else
#define DT_WAVES_ATTR
#endif
#define DT_CHUNK 512                 // entries of the level's list examined per pass (a 2000-feature level 0 holds 434)
#define DT_LIST (DT_NT > 160 ? DT_NT : 160)  // keypoints of the tile described per pass (typical: 6 on level 0, 60 on level 7; a sub-pass examines DT_NT records); 7 workgroups per CU
#define DT_RAW_LD ((DT_RAW_ROWS * (DT_RAW_P / 8) + DT_NT - 1) / DT_NT)   // 8-byte loads per thread: 10
#define DT_BLR_LD ((DT_BLR_ROWS * (DT_BLR_P / 8) + DT_NT - 1) / DT_NT)   // 12

// rows [ry0, ry0 + NROWS) x 8-byte columns [cx_al, cx_al + 8 NQ) of a level -> registers (tile_issue), then LDS (tile_store): every load
// of both tiles is in flight before the first store - one global round trip per workgroup.  Rows are clamped into the level,
// 8-byte columns into the row pitch (clamped data is never sampled: a keypoint keeps edge_threshold >= 19 from the border).
template <int NROWS, int NQ, int NLD>
__device__ __forceinline__ void tile_issue(const uint8_t* img, int pitch, int h, int ry0, int cx_al, int tid, uint2 (&v)[NLD]) {
    const int last_q = (pitch >> 3) - 1;
#pragma unroll
    for (int u = 0; u < NLD; u++) {
        const int t = min(tid + u * DT_NT, NROWS * NQ - 1), r = t / NQ, c = t - r * NQ;  // (constant divisor)
        const int y = min(max(ry0 + r, 0), h - 1), q = min(max((cx_al >> 3) + c, 0), last_q);
        v[u] = ((const uint2*)(img + (size_t)y * pitch))[q];
    }
}
template <int NROWS, int NQ, int NLD>
__device__ __forceinline__ void tile_store(uint8_t* lds, int tid, const uint2 (&v)[NLD]) {
#pragma unroll
    for (int u = 0; u < NLD; u++) {
        const int t = tid + u * DT_NT, r = t / NQ, c = t - r * NQ;
        if (t < NROWS * NQ) *(uint2*)(lds + r * (NQ * 8) + 8 * c) = v[u];
    }
}
// rows that are not 8-byte aligned (a caller image whose width is not a multiple of 8): bytes
__device__ __forceinline__ void tile_load_bytes(const uint8_t* img, int pitch, int w, int h, int ry0, int nrows, int cx0, int lp, uint8_t* lds, int tid) {
    for (int i = tid; i < nrows * lp; i += DT_NT) {
        const int r = i / lp, c = i - r * lp;
        const int y = min(max(ry0 + r, 0), h - 1), x = min(max(cx0 + c, 0), w - 1);
        lds[i] = img[(size_t)y * pitch + x];
    }
}

// RARE = false: the common case in ONE pass (the level keeps <= DT_CHUNK keypoints and the tile <= DT_LIST of them); a tile that needs
// more appends itself to `todo` and is redone by k_describe_tiles_rare (RARE = true: chunks of DT_CHUNK records, DT_CHUNK / DT_NT
// passes each, both tiles fetched again for every pass).  Two kernels because the generic loop beside the one-pass form costs the
// latter 34 registers (125 instead of 91: 4 instead of 5 wavefronts per SIMD).
template <bool HAS_DESC, bool RARE>
__device__ __forceinline__ void describe_tile(const Plan& P, int frame, int tile, int tiles_per_frame, int split, int nsplit /* power of two */,
                                              const uint32_t* __restrict__ tile_tab,
                                              const uint8_t* __restrict__ gray, const uint8_t* __restrict__ pyr,
                                              const uint8_t* __restrict__ blur, const FinalKp* __restrict__ fin_all,
                                              const int* __restrict__ fin_cnt, mo_keypoint* __restrict__ kps,
                                              uint8_t* __restrict__ desc, int cap, int* __restrict__ counts, int* flags,
                                              const uint32_t* __restrict__ icw, int* __restrict__ todo) {
    // ONE tile buffer: the raw tile (intensity centroid, phase A) is replaced by the blurred tile (rBRIEF, phase C) once phase A is
    // done; the blurred tile's loads are in flight since the prologue and wait in registers (18 per thread)
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[HAS_DESC ? DT_BLR_ROWS * DT_BLR_P : DT_RAW_ROWS * DT_RAW_P];
    __shared__ __attribute__((aligned(16))) float4 s_pat[HAS_DESC ? 256 : 1];  // rBRIEF pattern as floats, [test k of a lane][lane]
    __shared__ uint32_t s_list[DT_LIST];  // index in the chunk | dx << 9 | dy << 17
    __shared__ float2 s_ab[DT_LIST];      // (-, response) -> (angle, -) -> (cos, sin)
    __shared__ int s_n;
    static_assert(DT_BLR_ROWS * DT_BLR_P >= DT_RAW_ROWS * DT_RAW_P, "the blurred tile is the larger one");
    const uint32_t te = tile_tab[tile];
    const int L = te & 0xFF;
    const int x0 = P.lv[L].bx0 + (int)((te >> 8) & 0xFFF) * DT_W, y0 = P.lv[L].by0 + (int)(te >> 20) * DT_H;
    const int lpitch = P.lv[L].pitch, lbpitch = P.lv[L].bpitch, lw = P.lv[L].w, lh = P.lv[L].h;
    const float lscale = P.lv[L].scale;
    const int tid = threadIdx.x, grp = tid / DG, gl = tid % DG;
    // per-level counts of the frame (wave-uniform, issued together): this level's list length and its first output row
    const int* fc = fin_cnt + (size_t)frame * MO_MAX_LEVELS;
    int total = 0, base = 0, nL = 0;
    {
        int cn[MO_MAX_LEVELS];
#pragma unroll
        for (int l = 0; l < MO_MAX_LEVELS; l++) cn[l] = fc[l];
#pragma unroll
        for (int l = 0; l < MO_MAX_LEVELS; l++) {
            const int n = l < P.nlevels ? cn[l] : 0;
            if (l == L) { base = total; nL = n; }
            total += n;
        }
    }
    if (!RARE && tile == 0 && split == 0 && tid == 0) {
        counts[frame] = total;
        if (total > cap) atomicOr(&flags[0], 2);
    }
    nL = min(nL, max(cap - base, 0));  // rows past cap are not produced
    if (nL == 0) return;               // block-uniform
    const uint8_t* img = level_ptr(P, L, gray, pyr, frame);
    const uint8_t* bl = blur + (size_t)frame * P.blur_stride + P.lv[L].boff;
    const bool al_raw = (lpitch & 7) == 0 && (((size_t)img) & 7) == 0;
    const bool al_blr = (((size_t)bl) & 7) == 0;  // (blurred rows have a pitch of a multiple of 16)
    const int xr_al = al_raw ? (x0 - 15) & ~7 : x0 - 15, lead_r = (x0 - 15) - xr_al;   // raw tile column 0 = level column xr_al
    const int xb_al = al_blr ? (x0 - 19) & ~7 : x0 - 19;
    const FinalKp* fin = fin_all + (size_t)frame * P.fin_stride + P.lv[L].fin_off;
    // ---- the prologue's global reads, in flight together: the first chunk of the level's list, the raw tile (empty tiles are rare: it is
    //      loaded unconditionally), the lane's centroid weights, the pattern
    FinalKp fk[DT_CHUNK / DT_NT];
    uint2 vr[DT_RAW_LD], vb[DT_BLR_LD];  // (vb stays unused, and is dropped by the compiler, without descriptors)
    if (!RARE) {
#pragma unroll
        for (int u = 0; u < DT_CHUNK / DT_NT; u++) fk[u] = fin[min(tid + u * DT_NT, nL - 1)];
        if (al_raw) tile_issue<DT_RAW_ROWS, DT_RAW_P / 8, DT_RAW_LD>(img, lpitch, lh, y0 - 15, xr_al, tid, vr);
    }
    // the two disc rows of this lane: rows gl and 30 - gl of the 31 (lane 15: row 15 once) have the same half-width, hence the same
    // weight bytes (u + 16 inside the disc) and mask bytes (1 inside): 16 registers, fetched once per workgroup
    uint32_t wt[8], mk[8];
    {
        const uint4* w = (const uint4*)(icw + gl * 16);
        const uint4 w0 = w[0], w1 = w[1], k0 = w[2], k1 = w[3];
        wt[0] = w0.x; wt[1] = w0.y; wt[2] = w0.z; wt[3] = w0.w; wt[4] = w1.x; wt[5] = w1.y; wt[6] = w1.z; wt[7] = w1.w;
        mk[0] = k0.x; mk[1] = k0.y; mk[2] = k0.z; mk[3] = k0.w; mk[4] = k1.x; mk[5] = k1.y; mk[6] = k1.z; mk[7] = k1.w;
    }
    if (HAS_DESC && tid < 256 && DT_NT >= 256) {  // thread t = lane l, test k of the lane: pattern row 16 l + k -> s_pat[k][l]
        const int l = tid & 15, k = tid >> 4;
        const int8_t* pt = &c_pattern[(l * 16 + k) * 4];
        s_pat[k * DG + l] = make_float4((float)pt[0], (float)pt[1], (float)pt[2], (float)pt[3]);
    } else if (HAS_DESC && DT_NT < 256) {
        for (int i = tid; i < 256; i += DT_NT) {
            const int l = i & 15, k = i >> 4;
            const int8_t* pt = &c_pattern[(l * 16 + k) * 4];
            s_pat[k * DG + l] = make_float4((float)pt[0], (float)pt[1], (float)pt[2], (float)pt[3]);
        }
    }
    const int row_b_w = gl == 15 ? 0 : 1;  // lane 15's second row is row 15 again: counted once
    const uint32_t tile_lds = (uint32_t)(uintptr_t)s_tile;

    // the chunk's records of this thread that fall into the tile -> list (sub >= 0: only record slot `sub` of the thread).  When nsplit
    // workgroups share the tile (calls on one or two frames), workgroup `split` lists the records whose index in the LEVEL's list is in its
    // residue class - a property every workgroup computes alike; the position in s_list is not one (atomicAdd order)
    auto select = [&](int c0, int sub) {
#pragma unroll
        for (int u = 0; u < DT_CHUNK / DT_NT; u++) {
            const int i = c0 + tid + u * DT_NT;
            const int dx = (int)fk[u].x - x0, dy = (int)fk[u].y - y0;
            if ((sub < 0 || u == sub) && i < nL && (unsigned)dx < DT_W && (unsigned)dy < DT_H && (i & (nsplit - 1)) == split) {
                const int pos = atomicAdd(&s_n, 1);
                if (pos < DT_LIST) {
                    s_list[pos] = (uint32_t)(i - c0) | ((uint32_t)dx << 9) | ((uint32_t)dy << 17);
                    s_ab[pos].y = fk[u].response;
                }
            }
        }
    };
    // phase A, 16 lanes per keypoint: intensity-centroid angle out of the RAW tile, keypoint record; the angle goes into the list
    auto phase_a = [&](int c0, int n) {
        for (int j = grp; j < n; j += DT_NT / DG) {  // uniform within a 16-lane group; no barriers inside
            const uint32_t e = s_list[j];
            const int i = c0 + (int)(e & 0x1FF), dx = (int)((e >> 9) & 255), dy = (int)(e >> 17);
            const int k = base + i, x = x0 + dx, y = y0 + dy;
            // rows gl and 30 - gl of the radius-15 disc on each lane.  A row is read as nine dwords, realigned so that dword c holds
            // u = 4c - 15 .. 4c - 12, and reduced with two v_dot4_u32_u8 per dword against the row's weight / mask bytes:
            // sum u p = dot(w) - 16 dot(mask)
            const int col = dx + lead_r, offr = col & 3;
            int m10 = 0, m01 = 0;
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int row_i = rr == 0 ? gl : 30 - gl;
                const uint32_t* rowp = (const uint32_t*)(s_tile + (dy + row_i) * DT_RAW_P + (col & ~3));
                uint32_t pxw[9];
#pragma unroll
                for (int c4 = 0; c4 < 9; c4++) pxw[c4] = rowp[c4];
                uint32_t sw = 0, rs = 0;
#pragma unroll
                for (int c4 = 0; c4 < 8; c4++) {
                    const uint32_t a = __builtin_amdgcn_alignbyte(pxw[c4 + 1], pxw[c4], (uint32_t)offr);
                    sw = __builtin_amdgcn_udot4(a, wt[c4], sw, false);
                    rs = __builtin_amdgcn_udot4(a, mk[c4], rs, false);
                }
                const int wgt = rr == 0 ? 1 : row_b_w;
                m10 += wgt * ((int)sw - 16 * (int)rs);
                m01 += wgt * (row_i - 15) * (int)rs;
            }
            m10 = group_sum(m10);
            m01 = group_sum(m01);
            const float angle = fast_atan2_deg((float)m01, (float)m10);
            if (gl == 0) {
                mo_keypoint* o = kps + (size_t)frame * cap + k;
                o->x = (float)x * lscale; o->y = (float)y * lscale;
                o->size = 31 * lscale;
                o->angle = angle;
                o->response = s_ab[j].y;
                o->octave = L;
                o->class_id = -1;
                s_ab[j].x = angle;
            }
        }
    };
    // phase B, ONE lane per keypoint: (float)cos / (float)sin of the angle through f64 as cv2 computes them - once per keypoint instead
    // of once per lane of its group (the f64 sincos is the longest straight-line piece of the kernel)
    auto phase_b = [&](int n) {
        for (int j = tid; j < n; j += DT_NT) {
            float angle = s_ab[j].x;
            angle *= (float)(3.14159265358979323846 / 180.f);
            double sd, cd;
            sincos((double)angle, &sd, &cd);
            s_ab[j] = make_float2((float)cd, (float)sd);
        }
    };
    // phase C, 16 lanes per keypoint: the 256 rotated tests, 16 per lane, out of the BLURRED tile
    auto phase_c = [&](int c0, int n) {
        for (int j = grp; j < n; j += DT_NT / DG) {
            const uint32_t ex = s_list[j];
            const int i = c0 + (int)(ex & 0x1FF), dx = (int)((ex >> 9) & 255), dy = (int)(ex >> 17);
            const int k = base + i;
            const float px = (float)(x0 + dx) * lscale, py = (float)(y0 + dy) * lscale, inv = 1.f / lscale;
            const int cx = __float2int_rn(px * inv), cy = __float2int_rn(py * inv);
            const float2 ab = s_ab[j];
            int glo = gl;
            asm volatile("" : "+v"(glo));  // opaque per keypoint: the 16 pattern reads stay in the loop (hoisted, they pin 64 registers)
            *(uint16_t*)(desc + ((size_t)frame * cap + k) * 32 + 2 * gl) =
                rbrief_u16_lds(tile_lds + (uint32_t)((cy - (y0 - 19)) * DT_BLR_P + (cx - xb_al)), DT_BLR_P, ab.x, ab.y, s_pat, glo);
        }
    };

    if (!RARE) {
        // ---- the common case in one pass: raw tile -> LDS, list, phase A, blurred tile over the raw one, phases B and C
        if (tid == 0) s_n = 0;
        if (al_raw) tile_store<DT_RAW_ROWS, DT_RAW_P / 8, DT_RAW_LD>(s_tile, tid, vr);
        else tile_load_bytes(img, lpitch, lw, lh, y0 - 15, DT_RAW_ROWS, xr_al, DT_RAW_P, s_tile, tid);
        __syncthreads();
        select(0, -1);
        __syncthreads();
        const int n0 = s_n;
        if (nL > DT_CHUNK || n0 > DT_LIST) {  // block-uniform: left to k_describe_tiles_rare
            // (a long level list is seen by all sharing workgroups alike: one entry; an overflow of its own list by the workgroup that has it)
            if (tid == 0 && (split == 0 || n0 > DT_LIST)) todo[1 + atomicAdd(&todo[0], 1)] = frame * tiles_per_frame + tile;
            return;
        }
        phase_a(0, n0);
        if (HAS_DESC) {
            // the blurred tile's loads are issued HERE, not in the prologue (held in registers through phase A they cost 18 VGPRs and the
            // sixth wavefront per SIMD: 91 -> 73); they are in flight during the barrier and the sincos phase
            if (al_blr) tile_issue<DT_BLR_ROWS, DT_BLR_P / 8, DT_BLR_LD>(bl, lbpitch, lh, y0 - 19, xb_al, tid, vb);
            __syncthreads();  // phase A is done with the raw tile, the angles are in the list
            phase_b(n0);
            if (al_blr) tile_store<DT_BLR_ROWS, DT_BLR_P / 8, DT_BLR_LD>(s_tile, tid, vb);
            else tile_load_bytes(bl, lbpitch, lw, lh, y0 - 19, DT_BLR_ROWS, xb_al, DT_BLR_P, s_tile, tid);
            __syncthreads();
            phase_c(0, n0);
        }
        return;
    }
    // ---- the rare case: chunks of DT_CHUNK records, each as DT_CHUNK / DT_NT passes over DT_NT records
    for (int c0 = 0; c0 < nL; c0 += DT_CHUNK) {
#pragma unroll
        for (int u = 0; u < DT_CHUNK / DT_NT; u++) fk[u] = fin[min(c0 + tid + u * DT_NT, nL - 1)];
#pragma unroll 1
        for (int sub = 0; sub < DT_CHUNK / DT_NT; sub++) {
            __syncthreads();  // the previous pass is done with the list and the tile
            if (tid == 0) s_n = 0;
            tile_load_bytes(img, lpitch, lw, lh, y0 - 15, DT_RAW_ROWS, xr_al, DT_RAW_P, s_tile, tid);
            __syncthreads();
            select(c0, sub);
            __syncthreads();
            const int n = s_n;  // <= DT_NT <= DT_LIST
            phase_a(c0, n);
            if (HAS_DESC) {
                __syncthreads();
                tile_load_bytes(bl, lbpitch, lw, lh, y0 - 19, DT_BLR_ROWS, xb_al, DT_BLR_P, s_tile, tid);
                phase_b(n);
                __syncthreads();
                phase_c(c0, n);
            }
        }
    }
    __syncthreads();  // (the next todo entry of this workgroup reuses the tile and the list)
}

template <bool HAS_DESC>
__global__ __launch_bounds__(DT_NT) DT_WAVES_ATTR void k_describe_tiles(Plan P, const uint32_t* __restrict__ tile_tab, uint32_t inv_per,
                                                          const uint8_t* __restrict__ gray, const uint8_t* __restrict__ pyr,
                                                          const uint8_t* __restrict__ blur, const FinalKp* __restrict__ fin_all,
                                                          const int* __restrict__ fin_cnt, mo_keypoint* __restrict__ kps,
                                                          uint8_t* __restrict__ desc, int cap, int* __restrict__ counts, int* flags,
                                                          const uint32_t* __restrict__ icw, int* __restrict__ todo) {
    int frame = blockIdx.y, tile = blockIdx.x;
    xcd_map(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, inv_per, gridDim.y, frame, tile);  // XCD affinity (speed only)
    describe_tile<HAS_DESC, false>(P, frame, tile, (int)gridDim.x, (int)blockIdx.z, (int)gridDim.z, tile_tab, gray, pyr, blur, fin_all, fin_cnt, kps, desc, cap, counts, flags, icw, todo);
}

// the tiles the one-pass kernel left over (todo[0] of them, usually none: the workgroups then leave at once); todo[0] is cleared by the
// next call's k_select
template <bool HAS_DESC>
__global__ __launch_bounds__(DT_NT) void k_describe_tiles_rare(Plan P, const uint32_t* __restrict__ tile_tab, int tiles_per_frame,
                                                               const uint8_t* __restrict__ gray, const uint8_t* __restrict__ pyr,
                                                               const uint8_t* __restrict__ blur, const FinalKp* __restrict__ fin_all,
                                                               const int* __restrict__ fin_cnt, mo_keypoint* __restrict__ kps,
                                                               uint8_t* __restrict__ desc, int cap, int* __restrict__ counts, int* flags,
                                                               const uint32_t* __restrict__ icw, int* __restrict__ todo) {
    const int n = todo[0];
    for (int e = blockIdx.x; e < n; e += gridDim.x) {  // block-uniform
        const int ft = todo[1 + e];
        describe_tile<HAS_DESC, true>(P, ft / tiles_per_frame, ft % tiles_per_frame, tiles_per_frame, 0, 1, tile_tab, gray, pyr, blur, fin_all, fin_cnt, kps,
                                      desc, cap, counts, flags, icw, todo);
    }
}

int orb_launch_describe(mo_ctx* c, const uint8_t* d_gray, int batch, mo_keypoint* d_kps, uint8_t* d_desc, int cap, int* d_counts) {
    const Plan& P = c->plan;
    if (!c->d_dtile_tab) {  // (re)built with the plan: free_plan_buffers drops it
        // tile table (level | tile column << 8 | tile row << 20, level-major) followed by the intensity-centroid weights
        // [32 rows][8 weight + 8 mask dwords]
        std::vector<uint32_t> tab;
        for (int L = 0; L < P.nlevels; L++)
            for (int y = 0; y < (P.lv[L].bh + DT_H - 1) / DT_H; y++)
                for (int x = 0; x < (P.lv[L].bw + DT_W - 1) / DT_W; x++) tab.push_back((uint32_t)L | ((uint32_t)x << 8) | ((uint32_t)y << 20));
        c->n_dtiles = (int)tab.size();
        while (tab.size() % 4) tab.push_back(0);  // the weight rows are read as uint4
        c->dtile_icw_off = (int)tab.size();
        tab.resize(tab.size() + 512, 0u);
        for (int r = 0; r < 31; r++) {
            const int d = P.umax[r < 15 ? 15 - r : r - 15];
            for (int c4 = 0; c4 < 8; c4++) {
                uint32_t wv = 0, mv = 0;
                for (int b = 0; b < 4; b++) {
                    const int u = 4 * c4 - 15 + b;
                    if (u >= -d && u <= d) { wv |= (uint32_t)(u + 16) << (8 * b); mv |= 1u << (8 * b); }
                }
                tab[c->dtile_icw_off + r * 16 + c4] = wv;
                tab[c->dtile_icw_off + r * 16 + 8 + c4] = mv;
            }
        }
        HIPCHK(c, hipMalloc((void**)&c->d_dtile_tab, tab.size() * sizeof(uint32_t)));
        HIPCHK(c, hipMemcpy(c->d_dtile_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (c->n_dtiles == 0) {  // no level has a border region: every frame has zero keypoints
        HIPCHK(c, hipMemsetAsync(d_counts, 0, (size_t)batch * sizeof(int), c->stream));
        return MO_OK;
    }
    // one or two frames: DT_SPLIT_LATENCY workgroups per tile, each describing the keypoints of one residue class of the level's list.  A coarse
    // level is one or two tiles holding all of its ~ 120 keypoints - eight passes of 16 in one workgroup while most CUs sit idle; the tile is
    // loaded once per sharing workgroup, which costs nothing there (profiles/r04_ab_describe_split.txt)
    const dim3 grid(c->n_dtiles, batch, batch <= 2 ? DT_SPLIT_LATENCY : 1);
    const uint32_t inv_per = grid.x > 1 ? 0xFFFFFFFFu / grid.x + 1u : 0u;
    // todo list of the tiles the one-pass kernel leaves to k_describe_tiles_rare: [0] count (cleared by k_select of the same call), then
    // frame * tiles + tile entries; sized for every tile of the largest batch
    const size_t todo_need = (1 + (size_t)c->n_dtiles * c->batch_alloc * DT_SPLIT_LATENCY) * sizeof(int);  // (each sharing workgroup may leave an entry)
    if (c->dtodo_bytes < todo_need) {
        if (c->d_dtodo) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->d_dtodo)); c->d_dtodo = nullptr; c->dtodo_bytes = 0; }
        HIPCHK(c, hipMalloc((void**)&c->d_dtodo, todo_need));
        HIPCHK(c, hipMemsetAsync(c->d_dtodo, 0, sizeof(int), c->stream));
        c->dtodo_bytes = todo_need;
    }
    const uint32_t* icw = c->d_dtile_tab + c->dtile_icw_off;
    const dim3 rare_grid(std::min(c->n_dtiles * batch, 256));
    if (d_desc) {
        hipLaunchKernelGGL(k_describe_tiles<true>, grid, dim3(DT_NT), 0, c->stream, P, c->d_dtile_tab, inv_per, d_gray, c->d_pyr, c->d_blur,
                           c->d_fin, c->d_fin_cnt, d_kps, d_desc, cap, d_counts, c->flags_cur, icw, c->d_dtodo);
        hipLaunchKernelGGL(k_describe_tiles_rare<true>, rare_grid, dim3(DT_NT), 0, c->stream, P, c->d_dtile_tab, c->n_dtiles, d_gray, c->d_pyr,
                           c->d_blur, c->d_fin, c->d_fin_cnt, d_kps, d_desc, cap, d_counts, c->flags_cur, icw, c->d_dtodo);
    } else {
        hipLaunchKernelGGL(k_describe_tiles<false>, grid, dim3(DT_NT), 0, c->stream, P, c->d_dtile_tab, inv_per, d_gray, c->d_pyr, c->d_blur,
                           c->d_fin, c->d_fin_cnt, d_kps, d_desc, cap, d_counts, c->flags_cur, icw, c->d_dtodo);
        hipLaunchKernelGGL(k_describe_tiles_rare<false>, rare_grid, dim3(DT_NT), 0, c->stream, P, c->d_dtile_tab, c->n_dtiles, d_gray, c->d_pyr,
                           c->d_blur, c->d_fin, c->d_fin_cnt, d_kps, d_desc, cap, d_counts, c->flags_cur, icw, c->d_dtodo);
    }
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// ------------------------------------------------------------------ describe, grid cells (batched grid detector) ----
// Descriptors of the records k_gftt_records keeps (octave 0, angle as recorded: -1), one workgroup per grid cell: the blurred level-0
// tile of the cell (+ 19 px) goes through LDS once, the cell's records - a contiguous range of the frame's list, kbase[cell] ..
// kbase[cell + 1] - are described out of it 16 lanes each, like phase B / C of k_describe_tiles (k_describe_given samples global
// memory through one wavefront per keypoint: 0.26 ms per 256 frames of 1 140 records).
#define DC_TILE_BYTES 13312   // (80 + 38 + 14 -> 128) x (60 + 38) at 640 x 480 = 12 544
#define DC_MAXREC 256         // GF_MAXCORNERS: records per cell
__global__ __launch_bounds__(256) void k_describe_cells(Plan P, const uint8_t* __restrict__ blur, const mo_keypoint* __restrict__ kps,
                                                        const int32_t* __restrict__ kbase, uint8_t* __restrict__ desc, int cap, int cw,
                                                        int ch, int tpitch) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[DC_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) float4 s_pat[256];
    __shared__ float2 s_ab[DC_MAXREC];
    __shared__ uint32_t s_ctr[DC_MAXREC];  // LDS address of the record's centre sample
    const int cell = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x, grp = tid / DG, gl = tid % DG;
    const int i0 = kbase[(size_t)frame * 65 + cell], n = min(kbase[(size_t)frame * 65 + cell + 1] - i0, DC_MAXREC);
    if (n <= 0) return;  // block-uniform
    const int x0 = (cell & 7) * cw, y0 = (cell >> 3) * ch;
    const LevelInfo lv = P.lv[0];
    const uint8_t* bl = blur + (size_t)frame * P.blur_stride + lv.boff;
    const int xb_al = (x0 - 19) & ~7, rows = ch + 38, nq = tpitch >> 3, last_q = (lv.bpitch >> 3) - 1;
    {   // tile: every 8-byte piece in flight before the first store (rows clamped into the level, pieces into the row pitch)
        constexpr int U = 8;
        const int total = rows * nq;
        const uint32_t inv = 0xFFFFFFFFu / (uint32_t)nq + 1u;
        for (int t0 = tid; t0 < total; t0 += 256 * U) {
            uint2 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int t = min(t0 + u * 256, total - 1), r = (int)__umulhi((uint32_t)t, inv), c = t - r * nq;
                const int y = min(max(y0 - 19 + r, 0), lv.h - 1), q = min(max((xb_al >> 3) + c, 0), last_q);
                v[u] = ((const uint2*)(bl + (size_t)y * lv.bpitch))[q];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int t = t0 + u * 256, r = (int)__umulhi((uint32_t)t, inv), c = t - r * nq;
                if (t < total) *(uint2*)(s_tile + r * tpitch + 8 * c) = v[u];
            }
        }
    }
    {
        const int l = tid & 15, k = tid >> 4;
        const int8_t* pt = &c_pattern[(l * 16 + k) * 4];
        s_pat[k * DG + l] = make_float4((float)pt[0], (float)pt[1], (float)pt[2], (float)pt[3]);
    }
    const uint32_t tile_lds = (uint32_t)(uintptr_t)s_tile;
    for (int j = tid; j < n; j += 256) {  // one lane per record: centre and (float)cos / (float)sin through f64 as cv2 computes them
        const mo_keypoint kp = kps[(size_t)frame * cap + i0 + j];
        const float inv = 1.f / lv.scale;
        const int cx = __float2int_rn(kp.x * inv), cy = __float2int_rn(kp.y * inv);
        float angle = kp.angle;
        angle *= (float)(3.14159265358979323846 / 180.f);
        double sd, cd;
        sincos((double)angle, &sd, &cd);
        s_ab[j] = make_float2((float)cd, (float)sd);
        s_ctr[j] = tile_lds + (uint32_t)((cy - (y0 - 19)) * tpitch + (cx - xb_al));
    }
    __syncthreads();
    for (int j = grp; j < n; j += 256 / DG) {
        const float2 ab = s_ab[j];
        int glo = gl;
        asm volatile("" : "+v"(glo));  // (keeps the 16 pattern reads inside the loop)
        *(uint16_t*)(desc + ((size_t)frame * cap + i0 + j) * 32 + 2 * gl) = rbrief_u16_lds(s_ctr[j], tpitch, ab.x, ab.y, s_pat, glo);
    }
}

int orb_launch_describe_cells(mo_ctx* c, const mo_keypoint* d_kps, const int32_t* d_kbase, uint8_t* d_desc, int cap, int batch) {
    const Plan& P = c->plan;
    const int cw = P.w / 8, ch = P.h / 8;
    const int tpitch = (cw + 38 + 7 + 7) & ~7;
    if ((size_t)tpitch * (ch + 38) > DC_TILE_BYTES || P.edge_threshold < 19) return MO_ERR_UNSUPPORTED;  // (the caller falls back)
    hipLaunchKernelGGL(k_describe_cells, dim3(64, batch), dim3(256), 0, c->stream, P, c->d_blur, d_kps, d_kbase, d_desc, cap, cw, ch, tpitch);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// compute() with caller keypoints: angle as supplied, level = kp.octave.  blockIdx.y = frame of a batch (the batched grid detector):
// records kps + frame * cap, descriptors desc + frame * cap * 32, record count n_dev[frame * n_stride] (n_dev null: n).
__global__ __launch_bounds__(256) void k_describe_given(Plan P, const uint8_t* __restrict__ gray,
                                                        const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                        const mo_keypoint* __restrict__ kps, int n, const int* __restrict__ n_dev,
                                                        int n_stride, uint8_t* __restrict__ desc) {
    const int lane = threadIdx.x & 63, frame = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int cap = n;  // records and descriptor rows of a frame are `n` apart
    if (n_dev) n = min(n, n_dev[(size_t)frame * n_stride]);  // (the count of records is only known on the device)
    if (k >= n) return;
    mo_keypoint kp = kps[(size_t)frame * cap + k];
    const int L = kp.octave;
    const LevelInfo lv = P.lv[L];
    float inv = 1.f / lv.scale;
    int cx = __float2int_rn(kp.x * inv), cy = __float2int_rn(kp.y * inv);
    const uint8_t* img = level_ptr(P, L, gray, pyr, frame);
    const uint8_t* bl = blur + (size_t)frame * P.blur_stride + lv.boff;
    rbrief_wave<true>(bl, lv.bpitch, img, lv.pitch, lv.w, lv.h, cx, cy, kp.angle, desc + ((size_t)frame * cap + k) * 32, lane);
}

int orb_launch_describe_given(mo_ctx* c, const uint8_t* d_gray, const mo_keypoint* d_kps, int n, uint8_t* d_desc, const int* d_n, int batch,
                              int n_stride) {
    if (n <= 0) return MO_OK;
    const Plan& P = c->plan;
    hipLaunchKernelGGL(k_describe_given, dim3((n + 3) / 4, batch), dim3(256), 0, c->stream, P, d_gray, c->d_pyr, c->d_blur, d_kps,
                       n, d_n, n_stride, d_desc);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

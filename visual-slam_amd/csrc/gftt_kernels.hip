// gftt_kernels.hip -- grid Shi-Tomasi corners for ORBExtractor.distribute_keypoints (gfx950).
// Replaces the 64 x cv2.goodFeaturesToTrack(image, maxCorners = n // 64, qualityLevel 0.01, minDistance 10, mask = cell)
// calls of the reference (src/orbslam2/extractor.py:115-129): the min-eigenvalue map is computed ONCE per frame
// (the reference recomputes it for every cell), then one workgroup per grid cell thresholds at 1 % of the cell's
// maximum, keeps 3x3 local maxima, sorts them by (quality desc, address desc) and runs the greedy min-distance pick.
// Float arithmetic order is the one fixed in oracle/orb_oracle.cpp (min_eigen_map); compile with -ffp-contract=off.
#include <cfloat>

#include "common.h"

__device__ __forceinline__ int gf_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

#define ME_TW 64
#define ME_TH 32
#define ME_RPT (ME_TH / 4)  // output rows per thread: a wavefront owns ME_RPT consecutive rows of the tile, a lane one column

// cornerMinEigenVal(blockSize 3, Sobel 3, BORDER_REFLECT_101): u8 tile (+2 halo) -> Dx, Dy (+1 halo) -> 3x3 box of the
// covariance terms in f64 -> (a + c) - sqrt((a - c)^2 + b^2).
// Round 3: (1) the pixel tile goes through LDS once (rounds 1 - 2 read nine reflected bytes from global memory per gradient sample);
// (2) the 3x3 box sums are SEPARABLE: a thread walks down its column keeping the horizontal 3-sums of the last three rows, 13.5
// instead of 27 f64 additions (and conversions) per pixel.  The order of the f64 additions is free here: the terms are float32
// products of gradients that are multiples of 1 / 3060 up to rounding, i.e. zero or between 2^-24 and 2^-1 in magnitude with 24-bit
// mantissas - nine of them add up exactly in 53 bits whatever the order, so the result is the oracle's bit for bit.
__global__ __launch_bounds__(256) void k_min_eigen(const uint8_t* __restrict__ gray, int w, int h, float* __restrict__ eig) {
    __shared__ __attribute__((aligned(16))) uint8_t s_px[(ME_TH + 4) * (ME_TW + 8)];  // (pitch ME_TW + 8 in the dword path, ME_TW + 4 in the byte path)
    __shared__ float s_dx[(ME_TH + 2) * (ME_TW + 2)], s_dy[(ME_TH + 2) * (ME_TW + 2)];
    const int tx0 = blockIdx.x * ME_TW, ty0 = blockIdx.y * ME_TH, tid = threadIdx.x;
    gray += (size_t)blockIdx.z * w * h;  // frame of a batch (dense frames, dense maps)
    eig += (size_t)blockIdx.z * w * h;
    // The Sobel window of a REFLECTED gradient position is reflected again, which is not one reflection of the pixel coordinate:
    // tiles that touch the image border (block-uniform test; 31 % of the tiles of a 640 x 480 frame) compute the in-image gradients
    // from pixels staged with ONE reflection of the pixel coordinate - what Sobel applies at an in-image position - and copy them to
    // the ring just outside the image (the box filter's BORDER_REFLECT_101 of the covariance maps: gradient(-1) = gradient(1)).
    // (Rounds 1 - 3 evaluated the double reflection of border tiles from global memory, nine reflected byte reads per gradient position:
    //  0.464 against 0.474 ms for the grid stage of 256 frames - 2 % faster, at the price of a second implementation of the gradient.)
    const bool interior = tx0 >= 2 && ty0 >= 2 && tx0 + ME_TW + 2 <= w && ty0 + ME_TH + 2 <= h;
    const double scale_d = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    const float f1 = (float)(1.0f * scale_d), f0 = (float)(2.0f * scale_d);
    constexpr int PW = ME_TW + 8, PD = PW / 4;  // s_px: byte b of row r <-> pixel (tx0 - 4 + b, ty0 - 2 + r)
    auto refl = [](int p, int len) { p = p < 0 ? -p : p >= len ? 2 * len - 2 - p : p; return min(max(p, 0), len - 1); };  // (far outside: clamped, unused)
    if ((w & 3) == 0 && (((size_t)gray) & 3) == 0) {
        // dword-aligned frame: 3 loads per thread, all in flight before the first store; rows are reflected per row, a dword that
        // straddles the left or right image border (two per row of an edge tile) is assembled from reflected bytes
        constexpr int NDW = (ME_TH + 4) * PD, NLD = (NDW + 255) / 256;
        uint32_t v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int i = min(tid + u * 256, NDW - 1), r = i / PD, c = i - r * PD;
            const uint8_t* row = gray + (size_t)(interior ? ty0 + r - 2 : refl(ty0 + r - 2, h)) * w;
            const int x0 = tx0 - 4 + 4 * c;
            if (interior || (x0 >= 0 && x0 + 3 < w)) v[u] = *(const uint32_t*)(row + x0);
            else {
                v[u] = 0;
#pragma unroll
                for (int bb = 0; bb < 4; bb++) v[u] |= (uint32_t)row[refl(x0 + bb, w)] << (8 * bb);
            }
        }
#pragma unroll
        for (int u = 0; u < NLD; u++)
            if (tid + u * 256 < NDW) ((uint32_t*)s_px)[tid + u * 256] = v[u];
    } else {
        // a frame that is not dword-aligned: bytes 2 .. ME_TW + 5 of every row, coordinates reflected once
        constexpr int NB = ME_TW + 4, NPX = (ME_TH + 4) * NB, NLD = (NPX + 255) / 256;
        uint8_t v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int i = min(tid + u * 256, NPX - 1), r = i / NB, c = i - r * NB;
            v[u] = gray[(size_t)refl(ty0 + r - 2, h) * w + refl(tx0 + c - 2, w)];
        }
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int i = tid + u * 256, r = i / NB, c = i - r * NB;
            if (i < NPX) s_px[r * PW + c + 2] = v[u];
        }
    }
    __syncthreads();
    {
        // one task = 4 adjacent gradient positions out of 3 rows x 2 dwords, every byte converted once by v_cvt_f32_ubyteN (round 3's
        // first form: 9 LDS byte reads + 9 conversions per gradient position).  (float)(a - b) of two bytes equals (float)a - (float)b
        // exactly, so these are the oracle's float expressions in the oracle's order.
        constexpr int NQ = (ME_TW + 2 + 3) / 4;  // quads of gradient positions per row (the last one is partly outside: not stored)
        for (int i = tid; i < (ME_TH + 2) * NQ; i += 256) {
            const int r = i / NQ, q = i - r * NQ;
            // gradient column c = 4q + k (x = tx0 - 1 + c) reads pixel columns x - 1 .. x + 1 = bytes c + 2 .. c + 4: bytes 4q + 2 .. 4q + 7
            float px[3][6];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const uint32_t* row = (const uint32_t*)(s_px + (r + j) * PW) + q;
                const uint32_t d0 = row[0], d1 = row[1];
                px[j][0] = (float)((d0 >> 16) & 0xFFu); px[j][1] = (float)(d0 >> 24);
                px[j][2] = (float)(d1 & 0xFFu); px[j][3] = (float)((d1 >> 8) & 0xFFu); px[j][4] = (float)((d1 >> 16) & 0xFFu); px[j][5] = (float)(d1 >> 24);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = 4 * q + k;
                if (c < ME_TW + 2) {
                    const float Rm = px[0][k + 2] - px[0][k], R0 = px[1][k + 2] - px[1][k], Rp = px[2][k + 2] - px[2][k];
                    const float t = Rm + Rp;
                    const float uu = t * f1;
                    const float vv = R0 * f0;
                    s_dx[r * (ME_TW + 2) + c] = uu + vv;
                    const float Cm = ((f1 * px[0][k]) + f0 * px[0][k + 1]) + f1 * px[0][k + 2];
                    const float Cp = ((f1 * px[2][k]) + f0 * px[2][k + 1]) + f1 * px[2][k + 2];
                    s_dy[r * (ME_TW + 2) + c] = Cp - Cm;
                }
            }
        }
    }
    if (!interior) {  // block-uniform: the ring a box sum of an in-image output reaches takes the gradients of its reflected positions
        __syncthreads();
        for (int i = tid; i < (ME_TH + 2) * (ME_TW + 2); i += 256) {
            const int r = i / (ME_TW + 2), c = i - r * (ME_TW + 2);
            const int y = ty0 + r - 1, x = tx0 + c - 1;
            if ((y == -1 || y == h || x == -1 || x == w) && y <= h && x <= w) {
                const int src = (refl(y, h) - ty0 + 1) * (ME_TW + 2) + (refl(x, w) - tx0 + 1);  // an in-image position of this tile: never written here
                s_dx[i] = s_dx[src];
                s_dy[i] = s_dy[src];
            }
        }
    }
    __syncthreads();
    const int lx = tid & 63, ly0 = (tid >> 6) * ME_RPT;
    const int x = tx0 + lx;
    // horizontal 3-sums (f64) of the products on gradient row r of the tile, columns lx .. lx + 2
    auto hsum = [&](int r, double& hxx, double& hxy, double& hyy) {
        const float* dxp = s_dx + r * (ME_TW + 2) + lx;
        const float* dyp = s_dy + r * (ME_TW + 2) + lx;
        hxx = 0; hxy = 0; hyy = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float a = dxp[i], b = dyp[i];
            const float xx = a * a, xy = a * b, yy = b * b;
            hxx += (double)xx; hxy += (double)xy; hyy += (double)yy;
        }
    };
    double h0x, h0y, h0z, h1x, h1y, h1z;
    hsum(ly0, h0x, h0y, h0z);
    hsum(ly0 + 1, h1x, h1y, h1z);
#pragma unroll
    for (int k = 0; k < ME_RPT; k++) {
        double h2x, h2y, h2z;
        hsum(ly0 + k + 2, h2x, h2y, h2z);
        const int y = ty0 + ly0 + k;
        if (x < w && y < h) {
            const double sxx = (h0x + h1x) + h2x, sxy = (h0y + h1y) + h2y, syy = (h0z + h1z) + h2z;
            float a = (float)sxx * 0.5f, b = (float)sxy, c = (float)syy * 0.5f;
            float amc = a - c;
            float rad = amc * amc + b * b;
            eig[(size_t)y * w + x] = (a + c) - sqrtf(rad);
        }
        h0x = h1x; h0y = h1y; h0z = h1z;
        h1x = h2x; h1y = h2y; h1z = h2z;
    }
}

#define GF_CAP 2048       // local maxima per cell sorted at once (a cell with more is taken in rounds of the GF_CAP strongest remaining)
#define GF_MAXCORNERS 256

#define GF_TILE_MAX 5248  // floats of the cell + 1-px ring held in LDS ((80 + 2) x (60 + 2) = 5084 at 640 x 480); larger cells read the map from global memory
#define GF_TILE_KEYS 384   // keys that may be filled while the tile is in use (typical: 50 - 120 local maxima per cell); the tile lies behind
                          // them in the same array and becomes key space when a cell has more (24 KB in all, 6 workgroups per CU: 0.551 -> 0.535 ms
                          // for the grid stage against 1024 keys / 29 KB / 5 workgroups; a cell with more lists its maxima again from global memory)

// one workgroup per grid cell (blockIdx.y = frame of a batch)
__global__ __launch_bounds__(256) void k_gftt_cell(const float* __restrict__ eig, int w, int h, int cols, int cw, int ch,
                                                   int max_corners, double quality, double min_dist, float* __restrict__ out_xy,
                                                   int* __restrict__ out_n, int* flags) {
    __shared__ unsigned long long s_key[GF_TILE_KEYS + GF_TILE_MAX / 2];
    static_assert(GF_TILE_KEYS + GF_TILE_MAX / 2 >= GF_CAP, "the key array with the tile's space holds GF_CAP keys");
    float* const s_tile = (float*)(s_key + GF_TILE_KEYS);  // dead once the candidates are listed; a cell with more than GF_TILE_KEYS
                                                           // local maxima lists them again from global memory
    __shared__ float s_red[4];
    __shared__ int s_n;
    __shared__ float s_ax[GF_MAXCORNERS], s_ay[GF_MAXCORNERS];
    const int cell = blockIdx.x, ci = cell / cols, cj = cell - ci * cols;
    {   // frame of a batch: dense maps, 64 x lim corner slots and 64 + 2 counters per frame
        const int lim0 = min(max_corners > 0 ? max_corners : GF_MAXCORNERS, GF_MAXCORNERS);
        eig += (size_t)blockIdx.y * w * h;
        out_xy += (size_t)blockIdx.y * gridDim.x * lim0 * 2;
        out_n += (size_t)blockIdx.y * (gridDim.x + 2);
    }
    const int x0 = cj * cw, y0 = ci * ch, x1 = x0 + cw, y1 = y0 + ch;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the cell and its 1-px ring (clamped into the image: ring positions outside it are never read) go through LDS once; rounds 1 - 2
    // read the map from global memory up to ten times per pixel (maximum, threshold, 3x3 dilation)
    const int tw = cw + 2, th = ch + 2;
    const bool in_lds = tw * th <= GF_TILE_MAX;  // block-uniform
    float m = -FLT_MAX;
    if (in_lds) {
        // every load of a thread in flight before the first store (a plain loop makes one global round trip per value; round 3's first
        // form kept 8 in flight: three round trips for the 5084 floats of a 640 x 480 cell, 12.7 k of the workgroup's 61 k cycles).
        // The cell maximum (minMaxLoc with the cell mask) is taken from the registers on the way: no second pass over the tile.
        constexpr int GF_LD = (GF_TILE_MAX + 255) / 256;
        const uint32_t inv_tw = 0xFFFFFFFFu / (uint32_t)tw + 1u;
        float v[GF_LD];
#pragma unroll
        for (int u = 0; u < GF_LD; u++) {
            const int i = tid + u * 256;
            if (i < tw * th) {
                const int r = (int)__umulhi((uint32_t)i, inv_tw), c = i - r * tw;
                const int yy = min(max(y0 - 1 + r, 0), h - 1), xx = min(max(x0 - 1 + c, 0), w - 1);
                v[u] = eig[(size_t)yy * w + xx];
            }
        }
#pragma unroll
        for (int u = 0; u < GF_LD; u++) {
            const int i = tid + u * 256;
            if (i < tw * th) {
                s_tile[i] = v[u];
                const int r = (int)__umulhi((uint32_t)i, inv_tw), c = i - r * tw;
                if (r >= 1 && r <= ch && c >= 1 && c <= cw) m = fmaxf(m, v[u]);
            }
        }
    } else {
        for (int i = tid; i < cw * ch; i += 256) {
            const int r = i / cw, c = i - r * cw;
            m = fmaxf(m, eig[(size_t)(y0 + r) * w + x0 + c]);
        }
    }
    if (tid == 0) s_n = 0;
    // 1. cell maximum (minMaxLoc with the cell mask)
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) s_red[wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const double maxVal = m > 0 ? (double)m : 0.0;
    const float thr = (float)(maxVal * quality);
    // 2. candidates: above threshold (THRESH_TOZERO) and equal to the 3x3 dilation of the thresholded map
    const int ya = max(y0, 1), yb = min(y1, h - 1), xa = max(x0, 1), xb = min(x1, w - 1);
    const int cw2 = xb - xa, n_in = cw2 > 0 && yb > ya ? cw2 * (yb - ya) : 0;
    // (v > thr >= 0 and v == max over the 3x3 window of the THRESHOLDED map  <=>  v > thr and v >= its eight RAW neighbours: a
    //  neighbour at or below the threshold counts as 0 < v either way)
    if (in_lds && n_in > 0) {
        // out of the LDS tile, one task = 4 adjacent positions of a row: 18 LDS reads, 12 + 4 three-way maxima (the first form read
        // and thresholded nine values per position: 16 k of the workgroup's 41 k cycles)
        const int nq = (cw2 + 3) >> 2, ntask = nq * (yb - ya);
        const uint32_t inv_nq = 0xFFFFFFFFu / (uint32_t)nq + 1u;
        for (int i = tid; i < ntask; i += 256) {
            const int ry = nq > 1 ? (int)__umulhi((uint32_t)i, inv_nq) : i, q = i - ry * nq;
            const int yy = ya + ry, xx0 = xa + 4 * q;
            const float* p = s_tile + (yy - y0) * tw + (xx0 - x0);  // row yy - 1, column xx0 - 1 of the tile
            const int cmax = tw - 1 - (xx0 - x0);                   // last readable column offset of this row (masked positions re-read it)
            float hm[3][4], ctr[4];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                float a[6];
#pragma unroll
                for (int k = 0; k < 6; k++) a[k] = p[j * tw + min(k, cmax)];
#pragma unroll
                for (int k = 0; k < 4; k++) hm[j][k] = fmaxf(fmaxf(a[k], a[k + 1]), a[k + 2]);
                if (j == 1) { ctr[0] = a[1]; ctr[1] = a[2]; ctr[2] = a[3]; ctr[3] = a[4]; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float v = ctr[k], mx = fmaxf(fmaxf(hm[0][k], hm[1][k]), hm[2][k]);
                if (xx0 + k < xb && v > thr && v == mx) {
                    const int slot = atomicAdd(&s_n, 1);
                    if (slot < GF_TILE_KEYS) s_key[slot] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(yy * w + xx0 + k);
                }
            }
        }
        __syncthreads();
    }
    if (!in_lds || s_n > GF_TILE_KEYS) {  // (block-uniform) no tile, or more local maxima than fit beside it: list them from global memory
        __syncthreads();
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (int i = tid; i < n_in; i += 256) {
            const int yy = ya + i / cw2, xx = xa + i % cw2;
            const float* p = eig + (size_t)yy * w + xx;
            const float v = p[0];
            if (!(v > thr)) continue;
            float mx = v;
#pragma unroll
            for (int j = -1; j <= 1; j++)
#pragma unroll
                for (int k = -1; k <= 1; k++) mx = fmaxf(mx, p[j * w + k]);
            if (v == mx) {
                const int slot = atomicAdd(&s_n, 1);
                if (slot < GF_CAP) s_key[slot] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(yy * w + xx);
            }
        }
    }
    __syncthreads();
    const int n_all = s_n;  // every local maximum of the cell (the key array holds the first min(n_all, cap) of them)
    const int lim = min(max_corners > 0 ? max_corners : GF_MAXCORNERS, GF_MAXCORNERS);
    int nacc = 0;           // corners accepted so far (maintained by wavefront 0, published through s_n between rounds)
    // 3. sort descending by (value, address): bitonic network on the next power of two (padding keys = 0 sink to the end)
    // 4. greedy minimum-distance pick by one wavefront, 64 candidates of the sorted list at a time, one per lane: a lane first
    //    drops its candidate if it lies within min_dist of a corner accepted in earlier batches, then the surviving lanes are
    //    resolved in list order - the lowest one is accepted and knocks out the later lanes within min_dist of it.  The accepted
    //    set and its order are those of the sequential loop (a candidate is accepted iff no EARLIER accepted corner is near).
    auto sort_and_pick = [&](int n) {  // block-uniform n <= GF_CAP keys in s_key
        if (n <= 256) {
            // short list (typical: 50 - 120 keys): every key counts the keys above it - its position in the descending list, keys
            // are distinct - out of broadcast LDS reads: one pass and two barriers instead of the 28 barrier-separated steps of
            // the network on 128 keys (13.8 k of the workgroup's 61 k cycles)
            const unsigned long long mine = tid < n ? s_key[tid] : 0ull;
            int rank = 0;
            for (int j = 0; j < n; j++) rank += s_key[j] > mine ? 1 : 0;
            __syncthreads();
            if (tid < n) s_key[rank] = mine;
            __syncthreads();
        } else {
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        for (int i = n + tid; i < np2; i += 256) s_key[i] = 0ull;
        __syncthreads();
        for (int k = 2; k <= np2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < np2; i += 256) {
                    int l = i ^ j;
                    if (l > i) {
                        unsigned long long a = s_key[i], b = s_key[l];
                        bool desc = (i & k) == 0;
                        if (desc ? a < b : a > b) { s_key[i] = b; s_key[l] = a; }
                    }
                }
                __syncthreads();
            }
        }
        if (wv == 0) {
            const float md2 = (float)(min_dist * min_dist);
            const bool use_dist = min_dist >= 1.0;
            __builtin_amdgcn_s_setprio(3);  // one dependent chain against the other workgroups' wavefronts on this SIMD: take the issue slots first
            const uint32_t inv_w = 0xFFFFFFFFu / (uint32_t)w + 1u;  // pos / w == mulhi(pos, inv_w): pos < w * h < 2^32 / w for every supported size
            const bool exact_div = (unsigned long long)w * h * w < (1ull << 32);
            for (int i0 = 0; i0 < n && nacc < lim; i0 += 64) {  // wave-uniform
                // straight-line code, scalar loop control: one wavefront works here alone (nothing hides a taken branch or an LDS
                // round trip), and the first form's per-lane early exits cost ~500 cycles per accepted corner
                const int i = i0 + lane;
                bool alive = i < n;
                const unsigned pos = (unsigned)(s_key[min(i, n - 1)] & 0xFFFFFFFFull);
                const int yy = exact_div ? (int)__umulhi(pos, inv_w) : (int)(pos / (unsigned)w), xx = (int)pos - yy * w;
                const float fx = (float)xx, fy = (float)yy;
                nacc = __builtin_amdgcn_readfirstlane(nacc);  // (wave-uniform by construction)
                if (use_dist)
                    for (int j = 0; j < nacc; j++) {  // corners accepted in earlier batches: broadcast LDS reads, no early exit
                        const float dx = fx - s_ax[j], dy = fy - s_ay[j];
                        alive = alive & !(dx * dx + dy * dy < md2);
                    }
                unsigned long long live = __ballot(alive);
                while (live && nacc < lim) {  // wave-uniform
                    const int l = __ffsll((long long)live) - 1;  // wave-uniform: v_readlane instead of two LDS-routed shuffles
                    const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fx), l));
                    const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fy), l));
                    if (lane == 0) { s_ax[nacc] = ax; s_ay[nacc] = ay; }  // (the corners leave for global memory after the last round)
                    nacc++;
                    const float dx = fx - ax, dy = fy - ay;
                    alive = alive & (lane != l) & !(use_dist & (dx * dx + dy * dy < md2));
                    live = __ballot(alive);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // s_ax / s_ay written by lane 0 are read by all lanes in the next batch
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            __builtin_amdgcn_s_setprio(0);
        }
    };
    if (n_all <= GF_CAP) {
        sort_and_pick(n_all);
    } else {
        // More local maxima than the key array sorts at once (plateaus of a synthetic pattern; the large cells of a Full-HD frame of
        // dense texture): the candidates are taken in rounds of the GF_CAP largest remaining keys.  Keys are distinct (the address is
        // part of them), so "the GF_CAP largest keys below `up`" is the set lo <= key < up for the one lo a bisection over the
        // 64-bit key finds (each probe counts the keys of the cell in a range: a scan of the map in global memory).  Rare and slow,
        // but the same corners in the same order as one long sorted list.
        auto scan = [&](unsigned long long lo, unsigned long long up, bool list) -> int {  // keys in [lo, up): count, or list into s_key
            int cnt = 0;
            for (int i = tid; i < n_in; i += 256) {
                const int yy = ya + i / cw2, xx = xa + i % cw2;
                const float* p = eig + (size_t)yy * w + xx;
                const float v = p[0];
                if (!(v > thr)) continue;
                const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(yy * w + xx);
                if (key < lo || key >= up) continue;
                float mx = v;
#pragma unroll
                for (int j = -1; j <= 1; j++)
#pragma unroll
                    for (int k = -1; k <= 1; k++) {
                        float q = p[j * w + k];
                        q = q > thr ? q : 0.f;
                        mx = fmaxf(mx, q);
                    }
                if (v == mx) {
                    if (list) { const int slot = atomicAdd(&s_n, 1); if (slot < GF_CAP) s_key[slot] = key; }
                    cnt++;
                }
            }
            if (list) return 0;
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            __syncthreads();  // (s_red: the previous probe's readers are done)
            if (lane == 0) s_red[wv] = __int_as_float(cnt);
            __syncthreads();
            return __float_as_int(s_red[0]) + __float_as_int(s_red[1]) + __float_as_int(s_red[2]) + __float_as_int(s_red[3]);
        };
        unsigned long long up = ((unsigned long long)__float_as_uint(m) + 1ull) << 32;  // above every key (m: the cell maximum)
        int remaining = n_all;
        while (remaining > 0) {  // block-uniform
            unsigned long long lo = 0ull;
            if (remaining > GF_CAP) {
                unsigned long long a = (unsigned long long)__float_as_uint(thr) << 32, b = up;  // keys in [a, up) > GF_CAP >= keys in [b, up)
                while (b - a > 1ull) {
                    const unsigned long long mid = a + ((b - a) >> 1);
                    if (scan(mid, up, false) <= GF_CAP) b = mid; else a = mid;
                }
                lo = b;
            }
            __syncthreads();
            if (tid == 0) s_n = 0;
            __syncthreads();
            scan(lo, up, true);
            __syncthreads();
            const int nr = min(s_n, GF_CAP);  // == min(remaining, GF_CAP)
            sort_and_pick(nr);
            if (tid == 0) s_n = nacc;  // (thread 0 is lane 0 of wavefront 0)
            __syncthreads();
            const int acc_all = s_n;
            __syncthreads();
            if (acc_all >= lim || nr <= 0) break;
            remaining -= nr;
            up = lo;
        }
    }
    if (wv == 0) {  // accepted corners in acceptance order: coalesced stores out of the LDS list (lane 0's writes are fenced above)
        for (int j = lane; j < 2 * nacc; j += 64) out_xy[(size_t)cell * lim * 2 + j] = (j & 1) ? s_ay[j >> 1] : s_ax[j >> 1];
        if (lane == 0) out_n[cell] = nacc;
    }
}

// batch frames [batch][h][w] -> d_eig [batch][h][w], d_xy [batch][64][per_cell][2], d_n [batch][64 + 2] (the two extra counters
// per frame are k_gftt_records' totals)
int gftt_launch(mo_ctx* c, const uint8_t* d_gray, int w, int h, int n_features, float* d_eig, float* d_xy, int* d_n, int batch) {
    const int rows = 8, cols = 8, ch = h / rows, cw = w / cols, per_cell = n_features / (rows * cols);
    if (per_cell > GF_MAXCORNERS) return mo_fail(c, MO_ERR_UNSUPPORTED, "more than 256 corners per grid cell");
    hipLaunchKernelGGL(k_min_eigen, dim3((w + ME_TW - 1) / ME_TW, (h + ME_TH - 1) / ME_TH, batch), dim3(256), 0, c->stream, d_gray, w, h, d_eig);
    hipLaunchKernelGGL(k_gftt_cell, dim3(rows * cols, batch), dim3(256), 0, c->stream, d_eig, w, h, cols, cw, ch, per_cell, 0.01, 10.0,
                       d_xy, d_n, c->flags_cur);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

// Fused distribute_keypoints (mo_orb_grid_detect_compute): the 64 per-cell corner lists -> KeyPoint(x, y, 31) records of the corners
// orb.compute keeps (Feature2D::compute drops keypoints whose ROUNDED position lies within edge_threshold of the border), in list
// order, plus their indices in the cell-major list of all corners.  One workgroup: two block scans over <= 64 x per_cell slots.
// counts2[0] = corners in all cells, counts2[1] = records kept.
// Batched (blockIdx.x = frame): xy / cell_n advance by 64 x per_cell x 2 / 66 per frame, rec / kept by rec_stride records; counts2
// = cell_n + 64 of the frame; counts_out (may be null) [frame] = records kept (the frame's keypoint count in the batched mode).
// Records beyond rec_stride are dropped and bit 1 of the flag word is raised (the count still reports the need).
__global__ __launch_bounds__(256) void k_gftt_records(const float* __restrict__ xy, int* __restrict__ cell_n, int per_cell, int w,
                                                      int h, int edge, mo_keypoint* __restrict__ rec, int32_t* __restrict__ kept,
                                                      int rec_stride, int32_t* __restrict__ counts_out, int* flags,
                                                      int32_t* __restrict__ kbase /* [frame][65] or null: records kept before cell c */) {
    __shared__ int s_base[65];
    __shared__ int s_kcnt[64];
    __shared__ int s_wsum[4];
    __shared__ int s_run;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    xy += (size_t)blockIdx.x * 64 * per_cell * 2;
    cell_n += (size_t)blockIdx.x * 66;
    rec += (size_t)blockIdx.x * rec_stride;
    if (kept) kept += (size_t)blockIdx.x * rec_stride;
    int* counts2 = cell_n + 64;
    if (tid < 64) s_kcnt[tid] = 0;
    if (tid == 0) {
        int a = 0;
        for (int cl = 0; cl < 64; cl++) { s_base[cl] = a; a += min(cell_n[cl], per_cell); }
        s_base[64] = a;
        s_run = 0;
    }
    __syncthreads();
    const int total = s_base[64];
    const bool no_border = !(edge > 0 && (h <= 2 * edge || w <= 2 * edge));  // (otherwise cv2 keeps nothing)
    for (int b0 = 0; b0 < total; b0 += 256) {  // list positions in order, 256 per trip
        const int g = b0 + tid;
        bool keep = false;
        float x = 0.f, y = 0.f;
        int cl = 0;
        if (g < total) {
            for (int step = 32; step > 0; step >>= 1)  // the cell whose list holds position g
                if (cl + step < 64 && s_base[cl + step] <= g) cl += step;
            const int i = g - s_base[cl];
            x = xy[((size_t)cl * per_cell + i) * 2]; y = xy[((size_t)cl * per_cell + i) * 2 + 1];
            const int xi = (int)rintf(x), yi = (int)rintf(y);  // cvRound
            keep = no_border && (edge <= 0 || (xi >= edge && xi < w - edge && yi >= edge && yi < h - edge));
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wsum[wv] = __popcll(m);
        __syncthreads();
        int base = s_run;
        for (int k = 0; k < wv; k++) base += s_wsum[k];
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < rec_stride) {
            mo_keypoint kp;
            kp.x = x; kp.y = y; kp.size = 31.f; kp.angle = -1.f; kp.response = 0.f; kp.octave = 0; kp.class_id = -1;
            rec[pos] = kp;
            if (kept) kept[pos] = g;
            atomicAdd(&s_kcnt[cl], 1);
        }
        __syncthreads();
        if (tid == 0) s_run += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        __syncthreads();
    }
    if (tid == 0) {
        counts2[0] = total; counts2[1] = s_run;
        if (counts_out) counts_out[blockIdx.x] = s_run;
        if (s_run > rec_stride) atomicOr(&flags[0], 2);
        if (kbase) {
            int a = 0;
            for (int c2 = 0; c2 < 64; c2++) { kbase[(size_t)blockIdx.x * 65 + c2] = a; a += s_kcnt[c2]; }
            kbase[(size_t)blockIdx.x * 65 + 64] = a;
        }
    }
}

int gftt_records_launch(mo_ctx* c, const float* d_xy, int* d_cell_n, int per_cell, int w, int h, int edge, mo_keypoint* d_rec,
                        int32_t* d_kept, int rec_stride, int32_t* d_counts_out, int batch, int32_t* d_kbase) {
    hipLaunchKernelGGL(k_gftt_records, dim3(batch), dim3(256), 0, c->stream, d_xy, d_cell_n, per_cell, w, h, edge, d_rec, d_kept, rec_stride,
                       d_counts_out, c->flags_cur, d_kbase);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

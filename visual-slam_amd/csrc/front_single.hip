// Pyramid + Gaussian blur of ONE frame (or a few) in one launch: the single-frame form of orb_launch_pyramid + orb_launch_blur.
//
// A host call on one 640x480 frame spent 62 of its 202 us of GPU time between the upload and FAST in seven dependent k_resize2
// launches and one k_blur (4 - 5 us each for a few tiles of work, plus 4 us of queue starvation after each of the last five: the
// host cannot enqueue as fast as these kernels end; profiles/r04_single_frame_timeline_detect.txt).  The levels depend on one
// another, so one launch needs either device-wide barriers between the levels or no exchange between workgroups at all.  This
// kernel takes the second way: a workgroup owns one tile of EVERY level (the same fraction of each level's columns and rows) and
// builds the chain level 0 -> 1 -> ... in LDS, recomputing the few pixels beside its tile that its own next level and its own blur
// read (INTER_LINEAR_EXACT is a function of the source pixels alone, so a recomputed pixel equals the neighbour's: the level-0
// halo is about 25 px at 8 levels of 1.2).  The arithmetic is k_resize2's and k_blur's, value for value (same packed coefficient
// entries, same u16 row interpolants / row sums, same rounding); the batched path keeps those kernels, whose tiles are sized for
// chip-wide throughput, and any geometry this kernel's boxes do not fit keeps them too (fs_build leaves fs_ok false).
//
// Reference: cv2.ORB's pyramid (resize of the previous level, INTER_LINEAR_EXACT) and GaussianBlur(7x7, sigma 2, REFLECT_101)
// behind /root/reference/src/orbslam2/extractor.py:50-67 (detectAndCompute).
#include "common.h"

#include <algorithm>
#include <cstring>
#include <vector>

#define FS_NT 1024
#define FS_BOX_INTS 14
// own box of a tile on level 0, about: the phases are bound by instruction issue inside a CU, so more, smaller tiles put idle CUs to
// work until the halo (about 25 px on each side at 8 levels of 1.2) outweighs it (A/B: profiles/r04_ab_front_single_tiles.txt)
#ifndef FS_TILE_W
#define FS_TILE_W 48
#endif
#ifndef FS_TILE_H
#define FS_TILE_H 40
#endif
#define FS_HDR_INTS (MO_MAX_LEVELS * FS_BOX_INTS + 4)
#define FS_MAX_LDS (144 * 1024)       // of the 160 KB of a gfx950 CU (the static header lives there too)

// What the kernel needs before it has read a tile's header: the tile grid and the level-0 halo, the same for all tiles (the largest
// any tile's chain asks for), so that a workgroup derives its level-0 box from its block index alone and fetches header, coefficient
// entries and level-0 pixels in ONE round trip (two dependent cold ones were 13 of the kernel's 49 k cycles).
struct FsGeom {
    int nx, ny;
    uint32_t inv_nx, inv_ny;  // floor(2^32 / n) + 1: n * W / nx == umulhi(n * W, inv_nx) while n * W * nx < 2^32
    int hl, hr, ht, hb;       // level-0 halo: columns left / right (multiples of 4), rows above / below the own box
    int tabmax;               // coefficient entries per tile (the blobs are zero-padded to it); they sit at LDS offset 0
    int a0_off;               // LDS byte offset of the level-0 box (behind the entries)
};

static inline __host__ __device__ int fs_cut(int i, int n, int len, uint32_t inv, int mask) {  // tile boundary i of n over [0, len)
    if (i >= n) return len;
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__umulhi((uint32_t)(i * len), inv) & mask;
#else
    return (int)(((unsigned long long)(uint32_t)(i * len) * inv) >> 32) & mask;
#endif
}
// level-0 box of tile (tx, ty): own box and extended box [ex0, ex1) x [ey0, ey1)
static inline __host__ __device__ void fs_box0(const FsGeom& G, int W, int H, int tx, int ty, int& ox0, int& ox1, int& oy0, int& oy1, int& ex0,
                                               int& ex1, int& ey0, int& ey1) {
    ox0 = fs_cut(tx, G.nx, W, G.inv_nx, ~3); ox1 = fs_cut(tx + 1, G.nx, W, G.inv_nx, ~3);
    oy0 = fs_cut(ty, G.ny, H, G.inv_ny, ~0); oy1 = fs_cut(ty + 1, G.ny, H, G.inv_ny, ~0);
    ex0 = ox0 == 0 ? -4 : max(ox0 - G.hl, 0);
    ex1 = ox1 == W ? ((W + 3 + 3) & ~3) : min(ox1 + G.hr, (W + 3) & ~3);
    ey0 = max(oy0 - G.ht, 0);
    ey1 = min(oy1 + G.hb, H);
}

// one level of one tile (FS_BOX_INTS ints in the tile's header)
struct FsBox {
    int ex0, ey0;            // origin of the extended box in level coordinates (ex0 is a multiple of 4)
    int ew, eh;              // its size; ew (a multiple of 4) is the LDS row pitch
    int ox0, oy0, ox1, oy1;  // own box [ox0, ox1) x [oy0, oy1): the pixels this workgroup stores (ox0 a multiple of 4; ox1 too, or the level width)
    int a_off;               // byte offset of the level's pixels in LDS
    int t_off;               // dword offset (inside the tile's coefficient block) of ew column entries followed by eh row entries (levels >= 1)
    int r_off;               // u16 offset (inside the row-sum block) of (oy1 - oy0 + 6) rows x ow4 columns
    int ow4;                 // own width rounded up to a multiple of 4
    uint32_t inv_e, inv_o;   // floor(2^20 / (ew / 4)) + 1 and floor(2^20 / (ow4 / 4)) + 1: task -> (row, quad) without a division
};

static inline int fs_reflect(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// a geometry the kernel does not take: the reason is kept for the probe's message (tools/fuzz_front_single.py tallies them)
static int fs_no(mo_ctx* c, const char* why) { c->fs_why = why; return MO_OK; }

// Builds the per-tile headers and coefficient slices of the current plan (c->d_fs_tab).  Leaves c->fs_ok false - the callers then
// take orb_launch_pyramid / orb_launch_blur - when there is nothing to fuse or a tile does not fit in LDS.
int fs_build(mo_ctx* c) {
    c->fs_ok = false; c->fs_why = "";
    if (c->d_fs_tab) { hipFree(c->d_fs_tab); c->d_fs_tab = nullptr; }
    const Plan& P = c->plan;
    const int nl = P.nlevels;
    if (nl < 2) return fs_no(c, "one level: nothing to chain");
    std::vector<std::vector<int>> xo(nl), xc(nl), yo(nl), yc(nl);
    for (int L = 1; L < nl; L++) {
        mo_linear_coeffs(P.lv[L - 1].w, P.lv[L].w, xo[L], xc[L]);
        mo_linear_coeffs(P.lv[L - 1].h, P.lv[L].h, yo[L], yc[L]);
    }
    const LevelInfo& top = P.lv[nl - 1];
    if (top.w < 8 || top.h < 8) return fs_no(c, "coarsest level smaller than 8 px");
    // tiles of about FS_TILE_W x FS_TILE_H level-0 pixels, and never narrower than 8 pixels on the coarsest level
    const int nx = std::max(1, std::min((P.w + FS_TILE_W / 2) / FS_TILE_W, top.w / 8)), ny = std::max(1, std::min((P.h + FS_TILE_H / 2) / FS_TILE_H, top.h / 8));
    const int ntiles = nx * ny;
    FsGeom G = {};
    G.nx = nx; G.ny = ny;
    G.inv_nx = (uint32_t)(0x100000000ull / (unsigned)nx) + 1u; G.inv_ny = (uint32_t)(0x100000000ull / (unsigned)ny) + 1u;
    if ((unsigned long long)nx * P.w * nx >= (1ull << 32) || (unsigned long long)ny * P.h * ny >= (1ull << 32)) return fs_no(c, "tile cuts beyond the reciprocal's range");
    auto X = [&](int L, int i) { return fs_cut(i, nx, P.lv[L].w, G.inv_nx, ~3); };
    auto Y = [&](int L, int j) { return fs_cut(j, ny, P.lv[L].h, G.inv_ny, ~0); };
    for (int L = 0; L < nl; L++) {  // the reciprocal form is the plain division (the kernel computes level 0's cuts itself)
        for (int i = 0; i <= nx; i++) if (X(L, i) != (i >= nx ? P.lv[L].w : (int)(((long long)i * P.lv[L].w / nx) & ~3ll))) return fs_no(c, "tile cut reciprocal inexact (x)");
        for (int j = 0; j <= ny; j++) if (Y(L, j) != (j >= ny ? P.lv[L].h : (int)((long long)j * P.lv[L].h / ny))) return fs_no(c, "tile cut reciprocal inexact (y)");
    }

    // two passes over the tiles: the first finds the level-0 halo every tile's chain fits in (and the entry count), the second lays
    // the tiles out with that common level-0 box
    std::vector<std::vector<uint32_t>> blobs((size_t)ntiles);
    size_t max_ints = 0, max_lds = 0;
    for (int pass = 0; pass < 2; pass++)
    for (int ty = 0; ty < ny; ty++)
        for (int tx = 0; tx < nx; tx++) {
            FsBox b[MO_MAX_LEVELS] = {};
            for (int L = nl - 1; L >= 0; L--) {
                const int W = P.lv[L].w, H = P.lv[L].h;
                FsBox& v = b[L];
                v.ox0 = X(L, tx); v.ox1 = X(L, tx + 1); v.oy0 = Y(L, ty); v.oy1 = Y(L, ty + 1);
                if (v.ox1 - v.ox0 < 4 || v.oy1 - v.oy0 < 4) return fs_no(c, "own box narrower than 4 px");  // (cannot happen with the tile counts above)
                v.ow4 = ((v.ox1 + 3) & ~3) - v.ox0;
                // what the blur of the own box reads.  Columns: own - 3 .. own + 2 as they are; on the level's left / right border these are
                // the pad columns -3 .. -1 / W .. W + 2, which the kernel fills with the REFLECT_101 pixels once the level is complete, so
                // that every quad of the row pass reads three aligned dwords (a lane on a per-tap reflection path held its whole wavefront
                // 16 x longer: 27 of this kernel's first 49 us, profiles/r04_front_single_stamps.txt).  Rows: reflected per row.
                int lx = v.ox0 - 3, hx = v.ox1 + 2, ly = H, hy = -1;
                if (v.ox0 > 0 && lx < 0) return fs_no(c, "blur halo left of the level on an inner tile");
                if (v.ox1 < W && hx > W - 1) return fs_no(c, "blur halo right of the level on an inner tile");
                for (int y = v.oy0 - 3; y < v.oy1 + 3; y++) { const int r = fs_reflect(y, H); ly = std::min(ly, r); hy = std::max(hy, r); }
                if (L + 1 < nl) {  // what the next level's extended box (all of it is computed, padding columns included) reads
                    const FsBox& n = b[L + 1];
                    const int W1 = P.lv[L + 1].w;
                    const int x0 = std::min(std::max(n.ex0, 0), W1 - 1), x1 = std::min(std::max(n.ex0 + n.ew - 1, 0), W1 - 1);  // (pad columns take the border column's entry)
                    lx = std::min(lx, xo[L + 1][x0]); hx = std::max(hx, std::min(xo[L + 1][x1] + 1, W - 1));
                    const int y0 = n.ey0, y1 = n.ey0 + n.eh - 1;
                    ly = std::min(ly, yo[L + 1][y0]); hy = std::max(hy, std::min(yo[L + 1][y1] + 1, H - 1));
                }
                v.ex0 = lx & ~3; v.ew = ((hx + 1 + 3) & ~3) - v.ex0;  // (-3 & ~3 == -4)
                v.ey0 = ly; v.eh = hy + 1 - ly;
                v.inv_e = (1u << 20) / (uint32_t)(v.ew >> 2) + 1u; v.inv_o = (1u << 20) / (uint32_t)(v.ow4 >> 2) + 1u;
                // (row, quad) of task i as (i * inv) >> 20: exact while i * quads < 2^20
                if ((long long)v.eh * (v.ew >> 2) * (v.ew >> 2) >= (1 << 20) || (long long)(v.oy1 - v.oy0 + 6) * (v.ow4 >> 2) * (v.ow4 >> 2) >= (1 << 20)) return fs_no(c, "task count beyond the reciprocal's range");
            }
            int nt = 0;
            for (int L = 1; L < nl; L++) { b[L].t_off = nt; nt += b[L].ew + ((b[L].eh + 3) & ~3); }  // 16-byte aligned slices (uint4 reads)
            if (pass == 0) {
                FsBox& v = b[0];
                if (v.ox0 > 0) G.hl = std::max(G.hl, v.ox0 - v.ex0);
                if (v.ox1 < P.w) G.hr = std::max(G.hr, v.ex0 + v.ew - v.ox1);
                G.ht = std::max(G.ht, v.oy0 - v.ey0); G.hb = std::max(G.hb, v.ey0 + v.eh - v.oy1);
                G.tabmax = std::max(G.tabmax, nt);
                if (tx == nx - 1 && ty == ny - 1) {
                    G.tabmax = (G.tabmax + 3) & ~3;
                    G.a0_off = G.tabmax * 4;
                }
                continue;
            }
            {  // level 0 takes the common box (fs_box0: what the kernel computes); it must hold what this tile's chain reads
                FsBox& v = b[0];
                int ox0, ox1, oy0, oy1, ex0, ex1, ey0, ey1;
                fs_box0(G, P.w, P.h, tx, ty, ox0, ox1, oy0, oy1, ex0, ex1, ey0, ey1);
                if (ox0 != v.ox0 || ox1 != v.ox1 || oy0 != v.oy0 || oy1 != v.oy1) return fs_no(c, "level-0 cuts differ between host and kernel formula");
                if (ex0 > v.ex0 || ex1 < v.ex0 + v.ew || ey0 > v.ey0 || ey1 < v.ey0 + v.eh) return fs_no(c, "common level-0 box does not hold a tile's chain");
                v.ex0 = ex0; v.ew = ex1 - ex0; v.ey0 = ey0; v.eh = ey1 - ey0;
                v.inv_e = (1u << 20) / (uint32_t)(v.ew >> 2) + 1u;
                if ((long long)v.eh * (v.ew >> 2) * (v.ew >> 2) >= (1 << 20)) return fs_no(c, "level-0 task count beyond the reciprocal's range");
            }
            // LDS layout: coefficient entries (the same size for all tiles) | pixels of every level | blur row sums
            const size_t tab_base = 0;
            size_t off = (size_t)G.a0_off;
            for (int L = 0; L < nl; L++) { b[L].a_off = (int)off; off += ((size_t)b[L].ew * b[L].eh + 15) & ~(size_t)15; }
            const size_t rs_base = off;
            int nr = 0;
            for (int L = 0; L < nl; L++) { b[L].r_off = nr; nr += ((b[L].oy1 - b[L].oy0 + 6) * b[L].ow4 + 3) & ~3; }
            off += (size_t)nr * 2;
            max_lds = std::max(max_lds, off);

            std::vector<uint32_t>& blob = blobs[(size_t)ty * nx + tx];
            blob.assign((size_t)FS_HDR_INTS + G.tabmax, 0u);
            for (int L = 0; L < nl; L++) std::memcpy(&blob[(size_t)L * FS_BOX_INTS], &b[L], sizeof(FsBox));
            blob[MO_MAX_LEVELS * FS_BOX_INTS + 0] = (uint32_t)nt;
            blob[MO_MAX_LEVELS * FS_BOX_INTS + 1] = (uint32_t)tab_base;
            blob[MO_MAX_LEVELS * FS_BOX_INTS + 2] = (uint32_t)rs_base;
            // coefficient entries in k_resize2's packing, source offsets relative to the previous level's extended box; every source
            // index is checked against that box here, so the kernel's LDS reads need no bounds test
            for (int L = 1; L < nl; L++) {
                const FsBox& v = b[L];
                const FsBox& s = b[L - 1];
                const int W = P.lv[L].w, SW = P.lv[L - 1].w, SH = P.lv[L - 1].h;
                uint32_t* e = &blob[(size_t)FS_HDR_INTS + v.t_off];
                for (int i = 0; i < v.ew; i++) {
                    const int j = std::min(std::max(v.ex0 + i, 0), W - 1), o = xo[L][j], o1 = std::min(o + 1, SW - 1), rel = o - s.ex0;
                    if (rel < 0 || rel + (o1 - o) >= s.ew || rel > 0x7FFF) return fs_no(c, "column source index outside the previous level's box");
                    e[i] = (uint32_t)rel | ((uint32_t)(o1 - o) << 15) | ((uint32_t)xc[L][j] << 16);
                }
                for (int i = 0; i < v.eh; i++) {
                    const int j = v.ey0 + i, o = yo[L][j], o1 = std::min(o + 1, SH - 1), rel = o - s.ey0;
                    if (rel < 0 || rel + (o1 - o) >= s.eh || rel > 0x7FFF) return fs_no(c, "row source index outside the previous level's box");
                    e[v.ew + i] = (uint32_t)rel | ((uint32_t)(o1 - o) << 15) | ((uint32_t)yc[L][j] << 16);
                }
            }
            // the blur's reads against the box of the level itself
            for (int L = 0; L < nl; L++) {
                const FsBox& v = b[L];
                const int W = P.lv[L].w, H = P.lv[L].h;
                for (int x = v.ox0 - 4; x < v.ox0 + v.ow4 + 3; x++) if (x < v.ex0 || (x >= v.ex0 + v.ew && x < std::min(v.ox1 + 3, W + 3))) return fs_no(c, "blur columns outside the level's box");
                if ((v.ox0 == 0 || v.ox1 == W) && W < 8) return fs_no(c, "border level narrower than 8 px");
                for (int y = v.oy0 - 3; y < v.oy1 + 3; y++) { const int r = fs_reflect(y, H); if (r < v.ey0 || r >= v.ey0 + v.eh) return fs_no(c, "blur rows outside the level's box"); }
            }
            max_ints = std::max(max_ints, blob.size());
        }
    if (max_lds > FS_MAX_LDS) return fs_no(c, "a tile's boxes exceed the LDS budget");
    const size_t stride = (max_ints + 3) & ~(size_t)3;
    std::vector<uint32_t> all(stride * ntiles, 0u);
    for (int t = 0; t < ntiles; t++) std::copy(blobs[t].begin(), blobs[t].end(), all.begin() + (size_t)t * stride);
    HIPCHK(c, hipMalloc((void**)&c->d_fs_tab, all.size() * sizeof(uint32_t)));
    HIPCHK(c, hipMemcpy(c->d_fs_tab, all.data(), all.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->fs_tiles = ntiles;
    static_assert(sizeof(FsGeom) == sizeof(c->fs_geom), "FsGeom and mo_ctx::fs_geom");
    std::memcpy(c->fs_geom, &G, sizeof(G));
    c->fs_stride = (int)stride;
    c->fs_lds = (int)max_lds;
    c->fs_ok = true;
    return MO_OK;
}

typedef unsigned short fs_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t fs_udot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(fs_u16x2, a), __builtin_bit_cast(fs_u16x2, b), c, false);
}

__global__ __launch_bounds__(FS_NT) void k_front_single(Plan P, FsGeom G, const uint32_t* __restrict__ tab, int stride,
                                                        const uint8_t* __restrict__ gray, uint8_t* __restrict__ pyr,
                                                        uint8_t* __restrict__ blur, int want_blur) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ int s_hdr[FS_HDR_INTS];
    const int tid = threadIdx.x, frame = blockIdx.z, nl = P.nlevels;
    const uint32_t* blob = tab + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * stride;
    uint32_t* s_tab = (uint32_t*)lds;
    // ---- header, coefficient entries of all levels and the level-0 box out of the frame (its geometry from the block index:
    //      fs_box0, the formula the host laid the tile out with), all in one round trip
    {
        int ox0, ox1, oy0, oy1, ex0, ex1, ey0, ey1;
        fs_box0(G, P.w, P.h, (int)blockIdx.x, (int)blockIdx.y, ox0, ox1, oy0, oy1, ex0, ex1, ey0, ey1);
        const uint8_t* g = gray + (size_t)frame * P.w * P.h;
        const int ew = ex1 - ex0, qw = ew >> 2, eh = ey1 - ey0;
        const bool al = (P.w & 3) == 0 && (((size_t)g) & 3) == 0;  // (then a quad lies inside the row or in a pad)
        if (tid < FS_HDR_INTS) s_hdr[tid] = (int)blob[tid];
        for (int i = tid; i < G.tabmax; i += FS_NT) s_tab[i] = blob[FS_HDR_INTS + i];
        // blocks of 128 rows x 64 quads, 4 x 2 per thread, ALL loads of a block issued before its first LDS store: a loop that
        // stores each dword before it loads the next made up to six dependent round trips out of this one
        for (int rb = 0; rb < eh; rb += 128)
            for (int qb = 0; qb < qw; qb += 64) {
                uint32_t v[4][2];
#pragma unroll
                for (int ri = 0; ri < 4; ri++)
#pragma unroll
                    for (int qi = 0; qi < 2; qi++) {
                        const int r = rb + (tid >> 5) + 32 * ri, q = qb + (tid & 31) + 32 * qi, x = ex0 + 4 * q;
                        v[ri][qi] = 0;
                        if (r < eh && q < qw) {
                            const uint8_t* row = g + (size_t)(ey0 + r) * P.w;
                            if (al) {
                                if (x >= 0 && x < P.w) v[ri][qi] = *(const uint32_t*)(row + x);  // (pad quads are filled once the levels are complete)
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; k++) v[ri][qi] |= (uint32_t)row[min(max(x + k, 0), P.w - 1)] << (8 * k);
                            }
                        }
                    }
#pragma unroll
                for (int ri = 0; ri < 4; ri++)
#pragma unroll
                    for (int qi = 0; qi < 2; qi++) {
                        const int r = rb + (tid >> 5) + 32 * ri, q = qb + (tid & 31) + 32 * qi;
                        if (r < eh && q < qw) ((uint32_t*)(lds + G.a0_off + r * ew))[q] = v[ri][qi];
                    }
            }
    }
    __syncthreads();
    const FsBox* box = (const FsBox*)s_hdr;
    const int rs_base = s_hdr[MO_MAX_LEVELS * FS_BOX_INTS + 2];
    // Every pass below is one loop over the (row, quad) tasks of a box, task i -> row (i * inv) >> 20 (host reciprocal): the phases
    // are bound by instruction issue, and a fixed thread -> column mapping left a fifth to a half of the lanes without a task.

    // ---- levels 1 .. : INTER_LINEAR_EXACT from the previous level's box in LDS (k_resize2's arithmetic: row interpolants
    //      (256 - cx) a + cx b < 2^16, then ((256 - cy) h0 + cy h1 + 32768) >> 16); the own part goes to the pyramid slab
    for (int L = 1; L < nl; L++) {
        const FsBox b = box[L];
        const int spitch = box[L - 1].ew;
        const uint8_t* A0 = lds + box[L - 1].a_off;
        uint8_t* A = lds + b.a_off;
        const uint32_t* xt = s_tab + b.t_off;
        const uint32_t* yt = xt + b.ew;
        const LevelInfo lv = P.lv[L];
        uint8_t* dst = pyr + (size_t)frame * P.pyr_stride + lv.off;
        const int qw = b.ew >> 2, n = qw * b.eh;
        for (int i = tid; i < n; i += FS_NT) {
            const int r = (int)(((uint32_t)i * b.inv_e) >> 20), q = i - r * qw;
            const uint32_t ye = yt[r], cy1 = ye >> 16, cy = (256u - cy1) | (cy1 << 16);
            const uint8_t* r0 = A0 + (ye & 0x7FFFu) * spitch;
            const uint8_t* r1 = r0 + ((ye >> 15) & 1u) * spitch;
            const uint4 xe4 = *(const uint4*)(xt + 4 * q);
            const uint32_t xe[4] = {xe4.x, xe4.y, xe4.z, xe4.w};
            uint32_t packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t o = xe[k] & 0x7FFFu, o1 = o + ((xe[k] >> 15) & 1u), c1 = xe[k] >> 16, c0 = 256u - c1;
                const uint32_t h0 = c0 * r0[o] + c1 * r0[o1], h1 = c0 * r1[o] + c1 * r1[o1];
                const uint32_t v = fs_udot2(h0 | (h1 << 16), cy, 32768u);
                packed |= ((v >> 16) & 0xFFu) << (8 * k);
            }
            ((uint32_t*)(A + r * b.ew))[q] = packed;
            const int x = b.ex0 + 4 * q, y = b.ey0 + r;
            // pitch is a multiple of 16 >= w: the <= 3 bytes past w of the last own quad land in row padding
            if (y >= b.oy0 && y < b.oy1 && x >= b.ox0 && x < b.ox1) *(uint32_t*)(dst + (size_t)y * lv.pitch + x) = packed;
        }
        __syncthreads();
    }
    if (!want_blur) return;

    // ---- pad columns of the boxes on a level's left / right border: REFLECT_101 (column -k = column k, column W - 1 + k = W - 1 - k);
    //      one task per (level, row), all levels in one pass (level after level: 5.4 k cycles of LDS round trips for ~ 450 rows)
    {
        const bool left = box[0].ox0 == 0, right = box[0].ox1 == P.w;  // (a tile is on the same border of every level)
        if (left || right) {   // block-uniform
            int total = 0;
            for (int L = 0; L < nl; L++) total += box[L].eh;
            for (int t = tid; t < total; t += FS_NT) {
                int L = 0, r = t;
                while (r >= box[L].eh) { r -= box[L].eh; L++; }
                const int W = P.lv[L].w;
                uint8_t* row = lds + box[L].a_off - box[L].ex0 + r * box[L].ew;  // indexed by level column
                if (left) { row[-1] = row[1]; row[-2] = row[2]; row[-3] = row[3]; }
                if (right) { row[W] = row[W - 2]; row[W + 1] = row[W - 3]; row[W + 2] = row[W - 4]; }
            }
        }
    }
    __syncthreads();

    // ---- 7x7 Gaussian of the own boxes of all levels (k_blur's arithmetic: u16 row sums of 8-bit taps, column sums + 2^15, >> 16,
    //      saturated).  Row pass: own rows - 3 .. + 3 (REFLECT_101 per row), own columns; the 7 taps of four adjacent outputs are
    //      two v_dot4 each on byte windows cut out of three aligned dwords (columns x0 - 4 .. x0 + 7)
    const uint32_t g0 = P.gk[0], g1 = P.gk[1], g2 = P.gk[2], g3 = P.gk[3];
    const uint32_t ta = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24), tb = g2 | (g1 << 8) | (g0 << 16);
    unsigned short* s_rs = (unsigned short*)(lds + rs_base);
    for (int L = 0; L < nl; L++) {
        const FsBox b = box[L];
        const int H = P.lv[L].h;
        const int qw = b.ow4 >> 2, n = (b.oy1 - b.oy0 + 6) * qw;
        const uint8_t* A = lds + b.a_off - b.ex0 + b.ox0 - 4;  // column x0 - 4 of quad 0
        unsigned short* R = s_rs + b.r_off;
        for (int i = tid; i < n; i += FS_NT) {
            const int rr = (int)(((uint32_t)i * b.inv_o) >> 20), q = i - rr * qw;
            int y = b.oy0 - 3 + rr;
            y = y < 0 ? -y : y;
            y = y >= H ? 2 * H - 2 - y : y;
            const uint32_t* pw = (const uint32_t*)(A + (y - b.ey0) * b.ew + 4 * q);
            const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2];
            // output k is centred on byte k + 4 of (w0, w1, w2): taps over bytes k + 1 .. k + 7
            const uint32_t o0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), ta, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), tb, 0, false), false);
            const uint32_t o1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), ta, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), tb, 0, false), false);
            const uint32_t o2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), ta, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), tb, 0, false), false);
            const uint32_t o3 = __builtin_amdgcn_udot4(w1, ta, __builtin_amdgcn_udot4(w2, tb, 0, false), false);
            *(uint2*)(R + rr * b.ow4 + 4 * q) = make_uint2(o0 | (o1 << 16), o2 | (o3 << 16));  // each sum <= 257 * 255
        }
    }
    __syncthreads();
    for (int L = 0; L < nl; L++) {
        const FsBox b = box[L];
        const LevelInfo lv = P.lv[L];
        const int qw = b.ow4 >> 2, n = (b.oy1 - b.oy0) * qw;
        const unsigned short* R = s_rs + b.r_off;
        uint8_t* out = blur + (size_t)frame * P.blur_stride + lv.boff;
        for (int i = tid; i < n; i += FS_NT) {
            const int r = (int)(((uint32_t)i * b.inv_o) >> 20), q = i - r * qw;
            uint32_t s[4] = {1u << 15, 1u << 15, 1u << 15, 1u << 15};
#pragma unroll
            for (int t = 0; t < 7; t++) {
                const uint2 v = *(const uint2*)(R + (r + t) * b.ow4 + 4 * q);
                const uint32_t g = (uint32_t)P.gk[t];
                s[0] += g * (v.x & 0xFFFFu); s[1] += g * (v.x >> 16); s[2] += g * (v.y & 0xFFFFu); s[3] += g * (v.y >> 16);
            }
            uint32_t packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) packed |= min(s[k] >> 16, 255u) << (8 * k);
            // bpitch is a multiple of 16 >= w: the <= 3 bytes past w (sums of pad columns) land in row padding
            *(uint32_t*)(out + (size_t)(b.oy0 + r) * lv.bpitch + b.ox0 + 4 * q) = packed;
        }
    }
}

// pyramid levels 1.. and (want_blur) the blurred levels 0.. of `batch` frames; the caller has checked c->fs_ok
int orb_launch_front_single(mo_ctx* c, const uint8_t* d_gray, int batch, int want_blur) {
    if (!c->fs_ok || !c->d_fs_tab) return mo_fail(c, MO_ERR_ARG, "front_single: no tile table for this plan");
    if (!(c->lds_attr_done & 64u)) {
        HIPCHK(c, hipFuncSetAttribute((const void*)k_front_single, hipFuncAttributeMaxDynamicSharedMemorySize, FS_MAX_LDS));
        c->lds_attr_done |= 64u;
    }
    FsGeom G;
    std::memcpy(&G, c->fs_geom, sizeof(G));
    hipLaunchKernelGGL(k_front_single, dim3((unsigned)G.nx, (unsigned)G.ny, (unsigned)batch), dim3(FS_NT), (size_t)c->fs_lds, c->stream, c->plan, G,
                       c->d_fs_tab, c->fs_stride, d_gray, c->d_pyr, c->d_blur, want_blur);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

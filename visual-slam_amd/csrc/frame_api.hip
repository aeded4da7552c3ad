// frame_api.hip -- the low-latency, one-frame-at-a-time form of the host API: what the reference's Tracker drives (one
// extract_features call per frame, then one pair step against the previous frame: src/orbslam2/tracker.py:87,198-266; the loop of
// src/tests/tester_map.py:57-75).
//
//   mo_detect_single   (behind mo_orb_detect_compute with batch == 1) the image goes through PINNED, device-mapped staging: k_ingest
//                      reads it over PCIe (gray conversion and the flag clear fused), the extraction writes into a RESIDENT RESULT SLOT
//                      of the context, k_pack_out pushes the valid rows back into the pinned buffer - no runtime copy, no fill, no stage
//                      event between the kernels (mo_set_host_timing turns the events on for a breakdown), one synchronisation.
//   mo_pair_frontend   matcher -> (tracking filters) -> two-view stage on two frames named by TOKENS of resident slots (nothing is
//                      uploaded) or by host arrays (uploaded into a slot, which makes them resident for the next call): the whole of
//                      MapInitializer.initialize's device work (initializer.py:67-120) or of Tracker._track_from_last_frame
//                      (tracker.py:214-254) in ONE call with one synchronisation.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"

static inline size_t al(size_t v, size_t a) { return (v + a - 1) & ~(a - 1); }

// (re)allocates the slot arrays when rows of `cap` records do not fit; growing them invalidates every token (the row stride is the cap).
// Callers use c->slot_cap - not their own cap - as the row stride afterwards.
static int slots_reserve(mo_ctx* c, int cap) {
    if (c->d_slot_kps && c->slot_cap >= cap) return MO_OK;
    cap = (int)al((size_t)cap, 16);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    void* old[] = {c->d_slot_kps, c->d_slot_desc, c->d_slot_cnt, c->d_slot_ids};
    for (void* b : old) if (b) hipFree(b);
    c->d_slot_kps = nullptr; c->d_slot_desc = nullptr; c->d_slot_cnt = nullptr; c->d_slot_ids = nullptr;
    c->slot_cap = 0; c->slot_cur = -1;
    for (int s = 0; s < MO_RESULT_SLOTS; s++) { c->slot_token[s] = 0; c->slot_n[s] = 0; }
    const size_t rows = (size_t)MO_RESULT_SLOTS * cap;
    HIPCHK(c, hipMalloc((void**)&c->d_slot_kps, rows * sizeof(mo_keypoint)));
    HIPCHK(c, hipMalloc((void**)&c->d_slot_desc, rows * 32));
    HIPCHK(c, hipMalloc((void**)&c->d_slot_cnt, MO_RESULT_SLOTS * sizeof(int32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_slot_ids, MO_RESULT_SLOTS * sizeof(int32_t)));
    int32_t ids[MO_RESULT_SLOTS];
    for (int s = 0; s < MO_RESULT_SLOTS; s++) ids[s] = s;
    HIPCHK(c, hipMemcpy(c->d_slot_ids, ids, sizeof(ids), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(c->d_slot_cnt, 0, MO_RESULT_SLOTS * sizeof(int32_t)));
    c->slot_cap = cap;
    return MO_OK;
}

static int slot_of(const mo_ctx* c, uint64_t token) {
    if (!token) return -1;
    for (int s = 0; s < MO_RESULT_SLOTS; s++)
        if (c->slot_token[s] == token) return s;
    return -1;
}

// next slot in round-robin order that is not `keep`
static int next_slot(mo_ctx* c, int keep) {
    int s = (c->slot_cur + 1) % MO_RESULT_SLOTS;
    if (s == keep) s = (s + 1) % MO_RESULT_SLOTS;
    c->slot_cur = s;
    c->slot_token[s] = 0;
    return s;
}

// for the other single-frame entry points (api.hip: the grid detector): a free slot with rows for `rows` records -> its index; commit
// names the records written into it (n of them) with a fresh token
int mo_slot_acquire(mo_ctx* c, int rows, int* slot) {
    int rc = slots_reserve(c, rows);
    if (rc) return rc;
    *slot = next_slot(c, -1);
    return MO_OK;
}
uint64_t mo_slot_commit(mo_ctx* c, int slot, int n) {
    c->slot_n[slot] = n;
    c->slot_token[slot] = c->token_next++;
    c->last_token = c->slot_token[slot];
    return c->last_token;
}
uint8_t* mo_stage_dev(mo_ctx* c) {
    void* d = nullptr;
    if (!c->h_stage || hipHostGetDevicePointer(&d, c->h_stage, 0) != hipSuccess) return nullptr;
    return (uint8_t*)d;
}

// device -> pinned host, valid rows only: out = [flags 4 x i32][count, cap, 0, 0][kps cap x 28 B, padded to 16][desc cap x 32 B]
__global__ __launch_bounds__(256) void k_pack_out(const mo_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc, const int32_t* __restrict__ cnt,
                                                  const int* __restrict__ flags, uint8_t* __restrict__ out, int cap) {
    const int total = *cnt, n = min(total, cap);
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) {
        ((int4*)out)[0] = make_int4(flags[0], flags[1], flags[2], flags[3]);
        ((int4*)out)[1] = make_int4(total, cap, 0, 0);
    }
    if (t < cap * 7) {
        if (t < n * 7) ((uint32_t*)(out + 32))[t] = ((const uint32_t*)kps)[t];
        return;
    }
    const int u = t - cap * 7;
    if (desc && u < n * 2) ((uint4*)(out + 32 + (((size_t)cap * 28 + 15) & ~(size_t)15)))[u] = ((const uint4*)desc)[u];
}

// pinned host -> a slot: in = [kps n x 28 B, padded to 16][desc n x 32 B]
__global__ __launch_bounds__(256) void k_unpack_in(const uint8_t* __restrict__ in, int n, mo_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                   int32_t* __restrict__ cnt) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) *cnt = n;
    if (t < n * 7) { ((uint32_t*)kps)[t] = ((const uint32_t*)in)[t]; return; }
    const int u = t - n * 7;
    if (u < n * 2) ((uint4*)desc)[u] = ((const uint4*)(in + (((size_t)n * 28 + 15) & ~(size_t)15)))[u];
}

// device -> pinned host, a contiguous region of 16-byte pieces (the results of the pair step: 50 - 100 KB; a runtime copy of that size
// into the mapped staging buffer took 17 us, this launch 5)
__global__ __launch_bounds__(256) void k_copy_out(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n16) dst[t] = src[t];
}

static uint8_t* stage_dev(mo_ctx* c) { return mo_stage_dev(c); }

void mo_copy_out_launch(mo_ctx* c, const void* d_src, void* h_dst_dev, size_t bytes) {
    const int n16 = (int)((bytes + 15) / 16);
    hipLaunchKernelGGL(k_copy_out, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream, (const uint4*)d_src, (uint4*)h_dst_dev, n16);
}

int mo_detect_single(mo_ctx* c, const mo_orb_params* p, const uint8_t* img, int w, int h, int stride, int ch, mo_keypoint* kps,
                     uint8_t* desc, int cap, int* counts) {
    HIPCHK(c, hipSetDevice(c->device));
    HostClock clk(c);
    if (!img) return mo_fail(c, MO_ERR_ARG, "img is NULL");
    if (ch != 1 && ch != 3) return mo_fail(c, MO_ERR_ARG, "ch must be 1 (gray) or 3 (BGR)");
    if (stride < w * ch) return mo_fail(c, MO_ERR_ARG, "stride smaller than a row");
    if (cap < 1) return mo_fail(c, MO_ERR_ARG, "cap must be >= 1");
    int rc = mo_build_plan(c, p, w, h, 1);  // validates sizes before any staging
    if (rc) return rc;
    if ((rc = slots_reserve(c, cap))) return rc;
    const int scap = c->slot_cap;  // row stride of the slot arrays (>= cap: the kernels write up to scap rows, the caller gets up to cap)
    const size_t row = (size_t)w * ch, in_bytes = row * h, o_out = al(in_bytes, 256);
    const size_t o_kps = 32, o_desc = o_kps + al((size_t)scap * 28, 16), out_bytes = o_desc + (size_t)scap * 32;
    if ((rc = mo_host_stage(c, o_out + out_bytes))) return rc;
    if ((rc = mo_reserve(c, c->d_in, c->d_in_bytes, al((size_t)w * h, 256)))) return rc;
    uint8_t* hs = c->h_stage;
    uint8_t* hs_dev = stage_dev(c);
    if (!hs_dev) return mo_fail(c, MO_ERR_HIP, "the pinned staging buffer is not mapped into the device");
    if ((size_t)stride == row) std::memcpy(hs, img, in_bytes);
    else for (int y = 0; y < h; y++) std::memcpy(hs + (size_t)y * row, img + (size_t)y * stride, row);
    const int slot = next_slot(c, -1);
    mo_keypoint* d_k = c->d_slot_kps + (size_t)slot * scap;
    uint8_t* d_d = c->d_slot_desc + (size_t)slot * scap * 32;
    int32_t* d_n = c->d_slot_cnt + slot;
    mo_stage_begin(c);
    if ((rc = orb_launch_ingest(c, hs_dev, w, h, ch, c->d_in, mo_host_flags(c)))) return rc;
    if ((rc = mo_run_extract(c, p, c->d_in, w, h, 1, d_k, desc ? d_d : nullptr, scap, d_n, 2))) return rc;
    hipLaunchKernelGGL(k_pack_out, dim3((unsigned)((scap * 9 + 255) / 256)), dim3(256), 0, c->stream, d_k, desc ? d_d : (const uint8_t*)nullptr, d_n,
                       mo_host_flags(c), hs_dev + o_out, scap);
    HIPCHK(c, hipGetLastError());
    mo_stage_mark(c, "d2h");
    clk.enqueued();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    clk.waited();
    const int* ho = (const int*)(hs + o_out);
    const int fl = ho[0], total = ho[4];
    counts[0] = total;  // MO_ERR_CAPACITY: counts already holds the size a retry needs
    if (fl & 1) { c->tie_overflow = true; c->tie_levels = ho[1]; return mo_fail(c, MO_ERR_CAPACITY, "internal per-level keypoint capacity exceeded (response ties)"); }
    if ((fl & 2) || total > cap) return mo_fail(c, MO_ERR_CAPACITY, "more keypoints than cap; counts holds the required sizes");
    const int n = std::min(std::max(total, 0), cap);
    if (n > 0) {
        std::memcpy(kps, hs + o_out + o_kps, (size_t)n * sizeof(mo_keypoint));
        if (desc) std::memcpy(desc, hs + o_out + o_desc, (size_t)n * 32);
    }
    c->slot_n[slot] = n;
    c->last_token = 0;
    if (desc) {  // (a detect-only result has no descriptors: nothing a pair step could use)
        c->slot_token[slot] = c->token_next++;
        c->last_token = c->slot_token[slot];
    }
    return MO_OK;
}

extern "C" int mo_last_token(mo_ctx* c, uint64_t* token) {
    if (!c || !token) return MO_ERR_ARG;
    *token = c->last_token;
    return MO_OK;
}

// a frame of the pair -> its slot: the token's when it is alive, else the host arrays are uploaded into the next free slot
static int resolve_frame(mo_ctx* c, const mo_frame_ref* f, int keep, uint8_t* hs, uint8_t* hs_dev, size_t stage_off, int* slot_out, int* n_out,
                         uint64_t* token_out) {
    int s = slot_of(c, f->token);
    if (s >= 0) { *slot_out = s; *n_out = c->slot_n[s]; *token_out = f->token; return MO_OK; }
    if (f->n < 0 || (f->n > 0 && (!f->kps || !f->desc))) return mo_fail(c, MO_ERR_ARG, "frame token is stale and no host arrays were given");
    if (f->n > c->slot_cap) return mo_fail(c, MO_ERR_CAPACITY, "frame has more keypoints than the resident slots hold");
    s = next_slot(c, keep);
    const int n = f->n;
    if (n > 0) {
        std::memcpy(hs + stage_off, f->kps, (size_t)n * 28);
        std::memcpy(hs + stage_off + al((size_t)n * 28, 16), f->desc, (size_t)n * 32);
    }
    hipLaunchKernelGGL(k_unpack_in, dim3((unsigned)((n * 9 + 255) / 256 + 1)), dim3(256), 0, c->stream, hs_dev + stage_off, n,
                       c->d_slot_kps + (size_t)s * c->slot_cap, c->d_slot_desc + (size_t)s * c->slot_cap * 32, c->d_slot_cnt + s);
    HIPCHK(c, hipGetLastError());
    c->slot_n[s] = n;
    c->slot_token[s] = c->token_next++;
    *slot_out = s; *n_out = n; *token_out = c->slot_token[s];
    return MO_OK;
}

extern "C" int mo_pair_frontend(mo_ctx* c, const mo_frame_ref* f1, const mo_frame_ref* f2, const mo_pair_params* pp, mo_pair_out* out) {
    if (!c) return MO_ERR_ARG;
    if (!f1 || !f2 || !pp || !out) return mo_fail(c, MO_ERR_ARG, "NULL argument");
    if (pp->mode != MO_MODE_INIT && pp->mode != MO_MODE_TRACK) return mo_fail(c, MO_ERR_ARG, "mode must be MO_MODE_INIT or MO_MODE_TRACK");
    if (pp->n_hyp < 0) return mo_fail(c, MO_ERR_ARG, "n_hyp must be >= 0 (0 = matcher only)");
    HIPCHK(c, hipSetDevice(c->device));
    HostClock clk(c);
    out->n_sel = 0; out->n_good = 0; out->n1 = 0; out->n2 = 0; out->token1 = 0; out->token2 = 0;
    for (int i = 0; i < 9; i++) { out->R[i] = NAN; out->E[i] = NAN; }
    for (int i = 0; i < 3; i++) out->t[i] = NAN;
    const bool track = pp->mode == MO_MODE_TRACK;
    int rc;
    // slots large enough for both frames (growing them invalidates the tokens: the host arrays then have to be there)
    int s1 = slot_of(c, f1->token), s2 = slot_of(c, f2->token);
    int need_cap = c->slot_cap;
    if (s1 < 0) need_cap = std::max(need_cap, (int)al((size_t)std::max(f1->n, 1), 16));
    if (s2 < 0) need_cap = std::max(need_cap, (int)al((size_t)std::max(f2->n, 1), 16));
    if (need_cap != c->slot_cap && (rc = slots_reserve(c, need_cap))) return rc;
    const int cap = c->slot_cap;
    const size_t up_bytes = al((size_t)cap * 28, 16) + (size_t)cap * 32;
    // device outputs (one region, ordered so that each mode's results are contiguous for ONE copy back)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes, 256); return o; };
    const size_t o_X = take((size_t)cap * 3 * sizeof(float)), o_ran = take(cap), o_midx = take((size_t)cap * 2 * sizeof(int32_t)),
                 o_mdist = take((size_t)cap * 2 * sizeof(int32_t)), o_mpass = take(cap), o_sel = take((size_t)cap * 2 * sizeof(int32_t)),
                 o_seld = take((size_t)cap * sizeof(int32_t)), o_seln = take(sizeof(int32_t)), o_pose = take(12 * sizeof(double)),
                 o_E = take(9 * sizeof(double)), o_inl = take(cap), o_np = take(sizeof(int32_t)), out_end = off;
    if ((rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, out_end))) return rc;
    if ((rc = mo_host_stage(c, std::max(2 * al(up_bytes, 256), out_end)))) return rc;
    uint8_t* hs = c->h_stage;
    uint8_t* hs_dev = stage_dev(c);
    if (!hs_dev) return mo_fail(c, MO_ERR_HIP, "the pinned staging buffer is not mapped into the device");
    uint8_t* b = (uint8_t*)c->d_tmp;
    mo_stage_begin(c);
    int n1 = 0, n2 = 0;
    s1 = slot_of(c, f1->token);  // (again: a reallocation above dropped the tokens)
    if ((rc = resolve_frame(c, f1, slot_of(c, f2->token), hs, hs_dev, 0, &s1, &n1, &out->token1))) return rc;
    if ((rc = resolve_frame(c, f2, s1, hs, hs_dev, al(up_bytes, 256), &s2, &n2, &out->token2))) return rc;
    out->n1 = n1; out->n2 = n2;
    if (n1 == 0 || n2 == 0) { HIPCHK(c, hipStreamSynchronize(c->stream)); return MO_OK; }  // (matcher.py:57-61: no matches)
    mo_stage_mark(c, "h2d");
    c->flags_cur = mo_host_flags(c);
    HIPCHK(c, hipMemsetAsync(mo_host_flags(c), 0, 4 * sizeof(int), c->stream));
    const int32_t* qf = c->d_slot_ids + s1;
    const int32_t* tf = c->d_slot_ids + s2;
    int32_t* d_midx = (int32_t*)(b + o_midx);
    int32_t* d_mdist = (int32_t*)(b + o_mdist);
    uint8_t* d_mpass = b + o_mpass;
    if ((rc = match_launch_pairs(c, c->d_slot_desc, c->d_slot_desc, (size_t)cap * 32, (size_t)cap * 32, c->d_slot_cnt, qf, tf, 0, 0, 1, cap, pp->ratio,
                                 d_midx, d_mdist, d_mpass)))
        return rc;
    mo_stage_mark(c, "match_knn2_ratio");
    const bool pose = pp->n_hyp > 0;
    if (track) {
        if ((rc = track_select_launch(c, c->d_slot_kps, c->d_slot_cnt, qf, tf, d_midx, d_mdist, d_mpass, cap, 1, pp->w, pp->h, pp->disp_frac,
                                      (int32_t*)(b + o_sel), (int32_t*)(b + o_seld), (int32_t*)(b + o_seln))))
            return rc;
        mo_stage_mark(c, "track_filters");
    }
    if (pose) {
        TwoViewArgs a;
        std::memset(&a, 0, sizeof(a));
        a.n_pairs = 1; a.cap = cap; a.n_hyp = pp->n_hyp;
        for (int i = 0; i < 9; i++) a.K[i] = pp->K[i];
        a.thr_px = pp->thr_px; a.seed = pp->seed; a.pair_base = pp->pair_index;
        a.d_kps = c->d_slot_kps; a.d_counts = c->d_slot_cnt; a.d_match_idx = d_midx; a.d_match_pass = d_mpass;
        a.d_qf = qf; a.d_tf = tf;
        if (track) { a.d_sel = (int32_t*)(b + o_sel); a.d_sel_n = (int32_t*)(b + o_seln); }
        a.d_pose = (double*)(b + o_pose); a.d_E = (double*)(b + o_E); a.d_points = (float*)(b + o_X); a.d_inlier = b + o_inl;
        a.d_ransac = track ? nullptr : b + o_ran;
        a.d_n_points = (int32_t*)(b + o_np);
        if ((rc = twoview_launch(c, a))) return rc;
        mo_stage_mark(c, "two_view");
    }
    const bool want_match = out->match_idx || out->match_dist || out->match_pass;
    const size_t from = !track ? (pose ? o_X : o_midx) : (want_match ? o_midx : o_sel);
    const size_t to = !track && !pose ? o_sel : out_end;
    {
        const int n16 = (int)((to - from) / 16);  // (every offset is a multiple of 256)
        hipLaunchKernelGGL(k_copy_out, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream, (const uint4*)(b + from), (uint4*)(hs_dev + from), n16);
        HIPCHK(c, hipGetLastError());
    }
    mo_stage_mark(c, "d2h");
    clk.enqueued();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    clk.waited();
    if (out->match_idx) std::memcpy(out->match_idx, hs + o_midx, (size_t)n1 * 2 * sizeof(int32_t));
    if (out->match_dist) std::memcpy(out->match_dist, hs + o_mdist, (size_t)n1 * 2 * sizeof(int32_t));
    if (out->match_pass) std::memcpy(out->match_pass, hs + o_mpass, (size_t)n1);
    const double* hp = (const double*)(hs + o_pose);
    const double* hE = (const double*)(hs + o_E);
    const int32_t np = pose ? *(const int32_t*)(hs + o_np) : 0;
    if (track) {
        int32_t ns = *(const int32_t*)(hs + o_seln);
        ns = std::min(std::max(ns, 0), n1);
        out->n_sel = ns;
        if (ns > 0) {
            if (out->sel_idx) std::memcpy(out->sel_idx, hs + o_sel, (size_t)ns * 2 * sizeof(int32_t));
            if (out->sel_dist) std::memcpy(out->sel_dist, hs + o_seld, (size_t)ns * sizeof(int32_t));
            if (out->inlier) {
                const int32_t* sl = (const int32_t*)(hs + o_sel);
                const uint8_t* mask = hs + o_inl;
                for (int j = 0; j < ns; j++) out->inlier[j] = pose && ns >= 8 ? mask[(size_t)sl[2 * j]] : 0;  // the pose mask is indexed by query keypoint
            }
        }
        if (pose && ns >= 8) {  // tracker.py:234: fewer than 8 matches -> tracking fails (R, t stay NaN)
            for (int i = 0; i < 9; i++) { out->R[i] = hp[i]; out->E[i] = hE[i]; }
            for (int i = 0; i < 3; i++) out->t[i] = hp[9 + i];
            out->n_good = np;
        }
        return MO_OK;
    }
    if (pose) {
        for (int i = 0; i < 9; i++) { out->R[i] = hp[i]; out->E[i] = hE[i]; }
        for (int i = 0; i < 3; i++) out->t[i] = hp[9 + i];
        out->n_good = np;
        if (out->inlier) std::memcpy(out->inlier, hs + o_inl, (size_t)n1);
        if (out->ransac) std::memcpy(out->ransac, hs + o_ran, (size_t)n1);
        if (out->X) std::memcpy(out->X, hs + o_X, (size_t)n1 * 3 * sizeof(float));
    }
    return MO_OK;
}

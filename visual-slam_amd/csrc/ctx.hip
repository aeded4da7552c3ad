// ctx.hip -- context lifetime, the per-(w,h,params) plan, work-buffer sizing and stage timing.
//
// The plan holds everything the kernels need that OpenCV derives on the host inside
// ORB_Impl::detectAndCompute (level scales and sizes, per-level quotas, the circular-patch umax table,
// the INTER_LINEAR_EXACT coefficient tables, the quantised Gaussian taps).  It is host arithmetic done
// once per image size, in the same float/double expressions cv2 uses (reference call site:
// src/orbslam2/extractor.py:38-48,65).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <cstdlib>

#include "common.h"

int mo_fail(mo_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

static std::string g_create_err;

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

extern "C" int mo_abi_version(void) { return MO_ABI_VERSION; }

extern "C" int mo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" mo_ctx* mo_create(int device, int max_w, int max_h, int max_batch) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_err = "mo_create: no HIP device available (this library has no CPU fallback)";
        return nullptr;
    }
    if (device < 0 || device >= n || max_w < 64 || max_h < 64 || max_w > 4095 || max_h > 4095 || max_batch < 1) {
        g_create_err = "mo_create: bad arguments (need 64 <= w,h <= 4095, batch >= 1, valid device)";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        g_create_err = "mo_create: hipSetDevice failed";
        return nullptr;
    }
    mo_ctx* c = new mo_ctx();
    c->device = device;
    c->max_w = max_w; c->max_h = max_h; c->max_batch = max_batch;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        g_create_err = "mo_create: hipStreamCreate failed";
        delete c;
        return nullptr;
    }
    c->stream = c->own_stream;
    // the one environment switch of the library: VSLAM_AMD_MATCHER=mfma selects the opt-in matrix-core matcher (identical results;
    // north_star prescribes XOR + popcount as the default, bench.py times the opt-in beside it)
    if (const char* e = getenv("VSLAM_AMD_MATCHER")) c->match_mode = std::strcmp(e, "mfma") == 0 ? 1 : 0;
    // hipEventDisableSystemFence: these events order work of ONE device (kernel boundaries already release / acquire at agent
    // scope); the default system-scope fence of an event record writes the L2 back and cost 6 - 17 us of idle GPU at every stage mark
    // (kernel trace: gaps only where an event sits between two kernels), 0.06 ms of a 2.2 ms step (profiles/r02_ab_event_fence.txt).
    const unsigned evf = (unsigned)hipEventDisableSystemFence;
    for (TimingSet& t : c->tsets)
        for (int i = 0; i <= MO_NSTAGES; i++) hipEventCreateWithFlags(&t.ev[i], evf);
    if (hipMalloc((void**)&c->d_flags, 8 * sizeof(int)) != hipSuccess) {
        g_create_err = "mo_create: hipMalloc failed";
        delete c;
        return nullptr;
    }
    hipMemset(c->d_flags, 0, 8 * sizeof(int));
    c->flags_cur = c->d_flags;
    return c;
}

static void free_plan_buffers(mo_ctx* c) {
    for (int L = 0; L < MO_MAX_LEVELS; L++) {
        ResizeTab& t = c->rtab[L];
        if (t.xpk) hipFree(t.xpk);
        t = ResizeTab();
    }
    void* bufs[] = {c->d_pyr, c->d_blur, c->d_cand, c->d_strip_cnt, c->d_scratch, c->d_fin, c->d_fin_cnt, c->d_tile_tab[0], c->d_tile_tab[1], c->d_strip_tab, c->d_dtile_tab};
    c->d_dtile_tab = nullptr; c->n_dtiles = 0;
    if (c->d_fs_tab) hipFree(c->d_fs_tab);
    c->d_fs_tab = nullptr; c->fs_ok = false;
    for (void* b : bufs) if (b) hipFree(b);
    c->d_tile_tab[0] = c->d_tile_tab[1] = nullptr; c->d_strip_tab = nullptr; c->n_strip_tab = 0;
    c->d_pyr = c->d_blur = nullptr; c->d_cand = nullptr; c->d_strip_cnt = nullptr; c->d_scratch = nullptr;
    c->d_fin = nullptr; c->d_fin_cnt = nullptr;
    c->batch_alloc = 0;
    c->plan_valid = false;
}

extern "C" void mo_destroy(mo_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    mo_comm_destroy(c);
    free_plan_buffers(c);
    void* bufs[] = {c->d_in, c->d_gray, c->d_flags, c->d_kps, c->d_desc, c->d_counts, c->d_mq, c->d_mt,
                    c->d_midx, c->d_mdist, c->d_mpass, c->d_match_part, c->d_tv, c->d_tmp, c->d_pair_frames, c->d_dtodo, c->d_comm_cnt,
                    c->d_slot_kps, c->d_slot_desc, c->d_slot_cnt, c->d_slot_ids, c->d_track_keys, c->d_stream_pts};
    for (void* b : bufs) if (b) hipFree(b);
    if (c->h_stage) hipHostFree(c->h_stage);
    for (TimingSet& t : c->tsets) {
        for (int i = 0; i <= MO_NSTAGES; i++) if (t.ev[i]) hipEventDestroy(t.ev[i]);
    }
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" const char* mo_last_error(mo_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

extern "C" int mo_set_stream(mo_ctx* c, void* s) {
    if (!c) return MO_ERR_ARG;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return MO_OK;
}

// The HIP null stream has the handle 0, which mo_set_stream reads as "the context's own stream": a caller whose work sits on the
// null stream (torch's default stream) selects it with this call, so that the library's launches are ordered with that work.
extern "C" int mo_set_stream_null(mo_ctx* c) {
    if (!c) return MO_ERR_ARG;
    c->stream = nullptr;
    return MO_OK;
}

extern "C" int mo_sync(mo_ctx* c) {
    if (!c) return MO_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MO_OK;
}

void mo_stage_begin(mo_ctx* c) {
    c->tcur = (c->tcur + 1) % MO_TIMING_SLOTS;
    TimingSet& t = c->tsets[c->tcur];
    t.n_stages = 0;
    if (c->timing) hipEventRecord(t.ev[0], c->stream);
}

void mo_stage_mark(mo_ctx* c, const char* name) {
    TimingSet& t = c->tsets[c->tcur];
    if (!c->timing || t.n_stages >= MO_NSTAGES) return;
    t.names[t.n_stages] = name;
    t.n_stages++;
    hipEventRecord(t.ev[t.n_stages], c->stream);
}

extern "C" int mo_stage_times_back(mo_ctx* c, int back, const char*** names, float* ms, int cap) {
    if (!c) return MO_ERR_ARG;
    if (back < 0 || back >= MO_TIMING_SLOTS) return mo_fail(c, MO_ERR_ARG, "mo_stage_times_back: back outside the ring of event sets");
    TimingSet& t = c->tsets[((c->tcur - back) % MO_TIMING_SLOTS + MO_TIMING_SLOTS) % MO_TIMING_SLOTS];
    if (t.n_stages == 0) return 0;
    HIPCHK(c, hipEventSynchronize(t.ev[t.n_stages]));
    int n = std::min(cap, t.n_stages);
    for (int i = 0; i < n; i++) {
        float v = 0;
        hipEventElapsedTime(&v, t.ev[i], t.ev[i + 1]);
        ms[i] = v;
    }
    t.names[t.n_stages] = nullptr;
    if (names) *names = t.names;
    return n;
}

extern "C" int mo_stage_times(mo_ctx* c, const char*** names, float* ms, int cap) { return mo_stage_times_back(c, 0, names, ms, cap); }

// INTER_LINEAR_EXACT coefficient table of one axis (interpolationLinear<ufixedpoint16>::getCoeffs):
// offset + the weight of the right/lower neighbour in 1/256 units (left weight = 256 - c1).
void mo_linear_coeffs(int srcsize, int dstsize, std::vector<int>& ofs, std::vector<int>& c1) {
    ofs.assign(dstsize, 0);
    c1.assign(dstsize, 0);
    double inv_scale = (double)dstsize / (double)srcsize;
    double scale = 1.0 / inv_scale;
    int minofst = 0, maxofst = dstsize;
    for (int val = 0; val < dstsize; val++) {
        double fval = scale * ((double)val + 0.5) - 0.5;
        int ival = (int)std::floor(fval);
        if (ival >= 0 && srcsize > 1) {
            if (ival < srcsize - 1) {
                ofs[val] = ival;
                c1[val] = cv_round_d((fval - (double)ival) * 256.0);
            } else {
                ofs[val] = srcsize - 1;
                maxofst = std::min(maxofst, val);
            }
        } else {
            minofst = std::max(minofst, val + 1);
        }
    }
    for (int val = 0; val < dstsize; val++) {
        if (val < minofst) { ofs[val] = 0; c1[val] = 0; }
        if (val >= maxofst) { ofs[val] = srcsize - 1; c1[val] = 0; }
    }
}

static bool params_equal(const mo_orb_params& a, const mo_orb_params& b) {
    return std::memcmp(&a, &b, sizeof(a)) == 0;
}

int mo_build_plan(mo_ctx* c, const mo_orb_params* p, int w, int h, int batch) {
    if (!p) return mo_fail(c, MO_ERR_ARG, "params is NULL");
    if (w < 64 || h < 64 || w > c->max_w || h > c->max_h)
        return mo_fail(c, MO_ERR_ARG, "image size outside the context's max_w/max_h (or < 64)");
    if (batch < 1 || batch > c->max_batch) return mo_fail(c, MO_ERR_ARG, "batch outside 1..max_batch");
    if (p->nlevels < 1 || p->nlevels > MO_MAX_LEVELS) return mo_fail(c, MO_ERR_ARG, "nlevels must be 1..12");
    if (p->first_level != 0 || p->wta_k != 2 || p->score_type != 0 || p->patch_size != 31)
        return mo_fail(c, MO_ERR_UNSUPPORTED,
                       "only firstLevel=0, WTA_K=2, HARRIS_SCORE, patchSize=31 (the reference's values) are built");
    if (p->edge_threshold < 19 || p->edge_threshold > 1024)
        return mo_fail(c, MO_ERR_UNSUPPORTED, "edge_threshold must be >= 19 (descriptor radius)");
    if (p->nfeatures < 1 || !(p->scale_factor > 1.0f) || p->scale_factor > 2.0f)
        return mo_fail(c, MO_ERR_ARG, "nfeatures >= 1, 1 < scale_factor <= 2");
    if (p->select_order != MO_ORDER_LIBSTDCXX && p->select_order != MO_ORDER_MSVC)
        return mo_fail(c, MO_ERR_ARG, "select_order must be MO_ORDER_LIBSTDCXX or MO_ORDER_MSVC");

    const bool same_key = c->plan_valid && params_equal(c->plan_params, *p) && c->plan.w == w && c->plan.h == h;
    const bool same = same_key && !c->fin_slack_dirty;
    if (!same_key)   // another image size / parameter set: the grown slots were that one's
        for (int L = 0; L < MO_MAX_LEVELS; L++) c->fin_slack[L] = 1;
    if (same && batch <= c->batch_alloc) return MO_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_plan_buffers(c);

    Plan& P = c->plan;
    std::memset(&P, 0, sizeof(P));
    P.w = w; P.h = h; P.nlevels = p->nlevels;
    P.edge_threshold = p->edge_threshold;
    P.fast_threshold = std::min(std::max(p->fast_threshold, 0), 255);
    P.select_order = p->select_order;
    P.nfeatures = p->nfeatures;

    // per-level quotas (computeKeyPoints)
    int nl = p->nlevels;
    {
        float factor = (float)(1.0 / (double)p->scale_factor);
        float nd = p->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
        int sum = 0;
        for (int L = 0; L < nl - 1; L++) {
            P.lv[L].quota = cv_round_f(nd);
            sum += P.lv[L].quota;
            nd *= factor;
        }
        P.lv[nl - 1].quota = std::max(p->nfeatures - sum, 0);
    }
    // umax of the radius-15 disc
    {
        int umax[MO_HALF_PATCH + 2];
        int vmax = (int)std::floor(MO_HALF_PATCH * std::sqrt(2.f) / 2 + 1);
        int vmin = (int)std::ceil(MO_HALF_PATCH * std::sqrt(2.f) / 2);
        for (int v = 0; v <= vmax; ++v) umax[v] = cv_round_d(std::sqrt((double)MO_HALF_PATCH * MO_HALF_PATCH - v * v));
        for (int v = MO_HALF_PATCH, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        for (int v = 0; v <= MO_HALF_PATCH; v++) P.umax[v] = umax[v];
    }
    // Gaussian 7 taps, sigma 2, quantised to 8 fractional bits (sepFilter2D 8u path)
    {
        double k[7], sum = 0;
        for (int i = 0; i < 7; i++) {
            double x = 2.0 * i - 6.0;
            k[i] = std::exp(x * x * (-0.125 / 4.0));
            sum += k[i];
        }
        double mul1 = 1.0 / sum;
        for (int i = 0; i < 7; i++) P.gk[i] = cv_round_f((float)(k[i] * mul1) * 256.f);
    }

    int et = p->edge_threshold;
    int pyr_off = 0, blur_off = 0, strip_base = 0, cand_off = 0, fin_off = 0, scr_off = 0;
    for (int L = 0; L < nl; L++) {
        LevelInfo& v = P.lv[L];
        v.scale = (float)std::pow((double)p->scale_factor, (double)L);
        float inv_scale = 1.0f / v.scale;
        v.w = cv_round_f((float)w * inv_scale);
        v.h = cv_round_f((float)h * inv_scale);
        if (v.w < 1 || v.h < 1) return mo_fail(c, MO_ERR_ARG, "pyramid level collapses to zero size; reduce nlevels");
        if (L == 0) { v.pitch = w; v.off = 0; }
        else {
            v.pitch = align_up(v.w, 16);
            v.off = pyr_off;
            pyr_off += align_up(v.pitch * v.h, 256);
        }
        v.bpitch = align_up(v.w, 16);
        v.boff = blur_off;
        blur_off += align_up(v.bpitch * v.h, 256);
        if (v.w <= 2 * et || v.h <= 2 * et) { v.bx0 = v.by0 = et; v.bw = v.bh = 0; }
        else { v.bx0 = et; v.by0 = et; v.bw = v.w - 2 * et; v.bh = v.h - 2 * et; }
        v.inv_bw = v.bw > 1 ? 0xFFFFFFFFu / (uint32_t)v.bw + 1u : 0u;
        // (a context for one or two frames at a time: shorter strips, more workgroups - MO_STRIP_ROWS_LATENCY in common.h)
        v.strip_rows = c->max_batch <= 2 ? MO_STRIP_ROWS_LATENCY : MO_STRIP_ROWS;
        while (v.strip_rows > 1 && v.strip_rows * v.bw > 16384) v.strip_rows /= 2;
        // (short strips only while the level stays below the selection kernel's strip limit; mo_create caps frames at 4095 px = 2017 two-row
        //  strips, so this matters only if that cap is raised)
        while (v.strip_rows < MO_STRIP_ROWS && v.bh > 0 && (v.bh + v.strip_rows - 1) / v.strip_rows > SEL_MAXSTRIPS && 2 * v.strip_rows * v.bw <= 16384)
            v.strip_rows *= 2;
        if (v.bw > 16384) return mo_fail(c, MO_ERR_UNSUPPORTED, "level too wide");
        v.nstrips = v.bh > 0 ? (v.bh + v.strip_rows - 1) / v.strip_rows : 0;
        v.strip_cap = ((v.strip_rows + 1) / 2) * ((v.bw + 1) / 2);
        v.strip_base = strip_base;
        strip_base += v.nstrips;
        v.cand_off = cand_off;
        v.cand_cap = v.nstrips * v.strip_cap;
        cand_off += v.cand_cap;
        v.fin_off = fin_off;
        v.fin_cap = (int)std::max<long long>(1, std::min<long long>(v.cand_cap, (4ll * v.quota + 256) * c->fin_slack[L]));
        fin_off += v.fin_cap;
        v.scr_off = scr_off;
        // u64 records B + u32 records A + u16 partner positions + u64 ballots, in u64 units
        scr_off += v.cand_cap + (v.cand_cap + 1) / 2 + (v.cand_cap / 2 + 8) / 4 + 2 + v.cand_cap / 64 + 12;
    }
    P.pyr_stride = std::max(pyr_off, 256);
    P.blur_stride = blur_off;
    P.strips_per_frame = std::max(strip_base, 1);
    P.cand_stride = std::max(cand_off, 1);
    P.fin_stride = fin_off;
    c->scratch_stride = (size_t)scr_off;

    // resize tables
    for (int L = 1; L < nl; L++) {
        std::vector<int> xo, xc, yo, yc;
        mo_linear_coeffs(P.lv[L - 1].w, P.lv[L].w, xo, xc);
        mo_linear_coeffs(P.lv[L - 1].h, P.lv[L].h, yo, yc);
        const int dw = P.lv[L].w, dh = P.lv[L].h, wp = ((dw + 63) & ~63) + 64, hp = ((dh + 63) & ~63) + 64;  // + 64: the tiling may start at a margin
        auto pack = [](const std::vector<int>& o, const std::vector<int>& c1, int srcsize, int padded) {
            std::vector<uint32_t> t((size_t)padded);
            for (int i = 0; i < padded; i++) {
                const int j = std::min(i, (int)o.size() - 1), o1 = std::min(o[j] + 1, srcsize - 1);
                t[i] = (uint32_t)o[j] | ((uint32_t)(o1 - o[j]) << 15) | ((uint32_t)c1[j] << 16);
            }
            return t;
        };
        const std::vector<uint32_t> xp = pack(xo, xc, P.lv[L - 1].w, wp), yp = pack(yo, yc, P.lv[L - 1].h, hp);
        // k_resize2 takes the source bytes of 4 adjacent output columns from ONE 8-byte window: right neighbour of the last
        // column - offset of the first <= 7.  Always true below a level ratio of 2; rounded level widths can put the ratio a
        // little above it at scale_factor 2 (333 -> 166), and such a level keeps the gather kernel
        bool window_ok = true;
        for (int x = 0; x < dw && window_ok; x++) {
            const int xl = std::min(x + 3, dw - 1);
            window_ok = std::min(xo[xl] + 1, P.lv[L - 1].w - 1) - xo[x] <= 7;
        }
        c->rtab[L].two_pass_ok = window_ok;
        size_t n = (size_t)wp + hp + 2 * dw + 2 * dh;
        int* d = nullptr;
        HIPCHK(c, hipMalloc((void**)&d, n * sizeof(int)));
        ResizeTab& t = c->rtab[L];
        t.xpk = (uint32_t*)d; t.ypk = t.xpk + wp;
        t.xofs = d + wp + hp; t.xc1 = t.xofs + dw; t.yofs = t.xc1 + dw; t.yc1 = t.yofs + dh;
        HIPCHK(c, hipMemcpy(t.xpk, xp.data(), xp.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(t.ypk, yp.data(), yp.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(t.xofs, xo.data(), xo.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(t.xc1, xc.data(), xc.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(t.yofs, yo.data(), yo.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(t.yc1, yc.data(), yc.size() * sizeof(int), hipMemcpyHostToDevice));
    }

    // work buffers for `batch` frames
    size_t B = (size_t)batch;
    HIPCHK(c, hipMalloc((void**)&c->d_pyr, B * P.pyr_stride));
    HIPCHK(c, hipMalloc((void**)&c->d_blur, B * P.blur_stride));
    HIPCHK(c, hipMalloc((void**)&c->d_cand, B * P.cand_stride * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_strip_cnt, B * P.strips_per_frame * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->d_scratch, B * c->scratch_stride * sizeof(uint64_t)));
    if (hipMalloc((void**)&c->d_fin, B * P.fin_stride * sizeof(FinalKp)) != hipSuccess)
        return mo_fail(c, MO_ERR_HIP, "hipMalloc of the final-keypoint slots failed: " + std::to_string(B * P.fin_stride * sizeof(FinalKp)) +
                                          " bytes (" + std::to_string(batch) + " frames x " + std::to_string(P.fin_stride) + " slots; slots grow with response ties)");
    HIPCHK(c, hipMalloc((void**)&c->d_fin_cnt, B * MO_MAX_LEVELS * sizeof(int)));
    {
        const int rc_fs = fs_build(c);  // single-frame pyramid + blur kernel: tile boxes of this plan
        if (rc_fs) return rc_fs;
    }
    c->batch_alloc = batch;
    c->plan_params = *p;
    c->plan_valid = true;
    c->fin_slack_dirty = false;
    return MO_OK;
}

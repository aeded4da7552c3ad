// api.hip -- extern "C" entry points declared in include/vslam_amd.h.
// Host entry points stage their inputs into HBM, run the same device pipeline the batched mode uses, and
// copy the results back; there is no CPU implementation behind any of them.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

// Flag words: kernels raise bits in c->flags_cur[0].  mo_dev_* calls point it at words 0..3 (accumulated until mo_dev_status reads
// and clears them); host entry points point it at words 4..7, which they clear before and check after their own kernels - a host
// call between mo_dev_frontend_batch and mo_dev_status neither erases nor inherits the pending device-call bits.
static inline int* host_flags(mo_ctx* c) { return mo_host_flags(c); }

extern "C" int mo_set_host_timing(mo_ctx* c, int on) {
    if (!c) return MO_ERR_ARG;
    c->host_timing = on != 0;
    return MO_OK;
}

extern "C" int mo_host_times(mo_ctx* c, double us[4]) {
    if (!c || !us) return MO_ERR_ARG;
    for (int i = 0; i < 4; i++) us[i] = c->host_us[i];
    return MO_OK;
}

static bool grow_fin_slots(mo_ctx* c, int levels);

static int check_flags(mo_ctx* c) {
    int f[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(f, host_flags(c), sizeof(f), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (f[0] & 1) { c->tie_overflow = true; c->tie_levels = f[1]; return mo_fail(c, MO_ERR_CAPACITY, "internal per-level keypoint capacity exceeded (response ties)"); }
    if (f[0] & 2) return mo_fail(c, MO_ERR_CAPACITY, "more keypoints than cap; counts holds the required sizes");
    return MO_OK;
}

extern "C" int mo_dev_status(mo_ctx* c, int32_t flags[4]) {
    if (!c) return MO_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int f[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(f, c->d_flags, sizeof(f), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0, 4 * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (flags) { flags[0] = f[0]; flags[1] = f[1]; flags[2] = flags[3] = 0; }
    if (f[0] & 1) grow_fin_slots(c, f[1]);  // response ties overflowed a level's slot: the next call of the same shape rebuilds its plan with larger ones
    if (f[0] & 15) return mo_fail(c, MO_ERR_CAPACITY, "a capacity flag was raised by a mo_dev_* call (see mo_dev_status in vslam_amd.h)");
    return MO_OK;
}

// device pipeline on frames already resident as dense gray [batch][h][w]
int mo_run_extract(mo_ctx* c, const mo_orb_params* p, const uint8_t* d_gray, int w, int h, int batch,
                   mo_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_counts, int host_call) {
    int rc = mo_build_plan(c, p, w, h, batch);
    if (rc) return rc;
    if (cap < 1) return mo_fail(c, MO_ERR_ARG, "cap must be >= 1");
    // host calls check their own flag words before they return; mo_dev_* calls accumulate theirs until mo_dev_status
    c->flags_cur = host_call ? host_flags(c) : c->d_flags;
    if (host_call == 1) HIPCHK(c, hipMemsetAsync(host_flags(c), 0, 4 * sizeof(int), c->stream));  // (2: the upload kernel cleared them)
    if (c->poison >= 0) {  // mo_dbg_set_poison (tests): whatever the margins skip must never reach a result
        HIPCHK(c, hipMemsetAsync(c->d_pyr, c->poison, (size_t)c->batch_alloc * c->plan.pyr_stride, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_blur, c->poison, (size_t)c->batch_alloc * c->plan.blur_stride, c->stream));
    }
    if (host_call) mo_stage_mark(c, "h2d");  // (the host call opened its event set before the upload)
    else mo_stage_begin(c);
    // margins of the levels nothing in this pipeline reads (see orb_launch_blur / orb_launch_pyramid)
    const int blur_margin = (c->plan.edge_threshold - 19) & ~3, pyr_margin = std::max(blur_margin - 4, 0);
    // (one frame of a host call: the finest level's FAST + selection on a second stream beside the pyramid and the other levels was
    //  measured in round 4 - no gain, level 1's chain is as long as level 0's: profiles/r04_ab_single_split.txt)
    if (batch <= MO_FS_MAX_BATCH && c->fs_ok) {
        // one or two frames: pyramid and blur in ONE launch whose workgroups chain the levels of their own tile through LDS
        // (front_single.hip; the seven dependent resize launches + the blur were 62 of 202 us of a one-frame call)
        if ((rc = orb_launch_front_single(c, d_gray, batch, d_desc != nullptr))) return rc;
        mo_stage_mark(c, "pyramid");
        if (d_desc) mo_stage_mark(c, "blur");  // (inside the same launch: the stage keeps its name, its time is in "pyramid")
    } else {
        if ((rc = orb_launch_pyramid(c, d_gray, batch, c->plan.nlevels, pyr_margin))) return rc;
        mo_stage_mark(c, "pyramid");
        // the Gaussian blur only depends on the pyramid: in line, right behind it (an aux-stream fork beside FAST + selection gained <= 1 %
        // in the BATCHED mode in rounds 1 - 2 and was retired there: profiles/r02_ab_serial_blur.txt)
        if (d_desc) {
            if ((rc = orb_launch_blur(c, d_gray, batch, c->plan.nlevels, blur_margin))) return rc;
            mo_stage_mark(c, "blur");
        }
    }
    if ((rc = orb_launch_fast(c, d_gray, batch))) return rc;
    mo_stage_mark(c, "fast_nms");
    if ((rc = orb_launch_select(c, d_gray, batch))) return rc;
    mo_stage_mark(c, "select_harris");
    if ((rc = orb_launch_describe(c, d_gray, batch, d_kps, d_desc, cap, d_counts))) return rc;
    mo_stage_mark(c, "angle_rbrief");
    return MO_OK;
}

// ORBExtractor.distribute_keypoints (reference extractor.py:85-144, the path Tracker.process_frame takes: tracker.py:87) for a whole
// batch of frames resident in HBM: min-eigenvalue map, the 64 per-cell corner picks, KeyPoint(x, y, 31) records of the corners
// orb.compute keeps, blur of level 0, descriptors at angle -1 - one launch per step for all frames.  d_kps / d_desc / d_counts receive
// the KEPT keypoints (record i belongs to descriptor row i, which is what the match / pose stages need); the reference's own list (all
// corners, misaligned with the rows whenever a corner lies in the border band) is in the optional d_grid_* outputs.
static int run_grid_extract(mo_ctx* c, const mo_orb_params* p, const mo_batch_io* io) {
    const int w = io->w, h = io->h, batch = io->batch, cap = io->cap;
    int rc = mo_build_plan(c, p, w, h, batch);
    if (rc) return rc;
    if (p->nfeatures < 64) return mo_fail(c, MO_ERR_ARG, "n_features must be >= 64 (8x8 grid)");
    if (cap < 1) return mo_fail(c, MO_ERR_ARG, "cap must be >= 1");
    const int per_cell = p->nfeatures / 64, slots = 64 * per_cell;
    const size_t B = (size_t)batch;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_eig = take(B * w * h * sizeof(float)), o_xy = take(B * slots * 2 * sizeof(float)), o_n = take(B * 66 * sizeof(int)),
                 o_kb = take(B * 65 * sizeof(int32_t));
    if ((rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, off))) return rc;
    uint8_t* b = (uint8_t*)c->d_tmp;
    float* d_eig = (float*)(b + o_eig);
    float* d_xy = io->d_grid_xy ? io->d_grid_xy : (float*)(b + o_xy);
    int* d_n = io->d_grid_n ? io->d_grid_n : (int*)(b + o_n);
    c->flags_cur = c->d_flags;
    mo_stage_begin(c);
    if ((rc = gftt_launch(c, io->d_gray, w, h, p->nfeatures, d_eig, d_xy, d_n, batch))) return rc;
    int32_t* d_kb = (int32_t*)(b + o_kb);
    if ((rc = gftt_records_launch(c, d_xy, d_n, per_cell, w, h, p->edge_threshold, io->d_kps, io->d_grid_kept, cap, io->d_counts, batch, d_kb))) return rc;
    mo_stage_mark(c, "grid_good_features");
    const int blur_margin = (c->plan.edge_threshold - 19) & ~3;
    if ((rc = orb_launch_blur(c, io->d_gray, batch, 1, blur_margin))) return rc;  // the records all sit on octave 0
    mo_stage_mark(c, "blur");
    // descriptors out of one blurred LDS tile per grid cell; cells too large for the tile (frames beyond ~ 720 x 480) take the
    // one-wavefront-per-keypoint kernel that samples global memory
    rc = orb_launch_describe_cells(c, io->d_kps, d_kb, io->d_desc, cap, batch);
    if (rc == MO_ERR_UNSUPPORTED) rc = orb_launch_describe_given(c, io->d_gray, io->d_kps, cap, io->d_desc, io->d_counts, batch, 1);
    if (rc) return rc;
    mo_stage_mark(c, "compute");
    return MO_OK;
}

// copy host images (any stride / 1 or 3 channels) into a dense device gray batch; returns the device pointer
static int stage_images(mo_ctx* c, const uint8_t* img, int w, int h, int stride, int ch, int batch, const uint8_t** d_gray) {
    if (!img) return mo_fail(c, MO_ERR_ARG, "img is NULL");
    if (ch != 1 && ch != 3) return mo_fail(c, MO_ERR_ARG, "ch must be 1 (gray) or 3 (BGR)");
    if (stride < w * ch) return mo_fail(c, MO_ERR_ARG, "stride smaller than a row");
    size_t row = (size_t)w * ch, frame = row * h;
    int rc = mo_reserve(c, c->d_in, c->d_in_bytes, frame * batch);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy2DAsync(c->d_in, row, img, (size_t)stride, row, (size_t)h * batch, hipMemcpyHostToDevice, c->stream));
    if (ch == 3) {
        rc = mo_reserve(c, c->d_gray, c->d_gray_bytes, (size_t)w * h * batch);
        if (rc) return rc;
        if ((rc = orb_launch_gray(c, c->d_in, w, h, batch, c->d_gray))) return rc;
        *d_gray = c->d_gray;
    } else {
        *d_gray = c->d_in;
    }
    return MO_OK;
}

static int reserve_out(mo_ctx* c, int batch, int cap) {
    if (c->d_kps && c->out_cap >= cap && c->out_batch >= batch) return MO_OK;
    // drop all three and the recorded sizes first: a failed hipMalloc below must not leave a size check that passes
    // with freed pointers
    if (c->d_kps) hipFree(c->d_kps);
    if (c->d_desc) hipFree(c->d_desc);
    if (c->d_counts) hipFree(c->d_counts);
    c->d_kps = nullptr; c->d_desc = nullptr; c->d_counts = nullptr;
    c->out_cap = 0; c->out_batch = 0;
    size_t n = (size_t)batch * cap;
    HIPCHK(c, hipMalloc((void**)&c->d_kps, n * sizeof(mo_keypoint)));
    HIPCHK(c, hipMalloc((void**)&c->d_desc, n * 32));
    HIPCHK(c, hipMalloc((void**)&c->d_counts, (size_t)batch * sizeof(int)));
    c->out_cap = cap; c->out_batch = batch;
    return MO_OK;
}

// pinned host staging owned by the context (small host-API transfers: one copy each way and one synchronisation instead of a blocking
// round trip per pageable array).  Mapped + coherent: k_ingest / k_pack_out read and write it from the device.
int mo_host_stage(mo_ctx* c, size_t bytes) {
    if (c->h_stage_bytes >= bytes) return MO_OK;
    if (c->h_stage) { HIPCHK(c, hipStreamSynchronize(c->stream)); hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_bytes = 0; }
    bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    HIPCHK(c, hipHostMalloc((void**)&c->h_stage, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    c->h_stage_bytes = bytes;
    return MO_OK;
}
static int host_stage(mo_ctx* c, size_t bytes) { return mo_host_stage(c, bytes); }

// true when an overflowed level's final-keypoint slot can still grow (then the plan is invalidated so that the next call rebuilds it).
// retainBest keeps EVERY element that ties with the quota boundary, so a level of a periodic synthetic pattern can keep all its
// candidates; the slots are sized for 4 quota + 256 and the levels named in `levels` (bit L) grow eightfold, up to the level's
// candidate capacity - the other levels, and other image sizes / parameter sets, keep their sizes.
static bool grow_fin_slots(mo_ctx* c, int levels) {
    if (!c->plan_valid) return false;
    if (!levels) levels = (1 << c->plan.nlevels) - 1;  // (no level named: an older flag word; grow them all)
    bool grown = false;
    for (int L = 0; L < c->plan.nlevels; L++)
        if (((levels >> L) & 1) && c->plan.lv[L].fin_cap < c->plan.lv[L].cand_cap && c->fin_slack[L] < (1 << 24)) { c->fin_slack[L] *= 8; grown = true; }
    c->fin_slack_dirty = c->fin_slack_dirty || grown;
    return grown;
}

static int detect_compute_once(mo_ctx* c, const mo_orb_params* p, const uint8_t* img, int w, int h, int stride, int ch,
                               int batch, mo_keypoint* kps, uint8_t* desc, int cap, int* counts);

extern "C" int mo_orb_detect_compute(mo_ctx* c, const mo_orb_params* p, const uint8_t* img, int w, int h, int stride, int ch,
                                     int batch, mo_keypoint* kps, uint8_t* desc, int cap, int* counts) {
    if (!c) return MO_ERR_ARG;
    if (!kps || !counts) return mo_fail(c, MO_ERR_ARG, "kps/counts is NULL");
    for (;;) {  // (at most 8 rounds: the slots reach the candidate capacity, where no overflow is possible)
        c->tie_overflow = false; c->tie_levels = 0;
        const int rc = detect_compute_once(c, p, img, w, h, stride, ch, batch, kps, desc, cap, counts);
        if (rc != MO_ERR_CAPACITY || !c->tie_overflow || !grow_fin_slots(c, c->tie_levels)) return rc;
    }
}

static int detect_compute_once(mo_ctx* c, const mo_orb_params* p, const uint8_t* img, int w, int h, int stride, int ch,
                               int batch, mo_keypoint* kps, uint8_t* desc, int cap, int* counts) {
    HIPCHK(c, hipSetDevice(c->device));
    // one frame (the drop-in classes, a Tracker): pinned staging both ways, a resident result slot, one synchronisation (frame_api.hip)
    if (batch == 1 && (size_t)w * h * ch + (size_t)cap * 60 <= (size_t)64 << 20) return mo_detect_single(c, p, img, w, h, stride, ch, kps, desc, cap, counts);
    HostClock clk(c);
    const uint8_t* d_gray = nullptr;
    int rc = mo_build_plan(c, p, w, h, batch);  // validates sizes before any staging
    if (rc) return rc;
    mo_stage_begin(c);
    if ((rc = stage_images(c, img, w, h, stride, ch, batch, &d_gray))) return rc;
    if ((rc = reserve_out(c, batch, cap))) return rc;
    if ((rc = mo_run_extract(c, p, d_gray, w, h, batch, c->d_kps, desc ? c->d_desc : nullptr, cap, c->d_counts, true))) return rc;
    // Small results (the single-frame calls of the drop-in classes): flags, counts, keypoints and descriptors travel into ONE pinned
    // staging buffer behind one synchronisation; copies into the caller's pageable arrays cost a blocking round trip each (four
    // per call before: counts, flags, keypoints, descriptors).
    const size_t n_rows = (size_t)batch * cap, o_cnt = 4 * sizeof(int), o_kps = (o_cnt + (size_t)batch * sizeof(int) + 15) & ~(size_t)15;
    const size_t o_desc = o_kps + n_rows * sizeof(mo_keypoint), total = o_desc + (desc ? n_rows * 32 : 0);
    if (total <= (size_t)2 << 20) {
        if ((rc = host_stage(c, total))) return rc;
        uint8_t* hs = c->h_stage;
        HIPCHK(c, hipMemcpyAsync(hs, host_flags(c), 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hs + o_cnt, c->d_counts, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hs + o_kps, c->d_kps, n_rows * sizeof(mo_keypoint), hipMemcpyDeviceToHost, c->stream));
        if (desc) HIPCHK(c, hipMemcpyAsync(hs + o_desc, c->d_desc, n_rows * 32, hipMemcpyDeviceToHost, c->stream));
        mo_stage_mark(c, "d2h");
        clk.enqueued();
        HIPCHK(c, hipStreamSynchronize(c->stream));
        clk.waited();
        std::memcpy(counts, hs + o_cnt, (size_t)batch * sizeof(int));  // MO_ERR_CAPACITY: counts already holds the sizes a retry needs
        const int fl = ((const int*)hs)[0];
        if (fl & 1) { c->tie_overflow = true; c->tie_levels = ((const int*)hs)[1]; return mo_fail(c, MO_ERR_CAPACITY, "internal per-level keypoint capacity exceeded (response ties)"); }
        if (fl & 2) return mo_fail(c, MO_ERR_CAPACITY, "more keypoints than cap; counts holds the required sizes");
        for (int f = 0; f < batch; f++) {
            const int n = std::min(counts[f], cap);
            if (n <= 0) continue;
            std::memcpy(kps + (size_t)f * cap, hs + o_kps + (size_t)f * cap * sizeof(mo_keypoint), (size_t)n * sizeof(mo_keypoint));
            if (desc) std::memcpy(desc + (size_t)f * cap * 32, hs + o_desc + (size_t)f * cap * 32, (size_t)n * 32);
        }
        return MO_OK;
    }
    HIPCHK(c, hipMemcpyAsync(counts, c->d_counts, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    rc = check_flags(c);  // MO_ERR_CAPACITY: counts already holds the sizes a retry needs
    if (rc) return rc;
    for (int f = 0; f < batch; f++) {
        int n = std::min(counts[f], cap);
        if (n <= 0) continue;
        HIPCHK(c, hipMemcpyAsync(kps + (size_t)f * cap, c->d_kps + (size_t)f * cap, (size_t)n * sizeof(mo_keypoint),
                                 hipMemcpyDeviceToHost, c->stream));
        if (desc)
            HIPCHK(c, hipMemcpyAsync(desc + (size_t)f * cap * 32, c->d_desc + (size_t)f * cap * 32, (size_t)n * 32,
                                     hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MO_OK;
}

extern "C" int mo_orb_compute(mo_ctx* c, const mo_orb_params* p, const uint8_t* img, int w, int h, int stride, int ch,
                              const mo_keypoint* kps_in, int n_in, int32_t* kept_idx, uint8_t* desc, int* n_out) {
    if (!c) return MO_ERR_ARG;
    if (!p || !n_out || (n_in > 0 && (!kps_in || !kept_idx || !desc))) return mo_fail(c, MO_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    *n_out = 0;
    // host-side list logic of Feature2D::compute / ORB_Impl::detectAndCompute(useProvidedKeypoints):
    // level count from the octaves, border filter on the full image, regroup by octave when unsorted
    int nlevels = 0;
    bool sorted = true;
    for (int i = 0; i < n_in; i++) {
        int L = kps_in[i].octave;
        if (L < 0) return mo_fail(c, MO_ERR_ARG, "keypoint octave < 0");
        if (i > 0 && L < kps_in[i - 1].octave) sorted = false;
        nlevels = std::max(nlevels, L);
    }
    nlevels++;
    if (nlevels > MO_MAX_LEVELS) return mo_fail(c, MO_ERR_UNSUPPORTED, "keypoint octave >= 12");
    std::vector<int> keep;
    int et = p->edge_threshold;
    if (!(et > 0 && (h <= et * 2 || w <= et * 2))) {
        for (int i = 0; i < n_in; i++) {
            int x = (int)lrintf(kps_in[i].x), y = (int)lrintf(kps_in[i].y);
            if (et > 0 && !(x >= et && x < w - et && y >= et && y < h - et)) continue;
            keep.push_back(i);
        }
    }
    if (!sorted) {
        std::vector<int> re;
        for (int L = 0; L < nlevels; L++)
            for (int i : keep)
                if (kps_in[i].octave == L) re.push_back(i);
        keep.swap(re);
    }
    int n = (int)keep.size();
    for (int i = 0; i < n; i++) kept_idx[i] = keep[i];
    *n_out = n;
    if (n == 0) return MO_OK;
    mo_orb_params pp = *p;
    if (nlevels > pp.nlevels) pp.nlevels = nlevels;
    int rc = mo_build_plan(c, &pp, w, h, 1);
    if (rc) return rc;
    const uint8_t* d_gray = nullptr;
    if ((rc = stage_images(c, img, w, h, stride, ch, 1, &d_gray))) return rc;
    std::vector<mo_keypoint> kk(n);
    for (int i = 0; i < n; i++) kk[i] = kps_in[keep[i]];
    size_t kb = (size_t)n * sizeof(mo_keypoint);
    if ((rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, kb + (size_t)n * 32))) return rc;
    mo_keypoint* d_k = (mo_keypoint*)c->d_tmp;
    uint8_t* d_d = (uint8_t*)c->d_tmp + kb;
    HIPCHK(c, hipMemcpyAsync(d_k, kk.data(), kb, hipMemcpyHostToDevice, c->stream));
    mo_stage_begin(c);
    if ((rc = orb_launch_pyramid(c, d_gray, 1, nlevels, 0))) return rc;
    if ((rc = orb_launch_blur(c, d_gray, 1, nlevels, 0))) return rc;
    if ((rc = orb_launch_describe_given(c, d_gray, d_k, n, d_d))) return rc;
    mo_stage_mark(c, "compute");
    HIPCHK(c, hipMemcpyAsync(desc, d_d, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MO_OK;
}

extern "C" int mo_dev_undistort(mo_ctx* c, const uint8_t* d_src, int w, int h, int ch, int batch, const double K[9],
                                const double dist[5], uint8_t* d_dst) {
    if (!c) return MO_ERR_ARG;
    if (!d_src || !d_dst || !K || !dist || d_src == d_dst) return mo_fail(c, MO_ERR_ARG, "NULL argument (or in-place)");
    if (w < 1 || h < 1 || batch < 1 || (ch != 1 && ch != 3)) return mo_fail(c, MO_ERR_ARG, "bad size / channel count");
    if (!(K[0] != 0.0) || !(K[4] != 0.0)) return mo_fail(c, MO_ERR_ARG, "focal length is zero");
    HIPCHK(c, hipSetDevice(c->device));
    return undistort_launch(c, d_src, d_dst, w, h, ch, batch, K, dist);
}

extern "C" int mo_undistort(mo_ctx* c, const uint8_t* img, int w, int h, int stride, int ch, const double K[9], const double dist[5],
                            uint8_t* out) {
    if (!c) return MO_ERR_ARG;
    if (!img || !out || !K || !dist) return mo_fail(c, MO_ERR_ARG, "NULL argument");
    if (w < 1 || h < 1 || (ch != 1 && ch != 3) || stride < w * ch) return mo_fail(c, MO_ERR_ARG, "bad size / stride / channel count");
    if (!(K[0] != 0.0) || !(K[4] != 0.0)) return mo_fail(c, MO_ERR_ARG, "focal length is zero");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t row = (size_t)w * ch, frame = row * h;
    int rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, 2 * frame + 256);
    if (rc) return rc;
    uint8_t* d_src = (uint8_t*)c->d_tmp;
    uint8_t* d_dst = d_src + ((frame + 255) & ~(size_t)255);
    HIPCHK(c, hipMemcpy2DAsync(d_src, row, img, (size_t)stride, row, (size_t)h, hipMemcpyHostToDevice, c->stream));
    mo_stage_begin(c);
    if ((rc = undistort_launch(c, d_src, d_dst, w, h, ch, 1, K, dist))) return rc;
    mo_stage_mark(c, "undistort");
    HIPCHK(c, hipMemcpyAsync(out, d_dst, frame, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MO_OK;
}

static int gftt_run(mo_ctx* c, const uint8_t* img, int w, int h, int stride, int ch, int n_features, float* xy, int* n_out,
                    float* eig_out) {
    if (w < 64 || h < 64 || w > c->max_w || h > c->max_h) return mo_fail(c, MO_ERR_ARG, "image size outside the context limits");
    if (n_features < 64) return mo_fail(c, MO_ERR_ARG, "n_features must be >= 64 (8x8 grid)");
    const uint8_t* d_gray = nullptr;
    int rc = stage_images(c, img, w, h, stride, ch, 1, &d_gray);
    if (rc) return rc;
    const int per_cell = n_features / 64;
    size_t eig_b = (size_t)w * h * sizeof(float), xy_b = (size_t)64 * per_cell * 2 * sizeof(float);
    if ((rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, eig_b + xy_b + 64 * sizeof(int) + 64))) return rc;
    float* d_eig = (float*)c->d_tmp;
    float* d_xy = (float*)((uint8_t*)c->d_tmp + eig_b);
    int* d_n = (int*)((uint8_t*)d_xy + xy_b);
    c->flags_cur = host_flags(c);
    HIPCHK(c, hipMemsetAsync(host_flags(c), 0, 4 * sizeof(int), c->stream));
    mo_stage_begin(c);
    if ((rc = gftt_launch(c, d_gray, w, h, n_features, d_eig, d_xy, d_n))) return rc;
    mo_stage_mark(c, "grid_good_features");
    if (eig_out) HIPCHK(c, hipMemcpyAsync(eig_out, d_eig, eig_b, hipMemcpyDeviceToHost, c->stream));
    std::vector<float> hxy((size_t)64 * per_cell * 2);
    int hn[64];
    HIPCHK(c, hipMemcpyAsync(hxy.data(), d_xy, xy_b, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hn, d_n, sizeof(hn), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int n = 0;
    if (xy)
        for (int cell = 0; cell < 64; cell++)
            for (int i = 0; i < hn[cell]; i++, n++) {
                xy[2 * n] = hxy[((size_t)cell * per_cell + i) * 2];
                xy[2 * n + 1] = hxy[((size_t)cell * per_cell + i) * 2 + 1];
            }
    if (n_out) *n_out = n;
    return MO_OK;
}

extern "C" int mo_orb_grid_good_features(mo_ctx* c, const uint8_t* img, int w, int h, int stride, int ch, int n_features,
                                         float* xy, int* n_out) {
    if (!c) return MO_ERR_ARG;
    if (!img || !xy || !n_out) return mo_fail(c, MO_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    return gftt_run(c, img, w, h, stride, ch, n_features, xy, n_out, nullptr);
}

// ORBExtractor.distribute_keypoints in one call (reference extractor.py:85-144: the grid corners, KeyPoint(x, y, 31) each, then
// orb.compute on all of them): ONE upload of the image, the corner lists stay on the device, the records of the corners cv2 keeps are
// built there (k_gftt_records), blur of level 0 + descriptors, ONE synchronisation.  Same outputs as mo_orb_grid_good_features followed
// by mo_orb_compute on KeyPoint(x, y, 31) records.
extern "C" int mo_orb_grid_detect_compute(mo_ctx* c, const mo_orb_params* p, const uint8_t* img, int w, int h, int stride, int ch,
                                          int n_features, float* xy, int* n_xy, int32_t* kept_idx, uint8_t* desc, int* n_kept) {
    if (!c) return MO_ERR_ARG;
    if (!p || !img || !xy || !n_xy || !kept_idx || !desc || !n_kept) return mo_fail(c, MO_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    HostClock clk(c);
    *n_xy = 0; *n_kept = 0;
    c->last_token = 0;
    if (w < 64 || h < 64 || w > c->max_w || h > c->max_h) return mo_fail(c, MO_ERR_ARG, "image size outside the context limits");
    if (n_features < 64) return mo_fail(c, MO_ERR_ARG, "n_features must be >= 64 (8x8 grid)");
    if (ch != 1 && ch != 3) return mo_fail(c, MO_ERR_ARG, "ch must be 1 (gray) or 3 (BGR)");
    if (stride < w * ch) return mo_fail(c, MO_ERR_ARG, "stride smaller than a row");
    int rc = mo_build_plan(c, p, w, h, 1);
    if (rc) return rc;
    const int per_cell = n_features / 64, slots = 64 * per_cell;
    // Like the single-frame ORB call (frame_api.hip): the image through pinned staging and the upload kernel, the KEPT keypoints
    // (KeyPoint(x, y, 31) records, aligned with the descriptor rows) and their descriptors in a resident result slot - mo_last_token
    // names them for mo_pair_frontend -, everything the caller gets in one region that one small kernel pushes back, one synchronisation.
    int slot = 0;
    if ((rc = mo_slot_acquire(c, slots, &slot))) return rc;
    const int scap = c->slot_cap;
    mo_keypoint* d_rec = c->d_slot_kps + (size_t)slot * scap;
    uint8_t* d_desc = c->d_slot_desc + (size_t)slot * scap * 32;
    int32_t* d_cnt = c->d_slot_cnt + slot;
    const size_t rowb = (size_t)w * ch, in_bytes = rowb * h, eig_b = (size_t)w * h * sizeof(float), xy_b = (size_t)slots * 2 * sizeof(float);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_eig = take(eig_b), o_out = off, o_fl = take(16), o_xy = take(xy_b), o_n = take(66 * sizeof(int)), o_kept = take((size_t)slots * sizeof(int32_t)),
                 o_desc = take((size_t)slots * 32), out_end = off;
    if ((rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, out_end))) return rc;
    if ((rc = mo_reserve(c, c->d_in, c->d_in_bytes, ((size_t)w * h + 255) & ~(size_t)255))) return rc;
    const size_t h_out = (in_bytes + 255) & ~(size_t)255;
    if ((rc = mo_host_stage(c, h_out + (out_end - o_out)))) return rc;
    uint8_t* hs = c->h_stage;
    uint8_t* hs_dev = mo_stage_dev(c);
    if (!hs_dev) return mo_fail(c, MO_ERR_HIP, "the pinned staging buffer is not mapped into the device");
    if ((size_t)stride == rowb) std::memcpy(hs, img, in_bytes);
    else for (int y = 0; y < h; y++) std::memcpy(hs + (size_t)y * rowb, img + (size_t)y * stride, rowb);
    uint8_t* b = (uint8_t*)c->d_tmp;
    float* d_eig = (float*)(b + o_eig); float* d_xy = (float*)(b + o_xy); int* d_n = (int*)(b + o_n);
    int32_t* d_kept = (int32_t*)(b + o_kept);
    c->flags_cur = host_flags(c);
    mo_stage_begin(c);
    if ((rc = orb_launch_ingest(c, hs_dev, w, h, ch, c->d_in, host_flags(c)))) return rc;
    mo_stage_mark(c, "h2d");
    const uint8_t* d_gray = c->d_in;
    if ((rc = gftt_launch(c, d_gray, w, h, n_features, d_eig, d_xy, d_n))) return rc;
    if ((rc = gftt_records_launch(c, d_xy, d_n, per_cell, w, h, p->edge_threshold, d_rec, d_kept, scap, d_cnt, 1))) return rc;  // (totals: d_n[64], d_n[65])
    mo_stage_mark(c, "grid_good_features");
    if ((rc = orb_launch_blur(c, d_gray, 1, 1, 0))) return rc;  // the records all sit on octave 0
    if ((rc = orb_launch_describe_given(c, d_gray, d_rec, slots, d_desc, d_n + 65, 1, 0))) return rc;
    mo_stage_mark(c, "compute");
    // descriptors of the kept corners and the flag words into the result region, then the region into the pinned buffer
    HIPCHK(c, hipMemcpyAsync(b + o_desc, d_desc, (size_t)slots * 32, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_fl, host_flags(c), 16, hipMemcpyDeviceToDevice, c->stream));
    mo_copy_out_launch(c, b + o_out, hs_dev + h_out, out_end - o_out);
    HIPCHK(c, hipGetLastError());
    mo_stage_mark(c, "d2h");
    clk.enqueued();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    clk.waited();
    const uint8_t* ho = hs + h_out;
    const int* hn = (const int*)(ho + (o_n - o_out));
    const float* hxy = (const float*)(ho + (o_xy - o_out));
    const int fl = *(const int*)(ho + (o_fl - o_out));
    if (fl & 2) return mo_fail(c, MO_ERR_CAPACITY, "more grid corners than the result slot holds");
    int n = 0;
    for (int cell = 0; cell < 64; cell++)
        for (int i = 0; i < std::min(hn[cell], per_cell); i++, n++) {
            xy[2 * n] = hxy[((size_t)cell * per_cell + i) * 2];
            xy[2 * n + 1] = hxy[((size_t)cell * per_cell + i) * 2 + 1];
        }
    const int nk = hn[65];
    if (hn[64] != n || nk < 0 || nk > n) return mo_fail(c, MO_ERR_HIP, "grid corner counts disagree between host and device");
    *n_xy = n; *n_kept = nk;
    std::memcpy(kept_idx, ho + (o_kept - o_out), (size_t)nk * sizeof(int32_t));
    std::memcpy(desc, ho + (o_desc - o_out), (size_t)nk * 32);
    mo_slot_commit(c, slot, nk);
    return MO_OK;
}

extern "C" int mo_dbg_set_poison(mo_ctx* c, int byte) {
    if (!c) return MO_ERR_ARG;
    c->poison = byte < 0 ? -1 : (byte & 255);
    return MO_OK;
}

extern "C" int mo_dbg_min_eigen(mo_ctx* c, const uint8_t* gray, int w, int h, float* eig) {
    if (!c) return MO_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return gftt_run(c, gray, w, h, w, 1, 64, nullptr, nullptr, eig);
}

extern "C" int mo_match_knn2_ratio(mo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, const double* ratio,
                                   int batch, int32_t* train_idx, int32_t* dist, uint8_t* pass) {
    if (!c) return MO_ERR_ARG;
    if (nq < 0 || nt < 0 || batch < 1) return mo_fail(c, MO_ERR_ARG, "bad sizes");
    if (nq == 0) return MO_OK;
    if (!q || !train_idx || !dist || !pass || (nt > 0 && !t)) return mo_fail(c, MO_ERR_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    HostClock clk(c);
    size_t qb = (size_t)batch * nq * 32, tb = (size_t)batch * std::max(nt, 1) * 32, n = (size_t)batch * nq;
    int rc;
    if ((rc = mo_reserve(c, c->d_mq, c->m_q_bytes, qb))) return rc;
    if ((rc = mo_reserve(c, c->d_mt, c->m_t_bytes, tb))) return rc;
    if (n > c->m_n) {
        if (c->d_midx) hipFree(c->d_midx);
        if (c->d_mdist) hipFree(c->d_mdist);
        if (c->d_mpass) hipFree(c->d_mpass);
        c->d_midx = nullptr; c->d_mdist = nullptr; c->d_mpass = nullptr; c->m_n = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_midx, n * 2 * sizeof(int32_t)));
        HIPCHK(c, hipMalloc((void**)&c->d_mdist, n * 2 * sizeof(int32_t)));
        HIPCHK(c, hipMalloc((void**)&c->d_mpass, n));
        c->m_n = n;
    }
    const size_t tbytes = (size_t)batch * nt * 32, o_t = (qb + 15) & ~(size_t)15, o_idx = (o_t + tbytes + 15) & ~(size_t)15;
    const size_t o_dist = o_idx + n * 2 * sizeof(int32_t), o_pass = o_dist + n * 2 * sizeof(int32_t), total = o_pass + n;
    const bool staged = total <= (size_t)2 << 20;  // the single-pair calls of the drop-in classes
    if (staged) {
        if ((rc = host_stage(c, total))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (the staging buffer of a previous call has been consumed)
        std::memcpy(c->h_stage, q, qb);
        if (nt > 0) std::memcpy(c->h_stage + o_t, t, tbytes);
        mo_stage_begin(c);
        HIPCHK(c, hipMemcpyAsync(c->d_mq, c->h_stage, qb, hipMemcpyHostToDevice, c->stream));
        if (nt > 0) HIPCHK(c, hipMemcpyAsync(c->d_mt, c->h_stage + o_t, tbytes, hipMemcpyHostToDevice, c->stream));
    } else {
        mo_stage_begin(c);
        HIPCHK(c, hipMemcpyAsync(c->d_mq, q, qb, hipMemcpyHostToDevice, c->stream));
        if (nt > 0) HIPCHK(c, hipMemcpyAsync(c->d_mt, t, tbytes, hipMemcpyHostToDevice, c->stream));
    }
    mo_stage_mark(c, "h2d");
    rc = match_launch_pairs(c, c->d_mq, c->d_mt, (size_t)nq * 32, (size_t)nt * 32, nullptr, nullptr, nullptr, nq, nt, batch,
                            nq, ratio ? *ratio : -1.0, c->d_midx, c->d_mdist, c->d_mpass);
    if (rc) return rc;
    mo_stage_mark(c, "match_knn2_ratio");
    if (staged) {
        HIPCHK(c, hipMemcpyAsync(c->h_stage + o_idx, c->d_midx, n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_stage + o_dist, c->d_mdist, n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_stage + o_pass, c->d_mpass, n, hipMemcpyDeviceToHost, c->stream));
        mo_stage_mark(c, "d2h");
        clk.enqueued();
        HIPCHK(c, hipStreamSynchronize(c->stream));
        clk.waited();
        std::memcpy(train_idx, c->h_stage + o_idx, n * 2 * sizeof(int32_t));
        std::memcpy(dist, c->h_stage + o_dist, n * 2 * sizeof(int32_t));
        std::memcpy(pass, c->h_stage + o_pass, n);
        return MO_OK;
    }
    HIPCHK(c, hipMemcpyAsync(train_idx, c->d_midx, n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dist, c->d_mdist, n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(pass, c->d_mpass, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MO_OK;
}

extern "C" int mo_init_two_view(mo_ctx* c, const float* p1, const float* p2, int m, const double K[9], double thr_px,
                                double prob, int n_hyp, uint64_t seed, double R[9], double t[3], double E[9],
                                uint8_t* ransac_inlier, uint8_t* inlier, float* X, int* n_good) {
    (void)prob;
    if (!c) return MO_ERR_ARG;
    if (!p1 || !p2 || !K || !R || !t || !inlier || !X || !n_good) return mo_fail(c, MO_ERR_ARG, "NULL argument");
    if (m < 0) return mo_fail(c, MO_ERR_ARG, "m must be >= 0");
    HIPCHK(c, hipSetDevice(c->device));
    *n_good = 0;
    if (m < 8) {
        for (int i = 0; i < 9; i++) R[i] = NAN;
        for (int i = 0; i < 3; i++) t[i] = NAN;
        std::memset(inlier, 0, (size_t)m);
        if (ransac_inlier) std::memset(ransac_inlier, 0, (size_t)m);
        if (E) for (int i = 0; i < 9; i++) E[i] = NAN;
        for (int i = 0; i < 3 * m; i++) X[i] = NAN;
        return MO_OK;
    }
    size_t pb = (size_t)m * 2 * sizeof(float);
    size_t need = 2 * pb + 12 * sizeof(double) + 9 * sizeof(double) + (size_t)m * 3 * sizeof(float) + 2 * (size_t)m + 64;
    int rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, need + 256);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)c->d_tmp;
    double* d_pose = (double*)b; b += 12 * sizeof(double);
    double* d_E = (double*)b; b += 9 * sizeof(double);
    int32_t* d_n = (int32_t*)b; b += 8;
    float* d_p1 = (float*)b; b += pb;
    float* d_p2 = (float*)b; b += pb;
    float* d_X = (float*)b; b += (size_t)m * 3 * sizeof(float);
    uint8_t* d_inl = b; b += m;
    uint8_t* d_ran = b;
    HIPCHK(c, hipMemcpyAsync(d_p1, p1, pb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_p2, p2, pb, hipMemcpyHostToDevice, c->stream));
    TwoViewArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_pairs = 1; a.cap = m; a.n_hyp = n_hyp;
    for (int i = 0; i < 9; i++) a.K[i] = K[i];
    a.thr_px = thr_px; a.seed = seed;
    a.d_p1 = d_p1; a.d_p2 = d_p2; a.m_fixed = m;
    a.d_pose = d_pose; a.d_E = d_E; a.d_points = d_X; a.d_inlier = d_inl; a.d_ransac = d_ran; a.d_n_points = d_n;
    mo_stage_begin(c);
    if ((rc = twoview_launch(c, a))) return rc;
    mo_stage_mark(c, "two_view");
    double pose[12];
    int32_t ng = 0;
    HIPCHK(c, hipMemcpyAsync(pose, d_pose, sizeof(pose), hipMemcpyDeviceToHost, c->stream));
    if (E) HIPCHK(c, hipMemcpyAsync(E, d_E, 9 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&ng, d_n, sizeof(ng), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(X, d_X, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(inlier, d_inl, (size_t)m, hipMemcpyDeviceToHost, c->stream));
    if (ransac_inlier) HIPCHK(c, hipMemcpyAsync(ransac_inlier, d_ran, (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 9; i++) R[i] = pose[i];
    for (int i = 0; i < 3; i++) t[i] = pose[9 + i];
    *n_good = ng;
    return MO_OK;
}

extern "C" int mo_recover_pose(mo_ctx* c, const double E[9], const float* p1, const float* p2, int m, const double K[9],
                               const uint8_t* mask_in, double R[9], double t[3], uint8_t* mask_out, float* X, int* n_good) {
    if (!c) return MO_ERR_ARG;
    if (!E || !K || !R || !t || !n_good || m < 0 || (m > 0 && (!p1 || !p2 || !mask_out))) return mo_fail(c, MO_ERR_ARG, "NULL / negative argument");
    HIPCHK(c, hipSetDevice(c->device));
    *n_good = 0;
    for (int i = 0; i < 9; i++) R[i] = NAN;
    for (int i = 0; i < 3; i++) t[i] = NAN;
    if (m == 0) return MO_OK;
    const size_t pb = (size_t)m * 2 * sizeof(float);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_p1 = take(pb), o_p2 = take(pb), o_E = take(9 * sizeof(double)), o_min = take(m), o_pose = take(12 * sizeof(double)),
                 o_X = take((size_t)m * 3 * sizeof(float)), o_inl = take(m), o_n = take(sizeof(int32_t));
    int rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, off);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)c->d_tmp;
    HIPCHK(c, hipMemcpyAsync(b + o_p1, p1, pb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_p2, p2, pb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_E, E, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (mask_in) HIPCHK(c, hipMemcpyAsync(b + o_min, mask_in, (size_t)m, hipMemcpyHostToDevice, c->stream));
    TwoViewArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_pairs = 1; a.cap = m; a.n_hyp = 0;
    for (int i = 0; i < 9; i++) a.K[i] = K[i];
    a.thr_px = 1.0;
    a.d_p1 = (const float*)(b + o_p1); a.d_p2 = (const float*)(b + o_p2); a.m_fixed = m;
    a.d_E_in = (const double*)(b + o_E); a.d_mask_in = mask_in ? b + o_min : nullptr;
    a.d_pose = (double*)(b + o_pose); a.d_points = (float*)(b + o_X); a.d_inlier = b + o_inl; a.d_n_points = (int32_t*)(b + o_n);
    mo_stage_begin(c);
    if ((rc = twoview_launch(c, a))) return rc;
    mo_stage_mark(c, "recover_pose");
    double pose[12];
    int32_t ng = 0;
    HIPCHK(c, hipMemcpyAsync(pose, b + o_pose, sizeof(pose), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(mask_out, b + o_inl, (size_t)m, hipMemcpyDeviceToHost, c->stream));
    if (X) HIPCHK(c, hipMemcpyAsync(X, b + o_X, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&ng, b + o_n, sizeof(ng), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 9; i++) R[i] = pose[i];
    for (int i = 0; i < 3; i++) t[i] = pose[9 + i];
    *n_good = ng;
    return MO_OK;
}

extern "C" int mo_find_fundamental(mo_ctx* c, const float* p1, const float* p2, int m, double thr_px, double prob, int n_hyp,
                                   uint64_t seed, double F[9], uint8_t* mask, int* n_inliers) {
    (void)prob;
    if (!c) return MO_ERR_ARG;
    if (!F || !n_inliers || m < 0 || (m > 0 && (!p1 || !p2 || !mask))) return mo_fail(c, MO_ERR_ARG, "NULL / negative argument");
    HIPCHK(c, hipSetDevice(c->device));
    *n_inliers = 0;
    for (int i = 0; i < 9; i++) F[i] = NAN;
    if (m < 8) { if (m > 0) std::memset(mask, 0, (size_t)m); return MO_OK; }
    const size_t pb = (size_t)m * 2 * sizeof(float);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_p1 = take(pb), o_p2 = take(pb), o_F = take(9 * sizeof(double)), o_X = take((size_t)m * 3 * sizeof(float)),
                 o_ran = take(m), o_n = take(sizeof(int32_t));
    int rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, off);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)c->d_tmp;
    HIPCHK(c, hipMemcpyAsync(b + o_p1, p1, pb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_p2, p2, pb, hipMemcpyHostToDevice, c->stream));
    TwoViewArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_pairs = 1; a.cap = m; a.n_hyp = n_hyp; a.model = 1;
    a.K[0] = a.K[4] = a.K[8] = 1.0;
    a.thr_px = thr_px; a.seed = seed;
    a.d_p1 = (const float*)(b + o_p1); a.d_p2 = (const float*)(b + o_p2); a.m_fixed = m;
    a.d_E = (double*)(b + o_F); a.d_points = (float*)(b + o_X); a.d_ransac = b + o_ran; a.d_n_points = (int32_t*)(b + o_n);
    mo_stage_begin(c);
    if ((rc = twoview_launch(c, a))) return rc;
    mo_stage_mark(c, "find_fundamental");
    int32_t ng = 0;
    HIPCHK(c, hipMemcpyAsync(F, b + o_F, 9 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(mask, b + o_ran, (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&ng, b + o_n, sizeof(ng), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_inliers = ng;
    return MO_OK;
}

extern "C" int mo_track_pair(mo_ctx* c, const mo_keypoint* kps1, int n1, const uint8_t* desc1, const mo_keypoint* kps2, int n2,
                             const uint8_t* desc2, int w, int h, double ratio, double disp_frac, const double K[9], double thr_px,
                             int n_hyp, uint64_t seed, double R[9], double t[3], double E[9], int32_t* sel_idx, int32_t* sel_dist,
                             int* n_sel, uint8_t* inlier, int* n_inliers) {
    if (!c) return MO_ERR_ARG;
    if (!K || !R || !t || !n_sel || !n_inliers || n1 < 0 || n2 < 0) return mo_fail(c, MO_ERR_ARG, "NULL / negative argument");
    *n_sel = 0; *n_inliers = 0;
    for (int i = 0; i < 9; i++) { R[i] = NAN; if (E) E[i] = NAN; }
    for (int i = 0; i < 3; i++) t[i] = NAN;
    if (n1 == 0 || n2 == 0) return MO_OK;
    if (!kps1 || !kps2 || !desc1 || !desc2 || !sel_idx || !inlier) return mo_fail(c, MO_ERR_ARG, "NULL argument");
    // the host-array form of mo_pair_frontend(MO_MODE_TRACK) (frame_api.hip): both frames are uploaded into resident slots
    mo_frame_ref f1 = {0, kps1, desc1, n1}, f2 = {0, kps2, desc2, n2};
    mo_pair_params pp;
    std::memset(&pp, 0, sizeof(pp));
    pp.mode = MO_MODE_TRACK; pp.w = w; pp.h = h; pp.ratio = ratio; pp.disp_frac = disp_frac; pp.thr_px = thr_px; pp.n_hyp = n_hyp; pp.seed = seed;
    for (int i = 0; i < 9; i++) pp.K[i] = K[i];
    mo_pair_out o;
    std::memset(&o, 0, sizeof(o));
    o.sel_idx = sel_idx; o.sel_dist = sel_dist; o.inlier = inlier;
    const int rc = mo_pair_frontend(c, &f1, &f2, &pp, &o);
    if (rc) return rc;
    *n_sel = o.n_sel; *n_inliers = o.n_good;
    for (int i = 0; i < 9; i++) { R[i] = o.R[i]; if (E) E[i] = o.E[i]; }
    for (int i = 0; i < 3; i++) t[i] = o.t[i];
    return MO_OK;
}

extern "C" int mo_triangulate_points(mo_ctx* c, const double P1[12], const double P2[12], const float* p1, const float* p2,
                                     int n, float* X4) {
    if (!c) return MO_ERR_ARG;
    if (n < 0 || !P1 || !P2 || (n > 0 && (!p1 || !p2 || !X4))) return mo_fail(c, MO_ERR_ARG, "NULL argument");
    if (n == 0) return MO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    size_t pb = (size_t)n * 2 * sizeof(float);
    int rc = mo_reserve(c, c->d_tmp, c->tmp_bytes, 2 * pb + (size_t)n * 4 * sizeof(float) + 64);
    if (rc) return rc;
    float* d_p1 = (float*)c->d_tmp;
    float* d_p2 = d_p1 + (size_t)n * 2;
    float* d_X = d_p2 + (size_t)n * 2;
    HIPCHK(c, hipMemcpyAsync(d_p1, p1, pb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_p2, p2, pb, hipMemcpyHostToDevice, c->stream));
    if ((rc = triangulate_launch(c, P1, P2, d_p1, d_p2, n, d_X))) return rc;
    HIPCHK(c, hipMemcpyAsync(X4, d_X, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MO_OK;
}

// ---- device-resident batched mode -----------------------------------------------------------------
extern "C" int mo_dev_orb_detect_compute(mo_ctx* c, const mo_orb_params* p, const uint8_t* d_gray, int w, int h, int batch,
                                         mo_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_counts) {
    if (!c) return MO_ERR_ARG;
    if (!d_gray || !d_kps || !d_counts) return mo_fail(c, MO_ERR_ARG, "NULL device pointer");
    HIPCHK(c, hipSetDevice(c->device));
    return mo_run_extract(c, p, d_gray, w, h, batch, d_kps, d_desc, cap, d_counts, false);
}

extern "C" int mo_dev_match_pairs(mo_ctx* c, const uint8_t* d_desc, const int32_t* d_counts, int cap, const int32_t* d_qf,
                                  const int32_t* d_tf, int n_pairs, double ratio, int32_t* d_idx, int32_t* d_dist,
                                  uint8_t* d_pass) {
    if (!c) return MO_ERR_ARG;
    if (!d_desc || !d_counts || !d_idx || !d_dist || !d_pass) return mo_fail(c, MO_ERR_ARG, "NULL device pointer");
    HIPCHK(c, hipSetDevice(c->device));
    mo_stage_begin(c);
    int rc = match_launch_pairs(c, d_desc, d_desc, (size_t)cap * 32, (size_t)cap * 32, d_counts, d_qf, d_tf, 0, 0, n_pairs, cap,
                                ratio, d_idx, d_dist, d_pass);
    mo_stage_mark(c, "match_knn2_ratio");
    return rc;
}

__global__ void k_pair_frames(int32_t* qf, int32_t* tf, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { qf[i] = i; tf[i] = i + 1; }
}

extern "C" int mo_dev_frontend_batch(mo_ctx* c, const mo_orb_params* p, const mo_batch_io* io) {
    if (!c) return MO_ERR_ARG;
    if (!io || !io->d_gray || !io->d_kps || !io->d_desc || !io->d_counts) return mo_fail(c, MO_ERR_ARG, "NULL in mo_batch_io");
    HIPCHK(c, hipSetDevice(c->device));
    if (io->detector != MO_DETECT_ORB && io->detector != MO_DETECT_GRID) return mo_fail(c, MO_ERR_ARG, "mo_batch_io.detector must be MO_DETECT_ORB or MO_DETECT_GRID");
    int rc = io->detector == MO_DETECT_GRID ? run_grid_extract(c, p, io)
                                            : mo_run_extract(c, p, io->d_gray, io->w, io->h, io->batch, io->d_kps, io->d_desc, io->cap, io->d_counts, false);
    if (rc) return rc;
    if (io->mode == MO_MODE_KEYFRAME) {
        // LocalMapper._process_new_keyframe (local_mapper.py:116-149) for n_kf_pairs (query keyframe, train keyframe) pairs of the batch
        const int np = io->n_kf_pairs;
        if (np < 1) return MO_OK;
        if (!io->d_kf_query || !io->d_kf_train || !io->d_kf_P1 || !io->d_kf_P2 || !io->d_match_idx || !io->d_match_dist || !io->d_match_pass ||
            !io->d_points || !io->d_n_points)
            return mo_fail(c, MO_ERR_ARG, "MO_MODE_KEYFRAME needs d_kf_query / d_kf_train / d_kf_P1 / d_kf_P2, the match outputs, d_points and d_n_points");
        if (io->n_hyp < 1) return mo_fail(c, MO_ERR_ARG, "MO_MODE_KEYFRAME needs n_hyp >= 1");
        rc = match_launch_pairs(c, io->d_desc, io->d_desc, (size_t)io->cap * 32, (size_t)io->cap * 32, io->d_counts, io->d_kf_query, io->d_kf_train,
                                0, 0, np, io->cap, io->ratio, io->d_match_idx, io->d_match_dist, io->d_match_pass);
        if (rc) return rc;
        mo_stage_mark(c, "match_knn2_ratio");
        TwoViewArgs a;
        std::memset(&a, 0, sizeof(a));
        a.n_pairs = np; a.cap = io->cap; a.n_hyp = io->n_hyp; a.model = 1;
        a.K[0] = a.K[4] = a.K[8] = 1.0;
        a.thr_px = io->thr_px; a.seed = io->seed; a.pair_base = io->pair_index_base;
        a.d_kps = io->d_kps; a.d_counts = io->d_counts; a.d_match_idx = io->d_match_idx; a.d_match_pass = io->d_match_pass;
        a.d_qf = io->d_kf_query; a.d_tf = io->d_kf_train; a.need_two = 1; a.d_P1 = io->d_kf_P1; a.d_P2 = io->d_kf_P2;
        a.d_E = io->d_kf_F; a.d_points = io->d_points; a.d_n_points = io->d_n_points; a.d_inlier = io->d_pose_mask;
        if ((rc = twoview_launch(c, a))) return rc;
        mo_stage_mark(c, "keyframe_f_ransac_triangulate");
        return MO_OK;
    }
    int n_pairs = io->batch - 1;
    if (n_pairs < 1 || !io->d_match_idx) return MO_OK;
    if (!io->d_match_dist || !io->d_match_pass) return mo_fail(c, MO_ERR_ARG, "match outputs missing");
    // (query, train) frame of every pair: written once per batch size into a buffer of its own (it was a 5 us launch per call)
    if (c->pair_frames_n < n_pairs) {
        if (c->d_pair_frames) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->d_pair_frames)); c->d_pair_frames = nullptr; }
        c->pair_frames_n = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_pair_frames, (size_t)n_pairs * 2 * sizeof(int32_t)));
        hipLaunchKernelGGL(k_pair_frames, dim3((n_pairs + 255) / 256), dim3(256), 0, c->stream, c->d_pair_frames, c->d_pair_frames + n_pairs,
                           n_pairs);
        HIPCHK(c, hipGetLastError());
        c->pair_frames_n = n_pairs;
        c->pair_frames_split = n_pairs;
    }
    int32_t* qf = c->d_pair_frames;
    int32_t* tf = qf + c->pair_frames_split;
    rc = match_launch_pairs(c, io->d_desc, io->d_desc, (size_t)io->cap * 32, (size_t)io->cap * 32, io->d_counts, qf, tf, 0, 0,
                            n_pairs, io->cap, io->ratio, io->d_match_idx, io->d_match_dist, io->d_match_pass);
    if (rc) return rc;
    mo_stage_mark(c, "match_knn2_ratio");
    if (io->n_hyp > 0) {
        if (!io->d_points || !io->d_n_points) return mo_fail(c, MO_ERR_ARG, "two-view outputs missing");
        TwoViewArgs a;
        std::memset(&a, 0, sizeof(a));
        a.n_pairs = n_pairs; a.cap = io->cap; a.n_hyp = io->n_hyp;
        for (int i = 0; i < 9; i++) a.K[i] = io->K[i];
        a.thr_px = io->thr_px; a.seed = io->seed; a.pair_base = io->pair_index_base;
        a.d_kps = io->d_kps; a.d_counts = io->d_counts; a.d_match_idx = io->d_match_idx; a.d_match_pass = io->d_match_pass;
        a.d_pose = io->d_pose; a.d_points = io->d_points; a.d_n_points = io->d_n_points; a.d_inlier = io->d_pose_mask;
        if (io->mode == MO_MODE_TRACK) {
            if (!io->d_sel_idx || !io->d_sel_n) return mo_fail(c, MO_ERR_ARG, "MO_MODE_TRACK needs d_sel_idx and d_sel_n");
            if ((rc = track_select_launch(c, io->d_kps, io->d_counts, qf, tf, io->d_match_idx, io->d_match_dist, io->d_match_pass,
                                          io->cap, n_pairs, io->w, io->h, io->disp_frac, io->d_sel_idx, io->d_sel_dist, io->d_sel_n)))
                return rc;
            mo_stage_mark(c, "track_filters");
            a.d_sel = io->d_sel_idx; a.d_sel_n = io->d_sel_n;
        } else if (io->mode != MO_MODE_INIT) return mo_fail(c, MO_ERR_ARG, "mo_batch_io.mode must be MO_MODE_INIT or MO_MODE_TRACK");
        if ((rc = twoview_launch(c, a))) return rc;
        mo_stage_mark(c, "two_view");
    }
    return MO_OK;
}

// ---- probes for stage-level parity tests -------------------------------------------------------------
extern "C" int mo_dbg_pyramid_level(mo_ctx* c, const mo_orb_params* p, const uint8_t* gray, int w, int h, int level,
                                    int blurred, uint8_t* out, int* lw, int* lh) {
    if (!c) return MO_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = mo_build_plan(c, p, w, h, 1);
    if (rc) return rc;
    if (level < 0 || level >= c->plan.nlevels) return mo_fail(c, MO_ERR_ARG, "level out of range");
    const uint8_t* d_gray = nullptr;
    if ((rc = stage_images(c, gray, w, h, w, 1, 1, &d_gray))) return rc;
    const bool fused = (blurred & 2) != 0;  // bit 1: through the single-frame kernel (front_single.hip) instead of k_resize2 / k_blur
    blurred &= 1;
    if (fused) {
        if (!c->fs_ok) return mo_fail(c, MO_ERR_UNSUPPORTED, std::string("the single-frame pyramid kernel does not cover this geometry: ") + c->fs_why);
        HIPCHK(c, hipMemsetAsync(c->d_pyr, 0xA5, (size_t)c->plan.pyr_stride, c->stream));   // whatever it does not write shows
        HIPCHK(c, hipMemsetAsync(c->d_blur, 0xA5, (size_t)c->plan.blur_stride, c->stream));
        if ((rc = orb_launch_front_single(c, d_gray, 1, 1))) return rc;
    } else if ((rc = orb_launch_pyramid(c, d_gray, 1, c->plan.nlevels, 0))) return rc;
    const LevelInfo& v = c->plan.lv[level];
    *lw = v.w; *lh = v.h;
    if (blurred) {
        if (!fused && (rc = orb_launch_blur(c, d_gray, 1, c->plan.nlevels, 0))) return rc;
        HIPCHK(c, hipMemcpy2DAsync(out, v.w, c->d_blur + v.boff, v.bpitch, v.w, v.h, hipMemcpyDeviceToHost, c->stream));
    } else {
        const uint8_t* src = level == 0 ? d_gray : c->d_pyr + v.off;
        HIPCHK(c, hipMemcpy2DAsync(out, v.w, src, v.pitch, v.w, v.h, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MO_OK;
}

extern "C" int mo_dbg_fast_level(mo_ctx* c, const mo_orb_params* p, const uint8_t* gray, int w, int h, int level,
                                 int32_t* xys, int cap, int* n) {
    if (!c) return MO_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = mo_build_plan(c, p, w, h, 1);
    if (rc) return rc;
    if (level < 0 || level >= c->plan.nlevels) return mo_fail(c, MO_ERR_ARG, "level out of range");
    const uint8_t* d_gray = nullptr;
    if ((rc = stage_images(c, gray, w, h, w, 1, 1, &d_gray))) return rc;
    if ((rc = orb_launch_pyramid(c, d_gray, 1, c->plan.nlevels, 0))) return rc;
    if ((rc = orb_launch_fast(c, d_gray, 1))) return rc;
    const LevelInfo& v = c->plan.lv[level];
    std::vector<int> cnt(std::max(v.nstrips, 1));
    std::vector<uint32_t> ent((size_t)std::max(v.cand_cap, 1));
    if (v.nstrips > 0) {
        HIPCHK(c, hipMemcpyAsync(cnt.data(), c->d_strip_cnt + v.strip_base, v.nstrips * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(ent.data(), c->d_cand + v.cand_off, (size_t)v.cand_cap * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int k = 0;
    for (int s = 0; s < v.nstrips; s++)
        for (int i = 0; i < cnt[s]; i++) {
            uint32_t e = ent[(size_t)s * v.strip_cap + i];
            if (k < cap) { xys[3 * k] = e & 0xFFF; xys[3 * k + 1] = (e >> 12) & 0xFFF; xys[3 * k + 2] = e >> 24; }
            k++;
        }
    *n = k;
    return MO_OK;
}

extern "C" int mo_dbg_retain_best(mo_ctx* c, const float* resp, int n, int n_points, int select_order, int32_t* order,
                                  int* n_out) {
    if (!c) return MO_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (n <= 0) { *n_out = 0; return MO_OK; }
    float* d_r = nullptr; int32_t* d_o = nullptr; int* d_n = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_r, (size_t)n * sizeof(float)));
    HIPCHK(c, hipMalloc((void**)&d_o, (size_t)n * sizeof(int32_t)));
    HIPCHK(c, hipMalloc((void**)&d_n, sizeof(int)));
    HIPCHK(c, hipMemcpyAsync(d_r, resp, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    int rc = orb_launch_retain_probe(c, d_r, n, n_points, select_order, d_o, d_n);
    if (!rc) {
        HIPCHK(c, hipMemcpyAsync(n_out, d_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(order, d_o, (size_t)(*n_out) * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    hipFree(d_r); hipFree(d_o); hipFree(d_n);
    return rc;
}

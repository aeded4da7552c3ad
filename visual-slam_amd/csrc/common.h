// common.h -- context, plan and helpers shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vslam_amd.h"

#define MO_MAX_LEVELS 12
#define MO_HALF_PATCH 15
#define MO_STRIP_ROWS 8
// rows per FAST strip in a context for one or two frames at a time (max_batch <= 2): such a call waits for the longest strip's chain, so
// shorter strips on more workgroups cut it (FAST stage of one 640x480 frame: 26.3 / 18.2 / 15.5 / 14.2 us at 8 / 4 / 3 / 2 rows; the
// selection's gather pays 2 us for the extra strips; 1 row: slower again; profiles/r04_ab_strip_rows_single.txt)
#ifndef MO_STRIP_ROWS_LATENCY
#define MO_STRIP_ROWS_LATENCY 2
#endif
#define SEL_MAXSTRIPS 2047  // strips of one level that k_select's prefix table (dynamic LDS, behind the record window) holds

// per-level geometry, uploaded by value as a kernel argument
struct LevelInfo {
    int w, h, pitch;      // level size and row pitch in bytes (level 0: pitch = w, aliases the input)
    int off;              // byte offset of the level inside one frame's raw pyramid slab (level 0: unused)
    int bpitch, boff;     // row pitch / byte offset inside one frame's blurred pyramid slab
    float scale;          // (float)pow((double)scale_factor, L)
    int quota;            // features wanted on this level
    int bx0, by0, bw, bh; // border region [bx0, bx0+bw) x [by0, by0+bh): keypoints allowed here
    uint32_t inv_bw;      // floor(2^32 / bw) + 1 for bw > 1: i / bw == mulhi(i, inv_bw) while i * bw < 2^32
    int strip_rows;       // rows per FAST strip
    int nstrips;          // strips covering the border region
    int strip_cap;        // entries per strip slot
    int strip_base;       // index of this level's first strip among one frame's strips
    int cand_off;         // entry offset of this level's first strip slot in one frame's candidate slab
    int cand_cap;         // total candidate capacity of the level (nstrips * strip_cap)
    int fin_off, fin_cap; // final-keypoint slot of the level in one frame's slab
    int scr_off;          // u64 offset of the level's overflow scratch inside one frame's scratch slab
};

struct Plan {
    int w, h, nlevels;
    int edge_threshold, fast_threshold, select_order, nfeatures;
    int pyr_stride;      // bytes per frame of the raw pyramid slab (levels 1..n-1)
    int blur_stride;     // bytes per frame of the blurred pyramid slab (levels 0..n-1)
    int strips_per_frame;
    int cand_stride;     // candidate entries (u32) per frame
    int fin_stride;      // final entries per frame
    int umax[MO_HALF_PATCH + 1];
    int gk[7];           // 7-tap Gaussian, 8 fractional bits
    LevelInfo lv[MO_MAX_LEVELS];
};

struct FinalKp {  // 8 bytes: survivor of both retainBest passes, level coordinates
    uint16_t x, y;
    float response;
};

struct ResizeTab {  // device arrays of one level's INTER_LINEAR_EXACT coefficients (one allocation, base = xpk)
    // packed per output column / row, padded to a multiple of 64 entries with the last one: source offset (15 bits) |
    // (right / lower neighbour offset - offset) << 15 | weight of that neighbour in 1/256 units << 16 (k_resize2)
    uint32_t* xpk = nullptr; uint32_t* ypk = nullptr;
    int* xofs = nullptr; int* xc1 = nullptr; int* yofs = nullptr; int* yc1 = nullptr;  // the same, unpacked (k_resize)
    bool two_pass_ok = true;  // k_resize2's 8-byte source window holds every group of 4 output columns
};

#define MO_RESULT_SLOTS 4
#define MO_NSTAGES 16
#define MO_TIMING_SLOTS 64

struct TimingSet {
    hipEvent_t ev[MO_NSTAGES + 1] = {};
    const char* names[MO_NSTAGES + 1] = {};
    int n_stages = 0;
};

struct mo_ctx {
    int device = 0;
    int max_w = 0, max_h = 0, max_batch = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int match_mode = 0;        // VSLAM_AMD_MATCHER: 0 default (XOR + popcount, train tiles through LDS), 1 "mfma" opt-in matrix-core matcher
    int poison = -1;           // mo_dbg_set_poison (tests): fill the pyramid buffers with that byte before every extraction
    std::string err;

    // plan (rebuilt when w, h or the ORB parameters change)
    bool plan_valid = false;
    // per-level final-keypoint slots = min(candidates, (4 quota + 256) x fin_slack[L]): a level whose response ties overflow its slot
    // (flag bit 0; the kernel names the level in flag word 1) grows eightfold - that level only - and the plan is rebuilt; the factors
    // start again at 1 whenever the image size or the ORB parameters change
    int fin_slack[MO_MAX_LEVELS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    bool fin_slack_dirty = false;           // a factor changed since the plan was built
    bool tie_overflow = false;              // the last host extraction raised flag bit 0
    int tie_levels = 0;                     // ... on these levels (bit L)
    mo_orb_params plan_params{};
    Plan plan{};
    ResizeTab rtab[MO_MAX_LEVELS];
    int batch_alloc = 0;  // frames the work buffers below are sized for

    // work buffers (device)
    uint8_t* d_in = nullptr;       size_t d_in_bytes = 0;      // staged host images (any ch)
    uint8_t* d_gray = nullptr;     size_t d_gray_bytes = 0;    // gray level 0 when converted / staged
    uint8_t* d_pyr = nullptr;      // [batch][pyr_stride]
    uint8_t* d_blur = nullptr;     // [batch][blur_stride]
    uint32_t* d_cand = nullptr;    // [batch][cand_stride]
    int* d_strip_cnt = nullptr;    // [batch][strips_per_frame]
    uint64_t* d_scratch = nullptr; // [batch][nlevels] overflow scratch for the selection replay
    size_t scratch_stride = 0;     // u64 entries per frame of overflow scratch
    FinalKp* d_fin = nullptr;      // [batch][fin_stride]
    int* d_fin_cnt = nullptr;      // [batch][MO_MAX_LEVELS]
    uint32_t* d_tile_tab[2] = {nullptr, nullptr};          // blur: tile -> level | tile column << 8 | tile row << 20 (built with the plan); [0]: whole levels, [1]: without the margin tile_margin
    int tile_cum[2][MO_MAX_LEVELS + 1] = {};               // tiles of levels < L (the tables are level-major: a prefix blurs the first levels)
    int tile_margin = 0;
    uint32_t* d_fs_tab = nullptr; int fs_tiles = 0, fs_stride = 0, fs_lds = 0; bool fs_ok = false; int fs_geom[10] = {}; const char* fs_why = "";  // k_front_single: per-tile headers + coefficient slices (built with the plan)
    uint32_t* d_strip_tab = nullptr; int n_strip_tab = 0;  // FAST: strip of a frame -> level | strip of the level << 8
    uint32_t* d_dtile_tab = nullptr; int n_dtiles = 0, dtile_icw_off = 0;  // k_describe_tiles: tile -> level | column << 8 | row << 20, then the centroid weights
    int* d_dtodo = nullptr; size_t dtodo_bytes = 0;  // k_describe_tiles -> k_describe_tiles_rare: [0] count, then frame * tiles + tile
    int* d_flags = nullptr;        // [8] error flags raised by kernels: words 0..3 belong to the mo_dev_* calls (they accumulate until
                                   // mo_dev_status), words 4..7 to the host entry points (cleared and checked inside each call)
    int* flags_cur = nullptr;      // the word block the kernels of the current call raise their bits in
    unsigned lds_attr_done = 0;    // bit per kernel whose max-dynamic-LDS attribute has been raised on this device
    // output staging for the host API (batches of frames)
    mo_keypoint* d_kps = nullptr; uint8_t* d_desc = nullptr; int* d_counts = nullptr; int out_cap = 0, out_batch = 0;
    // Resident results of the last MO_RESULT_SLOTS single-frame extractions of the host API: slot s is "frame s" of these arrays, so the
    // pair stages (matcher, tracking filters, two-view) run on two slots exactly as they run on two frames of a batch, and a Tracker-style
    // caller that hands a frame's token back (mo_pair_frontend) uploads nothing.
    mo_keypoint* d_slot_kps = nullptr; uint8_t* d_slot_desc = nullptr; int32_t* d_slot_cnt = nullptr; int32_t* d_slot_ids = nullptr;
    int slot_cap = 0, slot_cur = -1;
    uint64_t slot_token[MO_RESULT_SLOTS] = {}; int slot_n[MO_RESULT_SLOTS] = {}; uint64_t token_next = 1, last_token = 0;
    bool host_timing = false;      // stage events inside the single-call host entry points (mo_set_host_timing; an event between two
                                   // kernels idles the GPU for ~ 4.5 us, five of them were 9 % of a single-frame extraction)
    // matcher staging
    uint8_t* d_mq = nullptr; uint8_t* d_mt = nullptr; int32_t* d_midx = nullptr; int32_t* d_mdist = nullptr;
    uint8_t* d_mpass = nullptr; size_t m_q_bytes = 0, m_t_bytes = 0, m_n = 0;
    uint2* d_match_part = nullptr; size_t match_part_bytes = 0;  // per-slice keys of a split k_match_lds launch
    // two-view work buffers
    void* d_tv = nullptr; size_t tv_bytes = 0;
    float* d_stream_pts = nullptr; size_t stream_pts_bytes = 0;  // mo_stream: map-point scratch of chunks whose caller does not want them
    uint32_t* d_track_keys = nullptr; size_t track_keys_bytes = 0;  // k_track_select: key arrays of frames too large for LDS
    // generic temp
    void* d_tmp = nullptr; size_t tmp_bytes = 0;

    // RCCL communicator of the sharded batched mode (comm.hip); null until mo_comm_init
    void* comm = nullptr; int comm_rank = 0, comm_world = 1;
    int32_t* d_comm_cnt = nullptr;  // this rank's row count for mo_gather_map_points' all-gather

    double host_us[4] = {0, 0, 0, 0};  // mo_host_times: enqueue / wait / unpack / total of the last single-call host entry point
    uint8_t* h_stage = nullptr; size_t h_stage_bytes = 0;   // pinned host staging of small host-API results
    int32_t* d_pair_frames = nullptr; int pair_frames_n = 0, pair_frames_split = 0;  // mo_dev_frontend_batch: qf[i] = i, tf[i] = i + 1
    // stage timing: a ring of event sets, one per mo_* call (mo_stage_begin advances it), so that a caller can enqueue many calls
    // back to back and read the per-stage times of the last MO_TIMING_SLOTS of them after ONE synchronisation (mo_stage_times_back)
    TimingSet tsets[MO_TIMING_SLOTS];
    int tcur = 0;
    bool timing = true;
};

int mo_fail(mo_ctx* c, int code, const std::string& msg);

#define HIPCHK(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return mo_fail((c), MO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

// grow-only device buffer helper
template <class T> int mo_reserve(mo_ctx* c, T*& p, size_t& have_bytes, size_t need_bytes) {
    if (need_bytes <= have_bytes && p) return MO_OK;
    if (p) HIPCHK(c, hipFree(p));
    p = nullptr; have_bytes = 0;
    HIPCHK(c, hipMalloc((void**)&p, need_bytes ? need_bytes : 16));
    have_bytes = need_bytes;
    return MO_OK;
}

// pinned, device-mapped host staging of the single-call host entry points (grow-only)
int mo_host_stage(mo_ctx* c, size_t bytes);
// flag words of the host entry points (see api.hip)
static inline int* mo_host_flags(mo_ctx* c) { return c->d_flags + 4; }

// Host-side clock of the single-call entry points (mo_host_times): [0] entry -> everything enqueued (staging memcpy, copies, launches),
// [1] the wait for the stream, [2] unpacking into the caller's arrays, [3] the whole call; microseconds.
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct HostClock {
    mo_ctx* c; double t0, t1 = 0, t2 = 0;
    explicit HostClock(mo_ctx* c_) : c(c_), t0(now_us()) { c->timing = c->host_timing; }
    void enqueued() { t1 = now_us(); }
    void waited() { t2 = now_us(); }
    ~HostClock() {
        const double t3 = now_us();
        if (t1 == 0) t1 = t2 = t3; else if (t2 == 0) t2 = t3;
        c->host_us[0] = t1 - t0; c->host_us[1] = t2 - t1; c->host_us[2] = t3 - t2; c->host_us[3] = t3 - t0;
        c->timing = true;  // (the mo_dev_* calls always record their stage events: bench.py reads them)
    }
};

// device pipeline on frames already resident as dense gray [batch][h][w] (api.hip); host_call != 0: the caller opened the stage set and
// the kernels raise their bits in the host flag words (1: cleared here by a fill, 2: already cleared by the upload kernel)
int mo_run_extract(mo_ctx* c, const mo_orb_params* p, const uint8_t* d_gray, int w, int h, int batch, mo_keypoint* d_kps, uint8_t* d_desc,
                   int cap, int32_t* d_counts, int host_call);
// frame_api.hip: single-frame extraction into a resident result slot (pinned staging in and out, one synchronisation)
int mo_detect_single(mo_ctx* c, const mo_orb_params* p, const uint8_t* img, int w, int h, int stride, int ch, mo_keypoint* kps,
                     uint8_t* desc, int cap, int* counts);

int mo_slot_acquire(mo_ctx* c, int rows, int* slot);   // frame_api.hip: resident result slots for the other single-frame entry points
uint64_t mo_slot_commit(mo_ctx* c, int slot, int n);
uint8_t* mo_stage_dev(mo_ctx* c);                       // device address of the pinned staging buffer
void mo_copy_out_launch(mo_ctx* c, const void* d_src, void* h_dst_dev, size_t bytes);  // device -> pinned staging, one small kernel

// stage timing helpers (hipEvents on the context stream)
void mo_stage_begin(mo_ctx* c);
void mo_stage_mark(mo_ctx* c, const char* name);

// orb_plan.cpp-equivalent host logic (ctx.hip)
int mo_build_plan(mo_ctx* c, const mo_orb_params* p, int w, int h, int batch);

// kernel launchers (orb_kernels.hip)
int orb_launch_gray(mo_ctx* c, const uint8_t* d_bgr, int w, int h, int batch, uint8_t* d_gray);
int orb_launch_ingest(mo_ctx* c, const uint8_t* src_mapped, int w, int h, int ch, uint8_t* d_gray, int* flags_clear);
int orb_launch_pyramid(mo_ctx* c, const uint8_t* d_gray, int batch, int nlevels, int margin);
// front_single.hip: pyramid + blur of a few frames in ONE launch (single-frame calls); fs_build runs with the plan and leaves
// c->fs_ok false for geometries it does not cover, which keep orb_launch_pyramid + orb_launch_blur
#define MO_FS_MAX_BATCH 2
int fs_build(mo_ctx* c);
int orb_launch_front_single(mo_ctx* c, const uint8_t* d_gray, int batch, int want_blur);
void mo_linear_coeffs(int srcsize, int dstsize, std::vector<int>& ofs, std::vector<int>& c1);
int orb_launch_blur(mo_ctx* c, const uint8_t* d_gray, int batch, int nlevels, int margin);
int orb_launch_fast(mo_ctx* c, const uint8_t* d_gray, int batch, int level_lo = 0, int level_hi = MO_MAX_LEVELS);
int orb_launch_select(mo_ctx* c, const uint8_t* d_gray, int batch, int level_lo = 0, int level_hi = MO_MAX_LEVELS);
int orb_launch_describe(mo_ctx* c, const uint8_t* d_gray, int batch, mo_keypoint* d_kps, uint8_t* d_desc, int cap,
                        int* d_counts);
int orb_launch_describe_given(mo_ctx* c, const uint8_t* d_gray, const mo_keypoint* d_kps, int n, uint8_t* d_desc, const int* d_n = nullptr,
                              int batch = 1, int n_stride = 0);
int gftt_records_launch(mo_ctx* c, const float* d_xy, int* d_cell_n, int per_cell, int w, int h, int edge, mo_keypoint* d_rec,
                        int32_t* d_kept, int rec_stride, int32_t* d_counts_out, int batch, int32_t* d_kbase = nullptr);
// orb_kernels.hip: descriptors of the batched grid detector's records out of one blurred LDS tile per grid cell; returns MO_ERR_UNSUPPORTED
// (without setting the error text) when a cell + halo does not fit the tile, and the caller takes orb_launch_describe_given
int orb_launch_describe_cells(mo_ctx* c, const mo_keypoint* d_kps, const int32_t* d_kbase, uint8_t* d_desc, int cap, int batch);
int orb_launch_retain_probe(mo_ctx* c, const float* d_resp, int n, int n_points, int order, int32_t* d_order,
                            int* d_nout);
// match_kernels.hip
int match_launch_pairs(mo_ctx* c, const uint8_t* d_q, const uint8_t* d_t, size_t q_stride, size_t t_stride,
                       const int32_t* d_counts, const int32_t* d_qf, const int32_t* d_tf, int nq_fixed, int nt_fixed,
                       int n_pairs, int out_stride, double ratio, int32_t* d_idx, int32_t* d_dist, uint8_t* d_pass);
// gftt_kernels.hip
int gftt_launch(mo_ctx* c, const uint8_t* d_gray, int w, int h, int n_features, float* d_eig, float* d_xy, int* d_n, int batch = 1);
// twoview_kernels.hip
struct TwoViewArgs {
    int n_pairs, cap, n_hyp;
    int model;  // 0: essential matrix (K-normalised coordinates, thr_px / focal) + pose + triangulation;
                // 1: fundamental matrix (pixel coordinates, Hartley-normalised with one common scale): F in d_E, mask in d_ransac
    double K[9], thr_px;
    uint64_t seed;
    uint64_t pair_base;  // global index of pair 0 (sharded batches): the sampling stream of a pair depends on its global index only
    // per pair: matches are read from the matcher outputs + keypoints, or from explicit point arrays
    const mo_keypoint* d_kps; const int32_t* d_counts; const int32_t* d_match_idx; const uint8_t* d_match_pass;
    const int32_t* d_sel; const int32_t* d_sel_n;  // tracking mode: [pairs][cap][2] (queryIdx, trainIdx) in the caller's order + counts;
                                                   // when set, these replace the ratio-test flags as the list of correspondences
    const float* d_p1; const float* d_p2; int m_fixed;  // explicit points (host API): [m][2]
    const int32_t* d_qf; const int32_t* d_tf;           // keyframe mode: [pairs] query / train frame of each pair (null: pair p = frames p, p + 1)
    int need_two;                                       // keyframe mode: only queries with a second neighbour take part
    const double* d_P1; const double* d_P2;             // fundamental model + these ([pairs][12], pixel projection matrices): the inliers are
                                                        // triangulated with them into d_points (local_mapper.py:148-149)
    const double* d_E_in; const uint8_t* d_mask_in;     // recoverPose on a GIVEN essential matrix ([pairs][9]) and consensus mask
                                                        // ([pairs][cap] by query index, may be null = all): no RANSAC, no refit
    double* d_pose;   // [pairs][12]
    double* d_E;      // [pairs][9] or null
    float* d_points;  // [pairs][cap][3]
    uint8_t* d_inlier; // [pairs][cap] pose mask
    uint8_t* d_ransac; // [pairs][cap] RANSAC (Sampson) mask or null
    int32_t* d_n_points; // [pairs]
    int* flags;          // capacity flag word (none raised by this stage any more); set by twoview_launch
};
int twoview_launch(mo_ctx* c, const TwoViewArgs& a);
// undistort_kernels.hip
int undistort_launch(mo_ctx* c, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int ch, int batch, const double K[9],
                     const double dist[5]);
// track_kernels.hip
int track_select_launch(mo_ctx* c, const mo_keypoint* d_kps, const int32_t* d_counts, const int32_t* d_qf, const int32_t* d_tf,
                        const int32_t* d_midx, const int32_t* d_mdist, const uint8_t* d_mpass, int cap, int n_pairs, int w, int h,
                        double disp_frac, int32_t* d_sel, int32_t* d_sel_dist, int32_t* d_sel_n);
size_t twoview_workspace_bytes(int n_pairs, int cap, int n_hyp);
int triangulate_launch(mo_ctx* c, const double* P1, const double* P2, const float* d_p1, const float* d_p2, int n, float* d_X4);

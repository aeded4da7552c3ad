/* vslam_amd.h -- C-ABI of the MI355X-native ORB front-end (libvslam_amd.so).
 *
 * Drop-in boundary for ONE hot path of p2004dr/visual-slam: extract -> match -> two-view
 * initialisation.  The reference has no FFI layer; its boundary is the Python class API that
 * Tracker calls.  Each entry point below replaces the cv2 call(s) behind one reference method
 * (file:line relative to the reference tree) and is bound by the ctypes host classes in
 * visual-slam_amd/orbslam2/ which keep the reference's constructors and method signatures.
 *
 * Conventions: plain C, opaque context, caller-allocated outputs, int status (0 = MO_OK, <0 = error,
 * text via mo_last_error), no exceptions cross the boundary, no torch types.  One context is
 * thread-compatible (one caller at a time), like the single-threaded reference.
 * "host" entry points take host pointers and do the H2D/D2H themselves; "mo_dev_*" entry points take
 * DEVICE pointers (inputs already resident in HBM) and enqueue on the context's stream without
 * synchronising, for the batched / multi-GPU mode and the benchmark.  Exceptions, both one-off: the call that first sees a new
 * (image size, ORB parameters, batch) combination builds its plan and work buffers (hipMalloc, blocking table uploads).
 */
#ifndef VSLAM_AMD_H
#define VSLAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The structs below (mo_orb_params, mo_batch_io, mo_frame_ref, mo_pair_params, mo_pair_out, mo_stream_params, mo_stream_result) carry no size
 * field: their layout belongs to the header a caller was compiled against.  mo_batch_io gets fields APPENDED per round, so a caller built
 * against an older header hands in a shorter struct than the library reads: every caller must be recompiled with the header of the library
 * it loads, and can check that at load time: mo_abi_version() == MO_ABI_VERSION. */
#define MO_ABI_VERSION 5
int mo_abi_version(void);

#define MO_OK 0
#define MO_ERR_ARG (-1)       /* bad argument */
#define MO_ERR_HIP (-2)       /* HIP runtime error (see mo_last_error) */
#define MO_ERR_CAPACITY (-3)  /* caller buffer or internal capacity too small */
#define MO_ERR_UNSUPPORTED (-4)

#define MO_ORDER_LIBSTDCXX 0 /* retainBest order of cv2 wheels linked against libstdc++ (Linux) */
#define MO_ORDER_MSVC 1      /* retainBest order of cv2 wheels linked against the MSVC STL (Windows);
                                the order of the reference's gt.yaml fixtures */

typedef struct mo_ctx mo_ctx;

/* same fields as cv2.KeyPoint: pt, size, angle, response, octave, class_id (28 bytes) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} mo_keypoint;

/* cv2.ORB_create arguments as the reference passes them (src/orbslam2/extractor.py:38-48).
 * Only the values the reference uses are implemented: edge_threshold any >= 19 (31 in the reference),
 * first_level 0, wta_k 2, score_type 0 (HARRIS), patch_size 31. */
typedef struct {
    int32_t nfeatures;      /* extractor.py:39 */
    float scale_factor;     /* extractor.py:40 */
    int32_t nlevels;        /* extractor.py:41, 1..12 */
    int32_t edge_threshold; /* extractor.py:42 */
    int32_t first_level;    /* extractor.py:43 */
    int32_t wta_k;          /* extractor.py:44 */
    int32_t score_type;     /* extractor.py:45, 0 = HARRIS_SCORE */
    int32_t patch_size;     /* extractor.py:46 */
    int32_t fast_threshold; /* extractor.py:47 (= min_threshold) */
    int32_t select_order;   /* MO_ORDER_*: which STL's nth_element permutation to reproduce */
} mo_orb_params;

/* ---- context ------------------------------------------------------------------------------ */
mo_ctx* mo_create(int device, int max_w, int max_h, int max_batch);
void mo_destroy(mo_ctx*);
const char* mo_last_error(mo_ctx*); /* valid until the next call on ctx; ctx may be NULL (creation errors) */
int mo_set_stream(mo_ctx*, void* hip_stream); /* NULL = the context's own (non-blocking) stream */
int mo_set_stream_null(mo_ctx*);              /* the HIP null stream (handle 0, e.g. torch's default stream): mo_set_stream cannot name it */
int mo_sync(mo_ctx*);                         /* hipStreamSynchronize on the context stream */
int mo_device_count(void);                    /* hipGetDeviceCount, 0 when there is no GPU */

/* ---- ORBExtractor ----------------------------------------------------------------------- */
/* Replaces cv2.cvtColor + orb.detectAndCompute(image, None)   (extractor.py:61-65, detect_and_compute).
 * img: batch images, u8, ch = 1 (gray) or 3 (BGR), row stride in bytes, images packed at stride*h.
 * kps [batch*cap], desc [batch*cap*32] (may be NULL = detect only), counts [batch].
 * Returns MO_ERR_CAPACITY if a frame yields more than cap keypoints (counts then holds the needs). */
int mo_orb_detect_compute(mo_ctx*, const mo_orb_params*, const uint8_t* img, int w, int h, int stride, int ch,
                          int batch, mo_keypoint* kps, uint8_t* desc, int cap, int* counts);

/* Replaces orb.compute(image, keypoints)   (extractor.py:79-83 compute, :140 distribute_keypoints).
 * Keeps cv2's behaviour: keypoints within edge_threshold of the image border are dropped, kp.angle is
 * used as supplied (no re-orientation), kp.octave selects the level.  kept_idx[i] = index into kps_in
 * of descriptor row i; n_out = number of rows. */
int mo_orb_compute(mo_ctx*, const mo_orb_params*, const uint8_t* img, int w, int h, int stride, int ch,
                   const mo_keypoint* kps_in, int n_in, int32_t* kept_idx, uint8_t* desc, int* n_out);

/* Replaces the 64 cv2.goodFeaturesToTrack(image, maxCorners = n_features // 64, qualityLevel 0.01, minDistance 10,
 * mask = grid cell) calls of ORBExtractor.distribute_keypoints (extractor.py:104-136): 8x8 grid, Shi-Tomasi minimum
 * eigenvalue (blockSize 3, Sobel 3), corners in cell-major order, best first inside a cell.
 * xy [64 * (n_features / 64)][2] pixel coordinates; n_out = number of corners.  Descriptors for them come from
 * mo_orb_compute with angle -1 (extractor.py:135,140). */
int mo_orb_grid_good_features(mo_ctx*, const uint8_t* img, int w, int h, int stride, int ch, int n_features, float* xy,
                              int* n_out);
/* ORBExtractor.distribute_keypoints in ONE call (reference extractor.py:85-144, the path Tracker takes by default, tracker.py:87): the
 * grid corners as above, KeyPoint(x, y, 31) for each of them, orb.compute on that list.  xy [64 * (n_features / 64)][2] and n_xy as in
 * mo_orb_grid_good_features; kept_idx [n_kept] = indices into xy of the corners cv2's compute keeps (rounded position at least
 * edge_threshold inside the image), desc [n_kept][32] their descriptors (angle -1, octave 0).  One image upload and one
 * synchronisation instead of two each for the pair mo_orb_grid_good_features + mo_orb_compute; identical results. */
int mo_orb_grid_detect_compute(mo_ctx*, const mo_orb_params*, const uint8_t* img, int w, int h, int stride, int ch, int n_features,
                               float* xy, int* n_xy, int32_t* kept_idx, uint8_t* desc, int* n_kept);

/* Replaces cv2.undistort(image, camera_matrix, distortion)   (utils.py:40-52, applied by run_video.py:145-149 when a distortion
 * coefficient is non-zero): initUndistortRectifyMap (new camera matrix = K, 1/32-pixel fixed-point map) + remap(INTER_LINEAR,
 * constant 0 border).  img / out: h x w x ch u8 (ch 1 or 3), rows at `stride` bytes (out is dense: w * ch); dist = k1, k2, p1, p2, k3.
 * mo_dev_undistort: batch of dense frames already in HBM, enqueued on the context stream (run it ahead of
 * mo_dev_orb_detect_compute / mo_dev_frontend_batch on the same stream). */
int mo_undistort(mo_ctx*, const uint8_t* img, int w, int h, int stride, int ch, const double K[9], const double dist[5], uint8_t* out);
int mo_dev_undistort(mo_ctx*, const uint8_t* d_src, int w, int h, int ch, int batch, const double K[9], const double dist[5],
                     uint8_t* d_dst);

/* ---- DescriptorMatcher ------------------------------------------------------------------- */
/* Replaces BFMatcher(NORM_HAMMING).knnMatch(d1, d2, k=2) + the Lowe ratio loop (matcher.py:70,73-81).
 * q [batch][nq][32], t [batch][nt][32]; ratio NULL = no ratio test (ratio_test=False).
 * train_idx/dist [batch][nq][2] (missing neighbour: idx -1, dist INT32_MAX), pass [batch][nq]. */
int mo_match_knn2_ratio(mo_ctx*, const uint8_t* q, int nq, const uint8_t* t, int nt, const double* ratio,
                        int batch, int32_t* train_idx, int32_t* dist, uint8_t* pass);

/* ---- MapInitializer ---------------------------------------------------------------------- */
/* Replaces calculate_essential_matrix + recover_pose + triangulate_points + the cheirality loop
 * (initializer.py:79-120; utils.py:56-70,120-160).  8-point essential-matrix RANSAC over n_hyp
 * hypotheses scored in parallel, least-squares refit on the consensus set, recoverPose-style
 * cheirality vote (depth in (0, 50)), DLT triangulation.
 * p1,p2 [m][2] pixel coordinates; K row-major 3x3; thr_px Sampson threshold in pixels; prob is
 * accepted for signature compatibility (all n_hyp hypotheses are always scored).
 * Outputs: R [9] row-major, t [3] unit norm, E [9] (may be NULL), inlier [m] = pose mask (RANSAC inlier
 * AND cheirality of the winning pose), X [m][3] triangulated points (valid where inlier), n_good. */
int mo_init_two_view(mo_ctx*, const float* p1, const float* p2, int m, const double K[9], double thr_px,
                     double prob, int n_hyp, uint64_t seed, double R[9], double t[3], double E[9],
                     uint8_t* ransac_inlier /* [m] findEssentialMat mask, may be NULL */, uint8_t* inlier, float* X,
                     int* n_good);

/* Replaces cv2.recoverPose(E, p1, p2, K, mask)   (utils.py:129-134) for ANY essential matrix the caller holds: decomposition
 * (U W V^T, U W^T V^T, +-u3), cheirality vote over the four candidates on the points selected by mask_in (NULL = all; depth in
 * (0, 50) in both cameras, as cv2), winner's R, t, its mask, and the DLT points of the surviving correspondences.
 * p1, p2 [m][2] pixels, K row-major; mask_out [m]; X [m][3] (may be NULL; NaN where mask_out is 0); n_good. */
int mo_recover_pose(mo_ctx*, const double E[9], const float* p1, const float* p2, int m, const double K[9],
                    const uint8_t* mask_in, double R[9], double t[3], uint8_t* mask_out, float* X, int* n_good);

/* Replaces cv2.findFundamentalMat(points1, points2, cv2.FM_RANSAC, thr_px, prob)   (matcher.py:191 filter_matches_by_fundamental,
 * local_mapper.py:136 keyframe map growth).  Same machinery as mo_init_two_view with the fundamental-matrix model: n_hyp
 * 8-point hypotheses on Hartley-normalised pixel coordinates, rank-2 projection, MSAC ranking on the Sampson distance in pixels,
 * least-squares refit on the consensus set.  (cv2's FM_RANSAC draws 7-point samples; like the 8-point essential RANSAC this
 * is the parallel-hypothesis formulation north_star prescribes - parity is on the inlier set and the epipolar geometry.)
 * p1, p2 [m][2] pixels; F [9] row-major scaled to F[8] = 1 (NaN when m < 8 or no model is found); mask [m]; n_inliers. */
int mo_find_fundamental(mo_ctx*, const float* p1, const float* p2, int m, double thr_px, double prob, int n_hyp, uint64_t seed,
                        double F[9], uint8_t* mask, int* n_inliers);

/* One tracking step of Tracker._track_from_last_frame (tracker.py:214-254) for a single frame pair, host in / host out:
 * matcher.match(prev, cur) -> displacement filter (matcher.py:109-142, frac = 0.02 of (w + h) / 2) -> 2 x median distance
 * filter (matcher.py:144-169) -> cv2.findEssentialMat(RANSAC, 0.999, thr_px = 1.0) -> cv2.recoverPose, all on the device.
 * kps1/desc1 = previous frame (query), kps2/desc2 = current frame (train); a negative ratio disables the ratio test.
 * sel_idx [min(n1, 4096)][2] (queryIdx, trainIdx) and sel_dist receive the kept matches in the reference's order (ascending
 * distance, ties in query order), n_sel their number; inlier [n_sel] the recoverPose mask per kept match.  With fewer than 8
 * kept matches (tracker.py:234) R, t, E are NaN and n_inliers = 0. */
int mo_track_pair(mo_ctx*, const mo_keypoint* kps1, int n1, const uint8_t* desc1, const mo_keypoint* kps2, int n2,
                  const uint8_t* desc2, int w, int h, double ratio, double disp_frac, const double K[9], double thr_px,
                  int n_hyp, uint64_t seed, double R[9], double t[3], double E[9], int32_t* sel_idx, int32_t* sel_dist,
                  int* n_sel, uint8_t* inlier, int* n_inliers);

/* ---- one frame at a time: resident results and the fused pair step -------------------------------------------------------------
 * The reference's Tracker calls the classes once per frame (tracker.py:87 extract_features, then :168-170 initialize or :214-254
 * _track_from_last_frame against the PREVIOUS frame's keypoints and descriptors; tests/tester_map.py:57-75 is that loop).  A
 * single-frame mo_orb_detect_compute (batch == 1, desc != NULL) therefore leaves its keypoints and descriptors RESIDENT on the device
 * (the context keeps the last 4 such results) and mo_last_token names them; mo_pair_frontend runs matcher -> (tracking filters) ->
 * two-view stage on two frames given by token - nothing is uploaded - or by host arrays (uploaded into a slot; out->token1 / token2
 * then name them for the next call).  One call, one synchronisation:
 *   MO_MODE_INIT   MapInitializer.initialize's device work (initializer.py:67-120): matcher.match(d1, d2) with `ratio`, essential matrix at
 *                  thr_px (the reference passes 3.0) on the ratio-test survivors in query order, recoverPose, triangulation
 *   MO_MODE_TRACK  Tracker._track_from_last_frame (tracker.py:214-254): match, displacement filter (disp_frac of (w + h) / 2), 2 x median
 *                  distance filter, essential matrix at thr_px (1.0) + recoverPose on the kept matches in the reference's order
 * n_hyp = 0 stops after the matcher (and the filters).  A token that is no longer resident falls back to the arrays; MO_ERR_ARG when
 * there are none. */
typedef struct {
    uint64_t token;          /* 0 = none */
    const mo_keypoint* kps;  /* [n] host, may be NULL while the token is alive */
    const uint8_t* desc;     /* [n][32] host */
    int32_t n;
} mo_frame_ref;

typedef struct {
    int32_t mode;            /* MO_MODE_INIT or MO_MODE_TRACK */
    int32_t w, h;            /* image size (MO_MODE_TRACK: the displacement gate) */
    double ratio;            /* Lowe ratio; negative = no ratio test */
    double disp_frac;        /* MO_MODE_TRACK: threshold_percent (tracker.py:219: 0.02) */
    double K[9];
    double thr_px;
    int32_t n_hyp;           /* 0 = matcher (+ filters) only */
    uint64_t seed;
    uint64_t pair_index;     /* global index of this pair in a sequence: the sampling stream is a function of (seed, pair_index), so a
                                per-frame loop that counts its pairs gets the poses of the batched mode bit for bit; 0 = stand-alone pair */
} mo_pair_params;

typedef struct {
    /* caller-allocated arrays; n1 = keypoints of frame 1; any of them may be NULL */
    int32_t* match_idx;      /* [n1][2] knn train indices (missing neighbour: -1) */
    int32_t* match_dist;     /* [n1][2] */
    uint8_t* match_pass;     /* [n1] ratio test */
    int32_t* sel_idx;        /* MO_MODE_TRACK: [n1][2] (queryIdx, trainIdx) of the kept matches, ascending distance */
    int32_t* sel_dist;       /* MO_MODE_TRACK: [n1] */
    uint8_t* inlier;         /* MO_MODE_TRACK: [n_sel] recoverPose mask per kept match; MO_MODE_INIT: [n1] pose mask per QUERY keypoint */
    uint8_t* ransac;         /* MO_MODE_INIT: [n1] findEssentialMat mask per query keypoint */
    float* X;                /* MO_MODE_INIT: [n1][3] map point per query keypoint (NaN = none) */
    /* filled by the call */
    double R[9], t[3], E[9]; /* NaN without a model (MO_MODE_TRACK: fewer than 8 kept matches, tracker.py:234) */
    int32_t n_sel;           /* MO_MODE_TRACK: kept matches */
    int32_t n_good;          /* recoverPose inliers */
    int32_t n1, n2;          /* keypoints of the two frames */
    uint64_t token1, token2; /* tokens under which the two frames are resident now */
} mo_pair_out;

int mo_last_token(mo_ctx*, uint64_t* token); /* token of the last single-frame mo_orb_detect_compute with descriptors (0: none) */
int mo_pair_frontend(mo_ctx*, const mo_frame_ref* f1, const mo_frame_ref* f2, const mo_pair_params*, mo_pair_out*);

/* Replaces cv2.triangulatePoints(P1, P2, pts1, pts2)   (utils.py:56-60): per-point 4x4 DLT null vector.
 * P1, P2 row-major 3x4 f64; p1, p2 [n][2] f32; X4 [n][4] f32 homogeneous (unit norm, sign arbitrary -
 * the reference divides by w, utils.py:62-70). */
int mo_triangulate_points(mo_ctx*, const double P1[12], const double P2[12], const float* p1, const float* p2, int n,
                          float* X4);

/* ---- device-resident batched mode (frames independent; shards across GPUs by frame) -------- */
typedef struct {
    /* inputs */
    const uint8_t* d_gray;  /* [batch][h][w] u8, device */
    int32_t w, h, batch;
    int32_t cap;            /* keypoint capacity per frame */
    double ratio;           /* Lowe ratio; a NEGATIVE value disables the test (0.0 is a threshold: nothing with two neighbours passes) */
    double K[9];            /* intrinsics for the two-view stage */
    double thr_px;          /* RANSAC threshold (initializer.py:79 passes 3.0) */
    int32_t n_hyp;          /* hypotheses per pair, 0 = skip the two-view stage */
    uint64_t seed;
    /* outputs, all device pointers */
    mo_keypoint* d_kps;     /* [batch][cap] */
    uint8_t* d_desc;        /* [batch][cap][32] */
    int32_t* d_counts;      /* [batch] */
    int32_t* d_match_idx;   /* [batch-1][cap][2]  pair i = frame i (query) vs frame i+1 (train) */
    int32_t* d_match_dist;  /* [batch-1][cap][2] */
    uint8_t* d_match_pass;  /* [batch-1][cap] */
    double* d_pose;         /* [batch-1][12] R (9) then t (3); may be NULL when n_hyp == 0 */
    float* d_points;        /* [batch-1][cap][3] triangulated points per query keypoint (NaN = none) */
    int32_t* d_n_points;    /* [batch-1] number of valid map points per pair */
    /* ---- appended in round 2 (zero-initialise the struct: all of these are optional) ---- */
    int32_t mode;           /* MO_MODE_INIT (0): two-view stage on the ratio-test survivors in query order (MapInitializer.initialize);
                               MO_MODE_TRACK (1): Tracker._track_from_last_frame (tracker.py:214-254) - displacement filter
                               (matcher.py:109-142) and 2 x median distance filter (matcher.py:144-169) behind the matcher, then the
                               two-view stage on the kept matches in the reference's order (tracker.py:242 passes thr_px = 1.0) */
    double disp_frac;       /* MO_MODE_TRACK: threshold_percent of filter_matches_by_geometric_distance (tracker.py:219: 0.02) */
    int32_t* d_sel_idx;     /* MO_MODE_TRACK: [batch-1][cap][2] (queryIdx, trainIdx) of the kept matches, ascending distance */
    int32_t* d_sel_dist;    /* MO_MODE_TRACK: [batch-1][cap] their Hamming distances; may be NULL */
    int32_t* d_sel_n;       /* MO_MODE_TRACK: [batch-1] number of kept matches */
    uint8_t* d_pose_mask;   /* either mode, may be NULL: [batch-1][cap] recoverPose mask per QUERY keypoint */
    uint64_t pair_index_base; /* global index of this batch's pair 0 when a longer sequence is sharded over calls / ranks: the
                               RANSAC sampling stream of a pair is a function of (seed, global pair index), so the result
                               of a pair does not depend on how the sequence was cut */
    /* ---- appended in round 3 (optional, zero = rounds 1 - 2 behaviour) ---- */
    int32_t detector;       /* MO_DETECT_ORB (0): ORBExtractor.detect_and_compute per frame (FAST / pyramid; extractor.py:50-67);
                               MO_DETECT_GRID (1): ORBExtractor.distribute_keypoints per frame (extractor.py:85-144, what Tracker.process_frame
                               calls through extract_features(frame), tracker.py:87): 8x8 grid of Shi-Tomasi corners, params->nfeatures / 64
                               per cell, KeyPoint(x, y, 31) records, orb.compute at angle -1 on octave 0.  d_kps / d_desc / d_counts then
                               hold the keypoints orb.compute KEEPS (record i = descriptor row i); the reference's own misaligned list
                               (all corners) is available below */
    float* d_grid_xy;       /* MO_DETECT_GRID, may be NULL: [batch][64 * (nfeatures / 64)][2] every grid corner, slot (cell, rank) */
    int32_t* d_grid_n;      /* MO_DETECT_GRID, may be NULL: [batch][66] corners per cell (64), corners in all cells, keypoints kept */
    int32_t* d_grid_kept;   /* MO_DETECT_GRID, may be NULL: [batch][cap] index of keypoint i in the frame's cell-major corner list */
    /* MO_MODE_KEYFRAME (2): LocalMapper._process_new_keyframe (local_mapper.py:116-149) for n_kf_pairs keyframe pairs at once.  The batch's
     * frames are extracted as usual; pair p matches frame d_kf_query[p] (query, the previous keyframe) against frame d_kf_train[p]
     * (train, the new keyframe) with `ratio` (the reference passes 0.8) keeping only queries with two neighbours, runs the
     * fundamental-matrix RANSAC at thr_px (3.0) on the kept matches and triangulates the inliers with the two keyframes' projection
     * matrices K [R|t] (utils.compute_projection_matrix).  Outputs per PAIR (rows = n_kf_pairs instead of batch - 1): d_match_* ,
     * d_points [pair][cap][3] by query keypoint (NaN = no map point), d_n_points [pair] = F-RANSAC inliers, d_pose_mask [pair][cap],
     * d_kf_F [pair][9] (may be NULL; NaN with fewer than 8 matches or no model: the reference returns early in both cases). */
    int32_t n_kf_pairs;
    const int32_t* d_kf_query;  /* [n_kf_pairs] frame index of the query keyframe, 0 <= index < batch (not checked: device data) */
    const int32_t* d_kf_train;  /* [n_kf_pairs] frame index of the train keyframe */
    const double* d_kf_P1;      /* [n_kf_pairs][12] row-major 3x4 projection matrix of the query keyframe */
    const double* d_kf_P2;      /* [n_kf_pairs][12] ... of the train keyframe */
    double* d_kf_F;             /* [n_kf_pairs][9] */
} mo_batch_io;

#define MO_MODE_INIT 0
#define MO_MODE_TRACK 1
#define MO_MODE_KEYFRAME 2
#define MO_DETECT_ORB 0
#define MO_DETECT_GRID 1

/* One pass of the hot path over a batch: extract every frame, match consecutive frames, two-view pose +
 * map points per pair.  Enqueues on the context stream; call mo_sync (or sync the stream) before reading. */
int mo_dev_frontend_batch(mo_ctx*, const mo_orb_params*, const mo_batch_io*);
int mo_dev_orb_detect_compute(mo_ctx*, const mo_orb_params*, const uint8_t* d_gray, int w, int h, int batch,
                              mo_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_counts);
/* pairs: q/t frame indices into d_desc [*][cap][32] with per-frame counts d_counts */
/* ratio < 0 disables the ratio test */
int mo_dev_match_pairs(mo_ctx*, const uint8_t* d_desc, const int32_t* d_counts, int cap, const int32_t* d_qf,
                       const int32_t* d_tf, int n_pairs, double ratio, int32_t* d_idx, int32_t* d_dist,
                       uint8_t* d_pass);

/* ---- a frame SEQUENCE through the batched mode (stream.hip) ---------------------------------------------------------------------
 * The reference's driver hands frames over one at a time (src/tests/tester_map.py:57-75 -> Tracker.process_frame, tracker.py:73-146).
 * A caller that can look a few frames ahead gets the batched mode's rate from host frames: mo_stream cuts the sequence into chunks, runs
 * every chunk as ONE mo_dev_frontend_batch call on chunk + 1 frames (the previous chunk's last frame is staged again in front, so every
 * consecutive pair of the sequence is matched exactly once) and overlaps the upload of chunk i + 1 and the caller's handling of chunk
 * i - 1 with the compute of chunk i (three lanes of pinned / device buffers, an upload and a download stream beside the context's stream).  The sampling
 * stream of a pair is keyed by its global index: poses equal those of a per-frame loop that passes mo_pair_params.pair_index.
 *   mo_stream_submit   n <= chunk host frames (u8, rows of `stride` bytes, frames `frame_stride` bytes apart; 0 = dense): staged by a few
 *                      host threads, uploaded and enqueued; returns without waiting.  At most three chunks may be in flight.
 *   mo_stream_collect  waits for the OLDEST chunk in flight and describes its results: pointers into the lane's pinned host buffer,
 *                      valid until the third submit after the chunk's own.  MO_ERR_CAPACITY when a capacity flag was raised inside the
 *                      chunk (r->flags, bits as in mo_dev_status; the results are still described). */
typedef struct {
    int32_t w, h, ch;          /* frames: u8, ch = 1 (gray) or 3 (BGR, converted on the device) */
    int32_t chunk;             /* frames per batched call; the context needs max_batch >= chunk + 1 */
    int32_t cap;               /* keypoint rows per frame */
    int32_t detector;          /* MO_DETECT_ORB or MO_DETECT_GRID */
    int32_t mode;              /* MO_MODE_TRACK (tracker.py:214-254 on every consecutive pair) or MO_MODE_INIT (initializer.py:67-120) */
    double ratio, disp_frac, K[9], thr_px;
    int32_t n_hyp;
    uint64_t seed;
    uint64_t pair_index_base;  /* global index of the sequence's first pair */
    int32_t want_matches;      /* MO_MODE_TRACK: also return the knn lists (MO_MODE_INIT always does) */
    int32_t want_points;       /* return the map points */
} mo_stream_params;

typedef struct {
    int32_t n_frames, n_pairs;     /* frames of this chunk; pairs = n_frames - 1 for the first chunk, n_frames afterwards */
    uint64_t first_frame;          /* global index of frame row 0 */
    uint64_t first_pair;           /* global index of pair row 0: pair g = (frame g, frame g + 1) */
    int32_t cap, flags;
    int32_t prev_count;            /* keypoints of the frame in front of this chunk (the query frame of pair row 0 when first_pair < first_frame) */
    const int32_t* counts;         /* [n_frames] */
    const mo_keypoint* kps;        /* [n_frames][cap] */
    const uint8_t* desc;           /* [n_frames][cap][32] */
    const int32_t* sel_idx;        /* MO_MODE_TRACK: [n_pairs][cap][2] (queryIdx, trainIdx), the reference's order */
    const int32_t* sel_dist;       /* [n_pairs][cap] */
    const int32_t* sel_n;          /* [n_pairs] */
    const double* pose;            /* [n_pairs][12] R then t (NaN: no model) */
    const uint8_t* pose_mask;      /* [n_pairs][cap] recoverPose mask per QUERY keypoint */
    const int32_t* n_points;       /* [n_pairs] */
    const int32_t* match_idx;      /* [n_pairs][cap][2] or NULL */
    const int32_t* match_dist;     /* [n_pairs][cap][2] or NULL */
    const uint8_t* match_pass;     /* [n_pairs][cap] or NULL */
    const float* points;           /* [n_pairs][cap][3] or NULL */
} mo_stream_result;

typedef struct mo_stream mo_stream;
mo_stream* mo_stream_create(mo_ctx*, const mo_orb_params*, const mo_stream_params*);  /* NULL on failure: mo_last_error(ctx) */
void mo_stream_destroy(mo_stream*);
int mo_stream_submit(mo_stream*, const uint8_t* frames, int n, int stride, size_t frame_stride);
int mo_stream_collect(mo_stream*, mo_stream_result*);
int mo_stream_lanes(void);  /* chunks a stream holds in flight: submit refuses one more before a collect; a collected chunk's result
                               buffer is reused by the mo_stream_lanes()-th submit after it */
const char* mo_stream_last_error(mo_stream*);

/* ---- multi-GPU: the final map-point gather (SURVEY.md 8b mo_gather_map_points, 8e) --------------------------------------
 * One process per GPU, each with its own context; frames are sharded contiguously and nothing is exchanged on the data path.
 * The only collective is this padded gather of the per-pair map points to `root` over RCCL (bound with dlopen at first use).
 *   mo_comm_unique_id : rank 0 creates the 128-byte id; the host program hands it to the other ranks (any channel)
 *   mo_comm_init      : ncclCommInitRank on the context's device; mo_comm_destroy (also done by mo_destroy)
 *   mo_gather_map_points: d_local [rows_max][cap][3] f32 of this rank (rows_local valid rows), d_all on root
 *                       [world][rows_max][cap][3], d_rows_all [world] int32 on every rank; device pointers, enqueued on the
 *                       context stream, no host synchronisation. */
int mo_comm_unique_id(uint8_t id[128]);
int mo_comm_init(mo_ctx*, const uint8_t id[128], int rank, int world);
int mo_comm_destroy(mo_ctx*);
int mo_gather_map_points(mo_ctx*, const float* d_local, int rows_local, int rows_max, int cap, int root, float* d_all,
                         int32_t* d_rows_all);

/* Status of the mo_dev_* calls enqueued since the last mo_dev_status: the kernels never fault on overflow, they clamp and
 * raise a bit.  Host entry points keep their own flag words (checked inside each call): interleaving them with mo_dev_* calls
 * neither clears nor pollutes this status.  Synchronises the context stream, copies the flag word to flags[0] (flags may be NULL; [1..3] reserved, 0)
 * and clears it.  bit 0 (1): a level's internal keypoint slot overflowed (response ties at the quota cut: retainBest keeps every
 * element that ties with the boundary; the slots grow eightfold when this status is read, so repeating the call succeeds after
 * at most a few rounds - host entry points repeat by themselves);
 * bit 1 (2): a frame produced more keypoints than `cap` - its rows are truncated to cap while d_counts[frame] holds the
 * number it needed (so d_counts can EXCEED cap: clamp before indexing, or retry with cap >= max(d_counts));
 * bit 3 (8): a pair had more than 4096 correspondences (rows longer than 4096 are fine, a pair's ratio-test survivors or kept
 * tracking matches beyond that are not): that pair gets no model (NaN pose, 0 points), the others are unaffected;
 * bit 2 (4): not raised any more (rounds 2 - 3: more than 2048 local maxima in one grid cell; such cells are now processed in rounds).
 * Returns MO_OK when no bit is set, MO_ERR_CAPACITY otherwise. */
int mo_dev_status(mo_ctx*, int32_t flags[4]);

/* per-stage device time of the last mo_dev_* call, measured with hipEvents on the context stream.
 * names: NULL-terminated array of stage names owned by the library; ms [n] filled. Returns n stages. */
int mo_stage_times(mo_ctx*, const char*** names, float* ms, int cap);
/* The same for the call `back` calls ago (0 = the last one, at most MO_TIMING_SLOTS - 1 = 63): the library keeps a ring of event
 * sets, one per call, so a caller can enqueue many calls back to back and read their stage times after a single synchronisation
 * instead of waiting for every call's last event (bench.py's timed region). */
int mo_stage_times_back(mo_ctx*, int back, const char*** names, float* ms, int cap);

/* Host-side clock of the last single-call host entry point (mo_orb_detect_compute, mo_match_knn2_ratio, mo_track_pair, mo_init_two_view),
 * microseconds: us[0] entry -> everything enqueued (staging copies, launches), us[1] the wait for the stream, us[2] unpacking into
 * the caller's arrays, us[3] the whole call.  With mo_stage_times (device spans incl. the "h2d" / "d2h" copies) this is the batch-1
 * latency breakdown bench.py reports (single_frame_ms.breakdown). */
int mo_host_times(mo_ctx*, double us[4]);
/* Stage events inside the single-call host entry points (off by default: an event between two kernels idles the GPU for ~ 4.5 us,
 * five of them were 9 % of a single-frame extraction; the mo_dev_* calls always record theirs).  on != 0: mo_stage_times reports the
 * spans of the next host calls ("h2d", the kernel stages, "d2h"). */
int mo_set_host_timing(mo_ctx*, int on);

/* internal-stage probes used by the parity tests (device pipeline, host in/out) */
/* blurred: bit 0 = the blurred level instead of the raw one; bit 1 = through the single-frame kernel (pyramid + blur in one launch,
   the route of calls on one or two frames) instead of the batched path's kernels: MO_ERR_UNSUPPORTED for a geometry it does not cover */
int mo_dbg_pyramid_level(mo_ctx*, const mo_orb_params*, const uint8_t* gray, int w, int h, int level, int blurred,
                         uint8_t* out /* lw*lh */, int* lw, int* lh);
int mo_dbg_fast_level(mo_ctx*, const mo_orb_params*, const uint8_t* gray, int w, int h, int level,
                      int32_t* xys /* [cap][3] */, int cap, int* n);
int mo_dbg_min_eigen(mo_ctx*, const uint8_t* gray, int w, int h, float* eig /* [h*w] */);
int mo_dbg_set_poison(mo_ctx*, int byte /* 0..255: the pyramid buffers are filled with it before every extraction; -1: off */);
int mo_dbg_retain_best(mo_ctx*, const float* resp, int n, int n_points, int select_order, int32_t* order, int* n_out);

#ifdef __cplusplus
}
#endif
#endif

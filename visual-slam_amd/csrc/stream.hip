// stream.hip -- mo_stream: a frame sequence through the BATCHED mode in chunks, host frames in / host results out, with the upload
// of chunk i + 1 and the host-side collection of chunk i - 1 overlapped with the compute of chunk i.
//
// The reference's driver hands frames over one at a time (src/tests/tester_map.py:57-75; Tracker.process_frame, tracker.py:73-146) and
// pays a launch + synchronisation round trip per frame.  A caller that can look a few frames ahead (a video file, a recorded
// sequence, a camera with a queue) gets the batched mode's rate instead: frames are cut into chunks of `chunk`, every chunk is one
// mo_dev_frontend_batch call (detector + pair mode of the caller's choice) on chunk + 1 frames - the LAST frame of the previous chunk is
// staged again in front (halo: re-extracted rather than carried over, like the multi-GPU shards) so that every consecutive pair of the
// sequence is matched exactly once - and the sampling stream of a pair is keyed by its GLOBAL index (mo_batch_io.pair_index_base): poses
// equal those of a per-frame loop that counts its pairs (mo_pair_params.pair_index), bit for bit.
//
// Three lanes (pinned host input, device frames, device results, pinned host results, events) take turns:
//   submit(k):  host threads copy the caller's frames into the lane's pinned input; copy stream: H2D; compute stream (the context's):
//               wait for the copy, (BGR -> gray), mo_dev_frontend_batch, flag words into the result slab; download stream: results -> the
//               lane's pinned output (the compute stream is free for the next chunk while they travel)
//   collect():  waits for the oldest submitted chunk's results only (an event, not the stream: the next chunk keeps running)
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace {

inline size_t al(size_t v, size_t a) { return (v + a - 1) & ~(a - 1); }

// a few persistent host threads for the staging copies (spawning seven threads per chunk cost 0.2 ms of a 1.1 ms chunk period)
class CopyPool {
public:
    explicit CopyPool(int n) {
        for (int t = 0; t < n; t++) th_.emplace_back([this, t] { loop(t); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (std::thread& t : th_) t.join();
    }
    int size() const { return (int)th_.size() + 1; }
    // runs work(t) for t = 0 .. size() - 1 (the caller takes the last index itself) and returns when all are done
    void run(const std::function<void(int)>& work) {
        { std::lock_guard<std::mutex> g(m_); work_ = &work; left_ = (int)th_.size(); gen_++; }
        cv_.notify_all();
        work((int)th_.size());
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_ == 0; });
        work_ = nullptr;
    }
private:
    void loop(int t) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* w;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                w = work_;
            }
            (*w)(t);
            { std::lock_guard<std::mutex> g(m_); left_--; }
            done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* work_ = nullptr;
    int left_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

struct Lane {
    uint8_t* h_in = nullptr;    // pinned [chunk + 1][h][w][ch]
    uint8_t* d_in = nullptr;    // device, the same
    uint8_t* d_gray = nullptr;  // device [chunk + 1][h][w] (ch == 3 only; ch == 1: d_in is the gray batch)
    uint8_t* d_out = nullptr;   // device results, one slab (offsets in mo_stream)
    uint8_t* h_out = nullptr;   // pinned, the same layout
    hipEvent_t copied = nullptr, computed = nullptr, done = nullptr;
    int n_frames = 0;           // frames of the chunk in flight (without the halo); 0 = lane free
    int halo = 0;               // 1: frame 0 of the batch is the previous chunk's last frame
    uint64_t first_frame = 0;   // global index of the chunk's first own frame
};

}  // namespace

// chunks in flight: one being staged / uploaded, one computing, one downloading / being read by the caller - and more than that when the
// caller's own work per chunk sits beside a chunk's latency (staging + upload + compute): the steady-state period is bounded by
// (latency + caller time per chunk) / (lanes - 1); A/B of 3 / 4 / 5: profiles/r04_ab_stream_lanes.txt.  mo_stream_lanes() reports it.
#ifndef MO_STREAM_LANES
#define MO_STREAM_LANES 3
#endif

struct mo_stream {
    mo_ctx* c = nullptr;
    mo_orb_params orb{};
    mo_stream_params p{};
    hipStream_t copy_s = nullptr, down_s = nullptr;  // uploads / downloads, beside the context's stream (compute)
    Lane lane[MO_STREAM_LANES];
    size_t frame_in = 0, frame_px = 0;
    // result slab offsets (rows: B = chunk + 1 frames, P = chunk pairs)
    size_t o_flags = 0, o_counts = 0, o_kps = 0, o_desc = 0, o_midx = 0, o_mdist = 0, o_mpass = 0, o_sel = 0, o_seld = 0, o_seln = 0, o_pose = 0,
           o_mask = 0, o_npts = 0, o_pts = 0, out_bytes = 0;
    uint64_t submitted = 0, collected = 0;  // chunks
    uint64_t frames_in = 0;                 // frames submitted so far
    CopyPool* pool = nullptr;
    std::string err;
};

static int sfail(mo_stream* s, int code, const std::string& m) { s->err = m; if (s->c) mo_fail(s->c, code, m); return code; }
#define SCHK(s, expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return sfail((s), MO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } while (0)

extern "C" void mo_stream_destroy(mo_stream* s) {
    if (!s) return;
    if (s->c) { hipSetDevice(s->c->device); hipStreamSynchronize(s->c->stream); }
    if (s->copy_s) { hipStreamSynchronize(s->copy_s); hipStreamDestroy(s->copy_s); }
    if (s->down_s) { hipStreamSynchronize(s->down_s); hipStreamDestroy(s->down_s); }
    for (Lane& l : s->lane) {
        if (l.h_in) hipHostFree(l.h_in);
        if (l.h_out) hipHostFree(l.h_out);
        if (l.d_in) hipFree(l.d_in);
        if (l.d_gray) hipFree(l.d_gray);
        if (l.d_out) hipFree(l.d_out);
        if (l.copied) hipEventDestroy(l.copied);
        if (l.done) hipEventDestroy(l.done);
        if (l.computed) hipEventDestroy(l.computed);
    }
    delete s->pool;
    delete s;
}

extern "C" mo_stream* mo_stream_create(mo_ctx* c, const mo_orb_params* orb, const mo_stream_params* p) {
    if (!c) return nullptr;
    if (!orb || !p) { mo_fail(c, MO_ERR_ARG, "mo_stream_create: NULL argument"); return nullptr; }
    if (p->chunk < 1 || p->chunk + 1 > c->max_batch) { mo_fail(c, MO_ERR_ARG, "mo_stream_create: chunk + 1 (halo frame) must fit the context's max_batch"); return nullptr; }
    if (p->w < 64 || p->h < 64 || p->w > c->max_w || p->h > c->max_h || (p->ch != 1 && p->ch != 3) || p->cap < 1) {
        mo_fail(c, MO_ERR_ARG, "mo_stream_create: bad frame size / channel count / cap"); return nullptr;
    }
    if ((p->mode != MO_MODE_INIT && p->mode != MO_MODE_TRACK) || (p->detector != MO_DETECT_ORB && p->detector != MO_DETECT_GRID)) {
        mo_fail(c, MO_ERR_ARG, "mo_stream_create: mode must be MO_MODE_INIT / MO_MODE_TRACK, detector MO_DETECT_ORB / MO_DETECT_GRID"); return nullptr;
    }
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    mo_stream* s = new mo_stream();
    s->c = c; s->orb = *orb; s->p = *p;
    s->frame_px = (size_t)p->w * p->h;
    s->frame_in = s->frame_px * p->ch;
    const size_t B = (size_t)p->chunk + 1, P = (size_t)p->chunk, cap = (size_t)p->cap;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes, 256); return o; };
    s->o_flags = take(16); s->o_counts = take(B * 4); s->o_kps = take(B * cap * sizeof(mo_keypoint)); s->o_desc = take(B * cap * 32);
    s->o_sel = take(P * cap * 8); s->o_seld = take(P * cap * 4); s->o_seln = take(P * 4);
    s->o_pose = take(P * 12 * sizeof(double)); s->o_mask = take(P * cap); s->o_npts = take(P * 4);
    s->o_midx = take(P * cap * 8); s->o_mdist = take(P * cap * 8); s->o_mpass = take(P * cap);
    s->o_pts = take(p->want_points ? P * cap * 3 * sizeof(float) : 16);
    s->out_bytes = off;
    bool ok = hipStreamCreateWithFlags(&s->copy_s, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&s->down_s, hipStreamNonBlocking) == hipSuccess;
    for (Lane& l : s->lane) {
        ok = ok && hipHostMalloc((void**)&l.h_in, B * s->frame_in, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&l.h_out, s->out_bytes, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipMalloc((void**)&l.d_in, B * s->frame_in) == hipSuccess;
        if (p->ch == 3) ok = ok && hipMalloc((void**)&l.d_gray, B * s->frame_px) == hipSuccess;
        ok = ok && hipMalloc((void**)&l.d_out, s->out_bytes) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&l.copied, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&l.done, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&l.computed, hipEventDisableTiming) == hipSuccess;
        if (ok) hipMemset(l.d_out, 0, s->out_bytes);
    }
    if (!ok) { mo_fail(c, MO_ERR_HIP, "mo_stream_create: allocation failed"); mo_stream_destroy(s); return nullptr; }
    const size_t chunk_mb = (size_t)p->chunk * s->frame_in >> 20;
    s->pool = new CopyPool((int)std::min<size_t>(7, chunk_mb / 2));  // (+ the calling thread; small chunks are copied by the caller alone)
    return s;
}

// the caller's frames -> the lane's pinned input, rows of `stride` bytes, on the pool's host threads (a 64-frame chunk is 19.7 MB: one
// core's memcpy would cap the stream near 30 k frames/s)
static void stage_frames(CopyPool* pool, uint8_t* dst, const uint8_t* src, int n, size_t frame_in, size_t row, int h, size_t stride, size_t frame_stride) {
    const int nthreads = pool->size();
    std::function<void(int)> work = [=](int t) {
        for (int f = t; f < n; f += nthreads) {
            const uint8_t* sf = src + (size_t)f * frame_stride;
            uint8_t* df = dst + (size_t)f * frame_in;
            if (stride == row) std::memcpy(df, sf, frame_in);
            else for (int y = 0; y < h; y++) std::memcpy(df + (size_t)y * row, sf + (size_t)y * stride, row);
        }
    };
    pool->run(work);
}

extern "C" int mo_stream_submit(mo_stream* s, const uint8_t* frames, int n, int stride, size_t frame_stride) {
    if (!s) return MO_ERR_ARG;
    if (!frames || n < 1 || n > s->p.chunk) return sfail(s, MO_ERR_ARG, "mo_stream_submit: 1 <= n <= chunk frames");
    const size_t row = (size_t)s->p.w * s->p.ch;
    if (stride == 0) stride = (int)row;
    if (frame_stride == 0) frame_stride = (size_t)stride * s->p.h;
    if ((size_t)stride < row) return sfail(s, MO_ERR_ARG, "mo_stream_submit: stride smaller than a row");
    if (s->submitted - s->collected >= MO_STREAM_LANES) return sfail(s, MO_ERR_CAPACITY, "mo_stream_submit: every lane holds an uncollected chunk (call mo_stream_collect)");
    mo_ctx* c = s->c;
    SCHK(s, hipSetDevice(c->device));
    Lane& l = s->lane[s->submitted % MO_STREAM_LANES];
    const Lane& prev = s->lane[(s->submitted + MO_STREAM_LANES - 1) % MO_STREAM_LANES];
    const int halo = s->frames_in > 0 ? 1 : 0;
    // (the lane's buffers are free: its previous chunk was collected, i.e. its `done` event has been waited for)
    if (halo) {  // the previous chunk's last frame again, in front
        const Lane& src = s->submitted > 0 ? prev : l;
        std::memcpy(l.h_in, src.h_in + (size_t)(src.halo + src.n_frames - 1) * s->frame_in, s->frame_in);
    }
    stage_frames(s->pool, l.h_in + (size_t)halo * s->frame_in, frames, n, s->frame_in, row, s->p.h, (size_t)stride, frame_stride);
    l.n_frames = n; l.halo = halo; l.first_frame = s->frames_in;
    const int nb = n + halo;
    SCHK(s, hipMemcpyAsync(l.d_in, l.h_in, (size_t)nb * s->frame_in, hipMemcpyHostToDevice, s->copy_s));
    SCHK(s, hipEventRecord(l.copied, s->copy_s));
    SCHK(s, hipStreamWaitEvent(c->stream, l.copied, 0));
    const uint8_t* d_gray = l.d_in;
    int rc;
    if (s->p.ch == 3) {
        if ((rc = orb_launch_gray(c, l.d_in, s->p.w, s->p.h, nb, l.d_gray))) return rc;
        d_gray = l.d_gray;
    }
    uint8_t* o = l.d_out;
    mo_batch_io io;
    std::memset(&io, 0, sizeof(io));
    io.d_gray = d_gray; io.w = s->p.w; io.h = s->p.h; io.batch = nb; io.cap = s->p.cap;
    io.ratio = s->p.ratio; io.thr_px = s->p.thr_px; io.n_hyp = s->p.n_hyp; io.seed = s->p.seed;
    for (int i = 0; i < 9; i++) io.K[i] = s->p.K[i];
    io.d_kps = (mo_keypoint*)(o + s->o_kps); io.d_desc = o + s->o_desc; io.d_counts = (int32_t*)(o + s->o_counts);
    io.d_match_idx = (int32_t*)(o + s->o_midx); io.d_match_dist = (int32_t*)(o + s->o_mdist); io.d_match_pass = o + s->o_mpass;
    io.d_pose = (double*)(o + s->o_pose); io.d_n_points = (int32_t*)(o + s->o_npts); io.d_pose_mask = o + s->o_mask;
    // (map points are written per query keypoint whether or not the caller wants them back: the slab of a chunk without want_points
    //  is the context's own two-view scratch, never copied)
    float* d_pts = nullptr;
    if (s->p.want_points) d_pts = (float*)(o + s->o_pts);
    else {
        if ((rc = mo_reserve(c, c->d_stream_pts, c->stream_pts_bytes, (size_t)s->p.chunk * s->p.cap * 3 * sizeof(float)))) return rc;
        d_pts = c->d_stream_pts;
    }
    io.d_points = d_pts;
    io.mode = s->p.mode; io.disp_frac = s->p.disp_frac; io.detector = s->p.detector;
    io.d_sel_idx = (int32_t*)(o + s->o_sel); io.d_sel_dist = (int32_t*)(o + s->o_seld); io.d_sel_n = (int32_t*)(o + s->o_seln);
    // pair 0 of this batch is (halo frame, first own frame) = global pair first_frame - 1; without a halo it is global pair first_frame
    io.pair_index_base = s->p.pair_index_base + (halo ? l.first_frame - 1 : l.first_frame);
    if ((rc = mo_dev_frontend_batch(c, &s->orb, &io))) return rc;
    // results -> pinned host: the flag words of this call travel with them (and are cleared for the next chunk)
    SCHK(s, hipMemcpyAsync(o + s->o_flags, c->d_flags, 16, hipMemcpyDeviceToDevice, c->stream));
    SCHK(s, hipMemsetAsync(c->d_flags, 0, 16, c->stream));
    // ... on a THIRD stream: the compute stream goes straight on to the next chunk, and the upload stream is never held up behind a
    // download that waits for a compute (one copy stream for both directions serialised upload (k + 1) behind compute (k): 38 k frames/s
    // at any chunk size)
    const size_t upto = s->p.mode == MO_MODE_TRACK && !s->p.want_matches ? s->o_midx : s->p.want_points ? s->out_bytes : s->o_pts;
    SCHK(s, hipEventRecord(l.computed, c->stream));
    SCHK(s, hipStreamWaitEvent(s->down_s, l.computed, 0));
    // This call, 4 us as a rule, holds the host for 5.6 - 7.3 ms at chunks 2, 6 and 11 of the FIRST stream of a process and never again (later
    // streams, new contexts included, are flat): a first run of 64 chunks reads 47 - 52 k frames/s where the steady rate is 60 k.  Tried
    // (profiles/r04_ab_stream_d2h.txt): a copy kernel into a mapped buffer - no stalls, but it shares the CUs with the next chunk's compute:
    // 51.6 k against 60 k frames/s in steady state; a dozen result-sized downloads at creation - the stalls stay where they were.
    SCHK(s, hipMemcpyAsync(l.h_out, o, upto, hipMemcpyDeviceToHost, s->down_s));
    SCHK(s, hipEventRecord(l.done, s->down_s));
    s->submitted++;
    s->frames_in += n;
    return MO_OK;
}

extern "C" int mo_stream_lanes(void) { return MO_STREAM_LANES; }

extern "C" int mo_stream_collect(mo_stream* s, mo_stream_result* r) {
    if (!s || !r) return MO_ERR_ARG;
    if (s->collected >= s->submitted) return sfail(s, MO_ERR_ARG, "mo_stream_collect: nothing submitted");
    SCHK(s, hipSetDevice(s->c->device));
    Lane& l = s->lane[s->collected % MO_STREAM_LANES];
    SCHK(s, hipEventSynchronize(l.done));
    const uint8_t* h = l.h_out;
    const size_t cap = (size_t)s->p.cap;
    std::memset(r, 0, sizeof(*r));
    r->n_frames = l.n_frames; r->n_pairs = l.n_frames - 1 + l.halo; r->first_frame = l.first_frame;
    r->first_pair = l.halo ? l.first_frame - 1 : l.first_frame;
    r->cap = s->p.cap;
    r->flags = *(const int32_t*)(h + s->o_flags);
    // frame rows start behind the halo frame; pair rows at 0 (pair 0 = (halo, first own frame) or (frame 0, frame 1))
    r->counts = (const int32_t*)(h + s->o_counts) + l.halo;
    r->kps = (const mo_keypoint*)(h + s->o_kps) + (size_t)l.halo * cap;
    r->desc = h + s->o_desc + (size_t)l.halo * cap * 32;
    r->prev_count = l.halo ? ((const int32_t*)(h + s->o_counts))[0] : 0;
    r->sel_idx = (const int32_t*)(h + s->o_sel); r->sel_dist = (const int32_t*)(h + s->o_seld); r->sel_n = (const int32_t*)(h + s->o_seln);
    r->pose = (const double*)(h + s->o_pose); r->pose_mask = h + s->o_mask; r->n_points = (const int32_t*)(h + s->o_npts);
    const bool have_matches = !(s->p.mode == MO_MODE_TRACK && !s->p.want_matches);
    r->match_idx = have_matches ? (const int32_t*)(h + s->o_midx) : nullptr;
    r->match_dist = have_matches ? (const int32_t*)(h + s->o_mdist) : nullptr;
    r->match_pass = have_matches ? h + s->o_mpass : nullptr;
    r->points = s->p.want_points ? (const float*)(h + s->o_pts) : nullptr;
    s->collected++;
    // (the lane stays described until its next submit: the halo copy of the following chunk reads its last frame)
    if (r->flags & 15) return sfail(s, MO_ERR_CAPACITY, "a capacity flag was raised inside a streamed chunk (mo_stream_result.flags; see mo_dev_status)");
    return MO_OK;
}

extern "C" const char* mo_stream_last_error(mo_stream* s) { return s ? s->err.c_str() : ""; }

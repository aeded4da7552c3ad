#!/bin/bash
# diagnostic variant of mo_stream_submit: prints the wall time of its sections whenever a call takes longer than 2 ms (never shipped)
# -> visual-slam_amd/variants/libsubmit_timing.so; run: VSLAM_AMD_LIB=... python tools/stream_first.py
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/visual-slam_amd/variants"
tmp=$(mktemp -d /tmp/abl.XXXX)
mkdir -p "$tmp/visual-slam_amd" "$tmp/include"
cp -r "$root/visual-slam_amd/csrc" "$tmp/visual-slam_amd/"; cp "$root/include/vslam_amd.h" "$tmp/include/"
rm -rf "$tmp/visual-slam_amd/csrc/_obj"
python3 - "$tmp/visual-slam_amd/csrc/stream.hip" <<'PY'
import sys
f = sys.argv[1]; s = open(f).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a
    s = s.replace(a, b, 1)
rep("    mo_ctx* c = s->c;\n    SCHK(s, hipSetDevice(c->device));\n    Lane& l = s->lane[s->submitted % MO_STREAM_LANES];",
    "    mo_ctx* c = s->c;\n    const double T0 = now_us();\n    SCHK(s, hipSetDevice(c->device));\n    Lane& l = s->lane[s->submitted % MO_STREAM_LANES];")
rep("    l.n_frames = n; l.halo = halo; l.first_frame = s->frames_in;\n", "    l.n_frames = n; l.halo = halo; l.first_frame = s->frames_in;\n    const double T1 = now_us();\n")
rep("    const uint8_t* d_gray = l.d_in;\n    int rc;\n", "    const double T2 = now_us();\n    const uint8_t* d_gray = l.d_in;\n    int rc;\n")
rep("    if ((rc = mo_dev_frontend_batch(c, &s->orb, &io))) return rc;\n", "    const double T3 = now_us();\n    if ((rc = mo_dev_frontend_batch(c, &s->orb, &io))) return rc;\n    const double T4 = now_us();\n")
rep("    SCHK(s, hipMemcpyAsync(o + s->o_flags, c->d_flags, 16, hipMemcpyDeviceToDevice, c->stream));\n", "    SCHK(s, hipMemcpyAsync(o + s->o_flags, c->d_flags, 16, hipMemcpyDeviceToDevice, c->stream));\n    const double U1 = now_us();\n")
rep("    SCHK(s, hipMemsetAsync(c->d_flags, 0, 16, c->stream));\n", "    SCHK(s, hipMemsetAsync(c->d_flags, 0, 16, c->stream));\n    const double U2 = now_us();\n")
rep("    SCHK(s, hipEventRecord(l.computed, c->stream));\n", "    SCHK(s, hipEventRecord(l.computed, c->stream));\n    const double U3 = now_us();\n")
rep("    SCHK(s, hipStreamWaitEvent(s->down_s, l.computed, 0));\n", "    SCHK(s, hipStreamWaitEvent(s->down_s, l.computed, 0));\n    const double U4 = now_us();\n")
rep("    SCHK(s, hipMemcpyAsync(l.h_out, o, upto, hipMemcpyDeviceToHost, s->down_s));\n", "    SCHK(s, hipMemcpyAsync(l.h_out, o, upto, hipMemcpyDeviceToHost, s->down_s));\n    const double U5 = now_us();\n")
rep("    s->submitted++;\n    s->frames_in += n;\n    return MO_OK;",
    "    const double T5 = now_us();\n    if (T5 - T0 > 2000.0) fprintf(stderr, \"SUBMIT chunk %llu: staging %.0f us, upload enqueue %.0f, io setup %.0f, mo_dev_frontend_batch %.0f, download enqueue %.0f (flag copy %.0f, flag memset %.0f, event record %.0f, stream wait %.0f, D2H memcpyAsync of %zu bytes %.0f), total %.0f\\n\", (unsigned long long)s->submitted, T1 - T0, T2 - T1, T3 - T2, T4 - T3, T5 - T4, U1 - T4, U2 - U1, U3 - U2, U4 - U3, upto, U5 - U4, T5 - T0);\n    s->submitted++;\n    s->frames_in += n;\n    return MO_OK;")
open(f, "w").write(s)
PY
make -C "$tmp/visual-slam_amd/csrc" -j8 2>&1 | grep -E "error" -A3 | head
cp "$tmp/visual-slam_amd/libvslam_amd.so" "$root/visual-slam_amd/variants/libsubmit_timing.so"
rm -rf "$tmp"; echo built

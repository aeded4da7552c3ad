// undistort_kernels.hip -- cv2.undistort(image, K, dist) on the device (reference src/orbslam2/utils.py:40-52, applied by
// src/run_video.py:145-149 when any distortion coefficient is non-zero).
//
// Restates cv2's pipeline for 8-bit images: initUndistortRectifyMap(K, dist, I, K, CV_16SC2) + remap(INTER_LINEAR,
// BORDER_CONSTANT 0).  Per output pixel (j, i): x = (j - cx) / fx, y = (i - cy) / fy, r2 = x^2 + y^2,
//   kr = 1 + ((k3 r2 + k2) r2 + k1) r2,  xd = x kr + 2 p1 x y + p2 (r2 + 2 x^2),  yd = y kr + p1 (r2 + 2 y^2) + 2 p2 x y,
//   u = fx xd + cx, v = fy yd + cy                                   (all in double, like cv2)
// the map is quantised to 1/32 pixel, iu = round_half_even(32 u): source pixel (iu >> 5, iv >> 5), fractions a = iu & 31,
// b = iv & 31; cv2's fixed-point bilinear table for 32 steps is exact ((32 - a)(32 - b) * 32 of 2^15, no renormalisation), so
//   out = (sum of (32 - a | a)(32 - b | b) * p + 512) >> 10,   source pixels outside the image count as 0.
// dist = (k1, k2, p1, p2, k3), the five coefficients of configs/monocular.yaml.
#include "common.h"

struct UndistortArgs { double fx, fy, cx, cy, k1, k2, p1, p2, k3; };

__global__ __launch_bounds__(256) void k_undistort(UndistortArgs A, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w,
                                                   int h, int ch, size_t frame_bytes) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= w || i >= h) return;
    const uint8_t* s = src + (size_t)blockIdx.z * frame_bytes;
    uint8_t* d = dst + (size_t)blockIdx.z * frame_bytes;
    const double x = ((double)j - A.cx) / A.fx, y = ((double)i - A.cy) / A.fy;
    const double x2 = x * x, y2 = y * y, r2 = x2 + y2, xy2 = 2 * x * y;
    const double kr = 1 + ((A.k3 * r2 + A.k2) * r2 + A.k1) * r2;
    const double xd = x * kr + A.p1 * xy2 + A.p2 * (r2 + 2 * x2), yd = y * kr + A.p1 * (r2 + 2 * y2) + A.p2 * xy2;
    const double u = A.fx * xd + A.cx, v = A.fy * yd + A.cy;
    const double cl = 1e8;  // far outside any image; keeps the conversions below defined
    const int iu = (int)rint(fmin(fmax(u * 32.0, -cl), cl)), iv = (int)rint(fmin(fmax(v * 32.0, -cl), cl));
    const int sx = iu >> 5, sy = iv >> 5, a = iu & 31, b = iv & 31;
    const int w00 = (32 - a) * (32 - b), w01 = a * (32 - b), w10 = (32 - a) * b, w11 = a * b;
    const bool x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w, y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h;
    for (int c = 0; c < ch; c++) {
        const int p00 = x0 && y0 ? s[((size_t)sy * w + sx) * ch + c] : 0, p01 = x1 && y0 ? s[((size_t)sy * w + sx + 1) * ch + c] : 0;
        const int p10 = x0 && y1 ? s[((size_t)(sy + 1) * w + sx) * ch + c] : 0, p11 = x1 && y1 ? s[((size_t)(sy + 1) * w + sx + 1) * ch + c] : 0;
        d[((size_t)i * w + j) * ch + c] = (uint8_t)((w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11 + 512) >> 10);
    }
}

int undistort_launch(mo_ctx* c, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int ch, int batch, const double K[9],
                     const double dist[5]) {
    UndistortArgs A = {K[0], K[4], K[2], K[5], dist[0], dist[1], dist[2], dist[3], dist[4]};
    hipLaunchKernelGGL(k_undistort, dim3((w + 63) / 64, (h + 3) / 4, batch), dim3(256), 0, c->stream, A, d_src, d_dst, w, h, ch,
                       (size_t)w * h * ch);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

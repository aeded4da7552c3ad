// twoview_kernels.hip -- two-view initialisation on gfx950 (fp64 vector ALU; no MFMA: tiny per-thread solves).
// Replaces, for the reference's MapInitializer.initialize (src/orbslam2/initializer.py:75-120):
//   utils.py:120-126  cv2.findEssentialMat(p1, p2, K, RANSAC, prob, threshold)  -> k_tv_hyp + k_tv_finish
//   utils.py:129-134  cv2.recoverPose(E, p1, p2, K, mask)                        -> k_tv_finish (cheirality vote)
//   utils.py:56-70    cv2.triangulatePoints + divide by w                        -> k_tv_finish (DLT)
// north_star prescribes an 8-point essential-matrix RANSAC with thousands of hypotheses scored in parallel
// (cv2 itself runs a sequential 5-point RANSAC), so parity here is on the recovered R, t, X (1e-4 rel), not
// on hypothesis-level equality with cv2.
//
//   k_tv_prep    per pair: gather the ratio-test survivors (query order) -> normalised f64 correspondences
//   k_tv_hyp     one thread per hypothesis: counter-based 8-sample, 8x9 null vector by Householder QR (registers),
//                closed-form projection on the essential manifold (smallest_right3 / project_essential), then a float32
//                LOWER BOUND of the MSAC cost (truncated Sampson distance) over the first quarter of the correspondences,
//                which arrive as scalar operands; the block's most promising hypothesis is scored completely in fp64 ->
//                an upper bound of the pair's best cost; hypotheses whose bound already exceeds it cannot win, only the
//                others (about a quarter) are written out, compacted
//   k_tv_tasks / k_tv_score   the survivors of the whole launch are cut into tasks of 64 (dense table: busy workgroups first,
//                evenly spread over the chip); one workgroup per task sums their canonical fp64 costs (four contiguous
//                parts, one per wavefront) -> exact argmin, one atomicMin per task
//   k_tv_finish  one block per pair: consensus set of the best hypothesis -> adaptive-threshold least-squares
//                8-point refits (9x9 normal matrix, inverse iteration) -> final inliers -> decompose E -> cheirality
//                vote over the 4 (R, t) candidates with per-point DLT -> final DLT triangulation in pixel space
#include <cmath>

#include "common.h"

#define TV_BLOCK 256
typedef double v4d __attribute__((ext_vector_type(4)));
#define TVF_BLOCK 512  // k_tv_finish: one block per pair, about one correspondence per thread in the per-point phases

#define TV_PARTS 4  // the canonical cost is summed in this many contiguous parts of the correspondences (tv_cost_lane, k_tv_score)
#define TV_REC 10  // doubles per survivor record: E (9), hypothesis index (as its bit pattern)

struct TvWork {
    double* xn;        // [pairs][cap][4] normalised x1,y1,x2,y2
    float* xf;         // [pairs][cap][4] the same rounded to float32 (first scoring stage)
    float* px;         // [pairs][cap][4] pixel u1,v1,u2,v2
    int* qidx;         // [pairs][cap] query keypoint index of correspondence i
    uint8_t* cbits;    // [pairs][cap] k_tv_finish: cheirality bits of correspondence i (4 candidates + "considered")
    int* m;            // [pairs]
    double* surv;      // [pairs][n_hyp][TV_REC] survivors of the first scoring stage, dense from index 0 (any order)
    int* n_alive;      // [pairs] survivors listed so far
    int* n_tasks;      // [1] number of 64-survivor tasks of the whole launch
    int2* task;        // [pairs * ceil(n_hyp / 64)] (pair, task of the pair) of each task, dense from index 0
    unsigned long long* wkey;  // [pairs][ceil(n_hyp / 64)] best key of each k_tv_score wavefront (or k_tv_hyp block when unstaged)
    double* wE;        // [pairs][ceil(n_hyp / 64)][9] its matrix: k_tv_finish takes the one whose key equals best
    unsigned long long* best;  // [pairs] (float32 bits of the MSAC cost << 32) | hypothesis index, minimum wins
    double* norm;      // [pairs][8] fundamental-matrix model: common scale s, centroid 1 (x, y), centroid 2 (x, y) of the
                       // Hartley normalisation x_n = s (x - c); Sampson distances scale by s^2, so thr_n = thr_px * s;
                       // [5] = max(1, largest |coordinate| in xn) for the error bounds of the float32 stage;
                       // [6] (as 32 bits) = smallest candidate total posted by a k_tv_hyp block of the pair so far
};

size_t twoview_workspace_bytes(int n_pairs, int cap, int n_hyp) {
    size_t p = (size_t)n_pairs, nt = (size_t)((n_hyp + 63) / 64);
    return p * cap * 4 * sizeof(double) + 2 * p * cap * 4 * sizeof(float) + p * cap * sizeof(int) + p * cap + p * sizeof(int) * 2 + 16 +
           p * (size_t)n_hyp * TV_REC * sizeof(double) + p * nt * (sizeof(int2) + sizeof(unsigned long long) + 9 * sizeof(double)) +
           p * sizeof(unsigned long long) + p * 8 * sizeof(double) + 1024;
}

static TvWork carve(void* base, int n_pairs, int cap, int n_hyp) {
    TvWork w;
    uint8_t* b = (uint8_t*)base;
    size_t p = (size_t)n_pairs, nt = (size_t)((n_hyp + 63) / 64);
    w.xn = (double*)b; b += p * cap * 4 * sizeof(double);
    w.surv = (double*)b; b += p * (size_t)n_hyp * TV_REC * sizeof(double);
    w.wE = (double*)b; b += p * nt * 9 * sizeof(double);
    w.wkey = (unsigned long long*)b; b += p * nt * sizeof(unsigned long long);
    w.best = (unsigned long long*)b; b += p * sizeof(unsigned long long);
    w.norm = (double*)b; b += p * 8 * sizeof(double);
    w.task = (int2*)b; b += p * nt * sizeof(int2);
    w.px = (float*)b; b += p * cap * 4 * sizeof(float);
    w.xf = (float*)b; b += p * cap * 4 * sizeof(float);
    w.qidx = (int*)b; b += p * cap * sizeof(int);
    w.m = (int*)b; b += p * sizeof(int);
    w.n_alive = (int*)b; b += p * sizeof(int);
    w.n_tasks = (int*)b; b += 16;
    w.cbits = b;
    return w;
}

// ---------------------------------------------------------------- small dense helpers -------------
// cyclic Jacobi eigen-decomposition of a symmetric NxN matrix (row-major a, eigenvectors in columns of v)
template <int N> __device__ void jacobi_eig(double* a, double* v) {
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) v[i * N + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0, diag = 0;
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int j = 0; j < N; j++) {
                if (i != j) off += a[i * N + j] * a[i * N + j];
                else diag += a[i * N + j] * a[i * N + j];
            }
        if (off <= 1e-30 * diag || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < N - 1; p++)
#pragma unroll
            for (int q = p + 1; q < N; q++) {
                double apq = a[p * N + q];
                if (apq != 0.0) {
                    double theta = (a[q * N + q] - a[p * N + p]) / (2.0 * apq);
                    double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        double akp = a[k * N + p], akq = a[k * N + q];
                        a[k * N + p] = c * akp - s * akq;
                        a[k * N + q] = s * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        double apk = a[p * N + k], aqk = a[q * N + k];
                        a[p * N + k] = c * apk - s * aqk;
                        a[q * N + k] = s * apk + c * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        double vkp = v[k * N + p], vkq = v[k * N + q];
                        v[k * N + p] = c * vkp - s * vkq;
                        v[k * N + q] = s * vkp + c * vkq;
                    }
                }
            }
    }
}

// Smallest eigenvector of a symmetric positive semi-definite NxN matrix (upper triangle packed row-major, p <= q) by
// inverse iteration on M + delta*I with an LDL^T factorisation; x holds the starting vector (unit length) and receives
// the result.  All indices are compile-time constants (registers).  Converges at the rate lambda_min / lambda_next per
// step: a couple of steps on clean data, more on noisy data -> at most max_it steps with an early exit once the
// iterate stops moving (|y - x|^2 < 1e-28, measured on the difference so rounding in a dot product cannot hide it).
template <int N> __device__ void smallest_eigvec(const double* Msym, double* x, int max_it) {
    double L[N][N], D[N];
    double tr = 0;
#pragma unroll
    for (int p = 0, k = 0; p < N; p++)
#pragma unroll
        for (int q = p; q < N; q++, k++) { L[q][p] = Msym[k]; if (p == q) tr += Msym[k]; }
    const double delta = 1e-15 * tr + 1e-300;
    // LDL^T in place: L[i][j] (i > j) unit lower factor, D diagonal (stored inverted)
#pragma unroll
    for (int j = 0; j < N; j++) {
        double d = L[j][j] + delta;
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
        d = d > 1e-300 ? d : 1e-300;
        D[j] = d;
        const double inv = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double v = L[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
            L[i][j] = v * inv;
        }
    }
    double Dinv[N];
#pragma unroll
    for (int i = 0; i < N; i++) Dinv[i] = 1.0 / D[i];
    double y[N];
    for (int it = 0; it < max_it; it++) {
#pragma unroll
        for (int i = 0; i < N; i++) y[i] = x[i];
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int k = 0; k < i; k++) y[i] -= L[i][k] * y[k];
#pragma unroll
        for (int i = 0; i < N; i++) y[i] *= Dinv[i];
#pragma unroll
        for (int i = N - 1; i >= 0; i--)
#pragma unroll
            for (int k = i + 1; k < N; k++) y[i] -= L[k][i] * y[k];
        double nn = 0, dot = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { nn += y[i] * y[i]; dot += y[i] * x[i]; }
        nn = (dot < 0 ? -1.0 : 1.0) / sqrt(nn);  // keep the orientation of the previous iterate
        double d2 = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            y[i] *= nn;
            const double d = y[i] - x[i];
            d2 += d * d;
            x[i] = y[i];
        }
        if (d2 < 1e-28) break;
    }
}

__device__ inline void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// SVD of a (near) rank-2 3x3 matrix E (row-major): right vectors from eig(E^T E), u_i = E v_i / sigma_i for the
// two largest, third vectors by cross products so det(U) = det(V) = +1.  Returns false when degenerate.
__device__ bool svd3_rank2(const double* E, double* U, double* V, double* sig) {
    double a[9], v[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) a[i * 3 + j] = E[0 * 3 + i] * E[0 * 3 + j] + E[1 * 3 + i] * E[1 * 3 + j] + E[2 * 3 + i] * E[2 * 3 + j];
    jacobi_eig<3>(a, v);
    // sort the three eigenpairs by descending eigenvalue with compare-exchanges on registers (no dynamic indexing)
    double l0 = a[0], l1 = a[4], l2 = a[8];
    double c0[3] = {v[0], v[3], v[6]}, c1[3] = {v[1], v[4], v[7]}, c2[3] = {v[2], v[5], v[8]};
#define CSWAP(la, lb, ca, cb)                                   \
    if (la < lb) {                                              \
        double t_ = la; la = lb; lb = t_;                       \
        for (int i_ = 0; i_ < 3; i_++) { double u_ = ca[i_]; ca[i_] = cb[i_]; cb[i_] = u_; } \
    }
    CSWAP(l0, l1, c0, c1)
    CSWAP(l1, l2, c1, c2)
    CSWAP(l0, l1, c0, c1)
#undef CSWAP
    double v1[3] = {c0[0], c0[1], c0[2]};
    double v2[3] = {c1[0], c1[1], c1[2]};
    double l[3] = {l0, l1, l2};
    const int o0 = 0, o1 = 1, o2 = 2;
    double s1 = sqrt(fmax(l[o0], 0.0)), s2 = sqrt(fmax(l[o1], 0.0));
    if (!(s2 > 1e-12 * s1) || !(s1 > 0)) return false;
    double u1[3], u2[3], u3[3], v3[3];
    for (int i = 0; i < 3; i++) {
        u1[i] = (E[i * 3] * v1[0] + E[i * 3 + 1] * v1[1] + E[i * 3 + 2] * v1[2]) / s1;
        u2[i] = (E[i * 3] * v2[0] + E[i * 3 + 1] * v2[1] + E[i * 3 + 2] * v2[2]) / s2;
    }
    // re-orthogonalise u2 against u1 (they are orthogonal up to rounding)
    double d = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
    for (int i = 0; i < 3; i++) u2[i] -= d * u1[i];
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    if (!(n1 > 0) || !(n2 > 0)) return false;
    for (int i = 0; i < 3; i++) { u1[i] /= n1; u2[i] /= n2; }
    cross3(u1, u2, u3);
    cross3(v1, v2, v3);
    for (int i = 0; i < 3; i++) {
        U[i * 3] = u1[i]; U[i * 3 + 1] = u2[i]; U[i * 3 + 2] = u3[i];
        V[i * 3] = v1[i]; V[i * 3 + 1] = v2[i]; V[i * 3 + 2] = v3[i];
    }
    sig[0] = s1; sig[1] = s2; sig[2] = sqrt(fmax(l[o2], 0.0));
    return true;
}

// Right singular vector v3 of the smallest singular value of a 3x3 matrix, without an eigen-solver: the adjugate of
// A = E^T E is  l1 l2 v3 v3' + l1 l3 v2 v2' + l2 l3 v1 v1'  (l = eigenvalues of A, descending), so v3 is its dominant
// eigenvector with the gap l2 / l3 = (sigma2 / sigma3)^2.  Eight squarings of the (symmetric) matrix raise the gap to
// the 256th power - below 1e-12 up to sigma3 / sigma2 = 0.95, i.e. far beyond any matrix that resembles an essential
// matrix - and leave v3 v3' up to scale; its longest column, normalised, is v3.  Straight-line code, about 250 fp64
// instructions (the cyclic Jacobi sweeps this replaces in the hypothesis kernel: about 2000).  Returns false for rank < 2.
__device__ bool smallest_right3(const double* E, double* v3) {
    const double a00 = E[0] * E[0] + E[3] * E[3] + E[6] * E[6], a01 = E[0] * E[1] + E[3] * E[4] + E[6] * E[7];
    const double a02 = E[0] * E[2] + E[3] * E[5] + E[6] * E[8], a11 = E[1] * E[1] + E[4] * E[4] + E[7] * E[7];
    const double a12 = E[1] * E[2] + E[4] * E[5] + E[7] * E[8], a22 = E[2] * E[2] + E[5] * E[5] + E[8] * E[8];
    const double tr = a00 + a11 + a22;
    double b00 = a11 * a22 - a12 * a12, b01 = a02 * a12 - a01 * a22, b02 = a01 * a12 - a02 * a11;
    double b11 = a00 * a22 - a02 * a02, b12 = a01 * a02 - a00 * a12, b22 = a00 * a11 - a01 * a01;
    const double c1 = b00 + b11 + b22;  // l1 l2 + l1 l3 + l2 l3
    if (!(tr > 1e-300) || !(c1 > 1e-24 * tr * tr)) return false;  // sigma2 <= 1e-12 sigma1
#pragma unroll
    for (int k = 0; k < 4; k++) {  // trace -> 1, then two squarings (the trace stays above 1/9)
        const double t = 1.0 / (b00 + b11 + b22);
        b00 *= t; b01 *= t; b02 *= t; b11 *= t; b12 *= t; b22 *= t;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double c00 = b00 * b00 + b01 * b01 + b02 * b02, c01 = b00 * b01 + b01 * b11 + b02 * b12;
            const double c02 = b00 * b02 + b01 * b12 + b02 * b22, c11 = b01 * b01 + b11 * b11 + b12 * b12;
            const double c12 = b01 * b02 + b11 * b12 + b12 * b22, c22 = b02 * b02 + b12 * b12 + b22 * b22;
            b00 = c00; b01 = c01; b02 = c02; b11 = c11; b12 = c12; b22 = c22;
        }
    }
    const double n0 = b00 * b00 + b01 * b01 + b02 * b02, n1 = b01 * b01 + b11 * b11 + b12 * b12, n2 = b02 * b02 + b12 * b12 + b22 * b22;
    const bool p0 = n0 >= n1 && n0 >= n2, p1 = !p0 && n1 >= n2;
    const double x = p0 ? b00 : p1 ? b01 : b02, y = p0 ? b01 : p1 ? b11 : b12, z = p0 ? b02 : p1 ? b12 : b22;
    const double nn = p0 ? n0 : p1 ? n1 : n2;
    if (!(nn > 0)) return false;
    const double r = 1.0 / sqrt(nn);
    v3[0] = x * r; v3[1] = y * r; v3[2] = z * r;
    return true;
}

// nearest essential matrix U diag(1, 1, 0) V' in closed form: with an orthonormal basis (a, b) of the plane orthogonal to
// v3, E maps that plane onto the plane orthogonal to u3, so u3 = (E a) x (E b) normalised; in the bases (a, b) and
// (c, d) = (E a / |E a|, u3 x c) the restriction of E is the upper-triangular M = [p q; 0 s] with p, s > 0, whose
// orthogonal polar factor is the rotation [p + s, q; -q, p + s] / h - and U diag(1, 1, 0) V' = [c d] polar(M) [a b]'.
__device__ bool project_essential(double* E) {
    double v[3];
    if (!smallest_right3(E, v)) return false;
    // a = v x e_k (normalised) for the axis k with the smallest |v_k|, b = v x a
    const double ax = fabs(v[0]), ay = fabs(v[1]), az = fabs(v[2]);
    const bool kx = ax <= ay && ax <= az, ky = !kx && ay <= az;
    double a[3] = {kx ? 0.0 : ky ? -v[2] : v[1], kx ? v[2] : ky ? 0.0 : -v[0], kx ? -v[1] : ky ? v[0] : 0.0};
    const double ra = 1.0 / sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);  // 1 - v_k^2 >= 2/3
    a[0] *= ra; a[1] *= ra; a[2] *= ra;
    double b[3];
    cross3(v, a, b);
    double Ea[3], Eb[3], u3[3];
    for (int i = 0; i < 3; i++) {
        Ea[i] = E[i * 3] * a[0] + E[i * 3 + 1] * a[1] + E[i * 3 + 2] * a[2];
        Eb[i] = E[i * 3] * b[0] + E[i * 3 + 1] * b[1] + E[i * 3 + 2] * b[2];
    }
    cross3(Ea, Eb, u3);
    const double p2 = Ea[0] * Ea[0] + Ea[1] * Ea[1] + Ea[2] * Ea[2], e2 = Eb[0] * Eb[0] + Eb[1] * Eb[1] + Eb[2] * Eb[2];
    const double nu2 = u3[0] * u3[0] + u3[1] * u3[1] + u3[2] * u3[2];  // (sigma1 sigma2)^2
    if (!(nu2 > 1e-24 * (p2 + e2) * (p2 + e2)) || !(p2 > 0)) return false;
    const double ru = 1.0 / sqrt(nu2), rp = 1.0 / sqrt(p2);
    double c[3] = {Ea[0] * rp, Ea[1] * rp, Ea[2] * rp}, d[3];
    u3[0] *= ru; u3[1] *= ru; u3[2] *= ru;
    cross3(u3, c, d);
    const double pp = p2 * rp, q = c[0] * Eb[0] + c[1] * Eb[1] + c[2] * Eb[2], sd = d[0] * Eb[0] + d[1] * Eb[1] + d[2] * Eb[2];
    const double rh = 1.0 / sqrt((pp + sd) * (pp + sd) + q * q);
    const double r0 = (pp + sd) * rh, r1 = q * rh;
    double g[3], h[3];  // rows of polar(M) [a b]'
    for (int j = 0; j < 3; j++) { g[j] = r0 * a[j] + r1 * b[j]; h[j] = r0 * b[j] - r1 * a[j]; }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) E[i * 3 + j] = c[i] * g[j] + d[i] * h[j];
    return true;
}

__device__ inline double sampson(const double* E, double x1, double y1, double x2, double y2) {
    double ex0 = E[0] * x1 + E[1] * y1 + E[2];
    double ex1 = E[3] * x1 + E[4] * y1 + E[5];
    double ex2 = E[6] * x1 + E[7] * y1 + E[8];
    double et0 = E[0] * x2 + E[3] * y2 + E[6];
    double et1 = E[1] * x2 + E[4] * y2 + E[7];
    double num = x2 * ex0 + y2 * ex1 + ex2;
    double den = ex0 * ex0 + ex1 * ex1 + et0 * et0 + et1 * et1;
    return num * num / den;
}

// same quantity for the MSAC ranking of hypotheses (compared after rounding to float32): explicit fused multiply-adds
// (24 fp64 instructions per correspondence) and the division replaced by a v_rcp_f32 seed (about 2^-23 accurate) + one
// fp64 Newton step (about 2^-46)
__device__ __forceinline__ double sampson_fast(const double* E, double x1, double y1, double x2, double y2) {
    const double ex0 = fma(E[0], x1, fma(E[1], y1, E[2]));
    const double ex1 = fma(E[3], x1, fma(E[4], y1, E[5]));
    const double ex2 = fma(E[6], x1, fma(E[7], y1, E[8]));
    const double et0 = fma(E[0], x2, fma(E[3], y2, E[6]));
    const double et1 = fma(E[1], x2, fma(E[4], y2, E[7]));
    const double num = fma(x2, ex0, fma(y2, ex1, ex2));
    const double den = fma(ex0, ex0, fma(ex1, ex1, fma(et0, et0, et1 * et1)));
    double r = (double)__builtin_amdgcn_rcpf((float)den);
    r = fma(fma(-den, r, 1.0), r, r);
    return num * num * r;
}

__device__ inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// 8 distinct sample indices of hypothesis h (shared definition with oracle/geom_oracle.py): c = splitmix64 % m, redrawn
// while it repeats an earlier index.  For m <= 65536 the 64-bit remainder is taken exactly with three 32-bit
// multiply-high reductions instead of a software 64-bit division: with M = floor(2^32 / m), q = mulhi(x, M) is
// floor(x / m) or one less, so x - q m needs one conditional subtraction; and x = hi 2^32 + lo gives
// x mod m = ((hi mod m) (2^32 mod m) + lo mod m) mod m with every intermediate below m^2 + m <= 2^32.  Beyond that
// (no frame pair of this path gets there) the remainder is the plain 64-bit one.
__device__ __forceinline__ void sample8(uint64_t seed, int h, int m, int (&idx)[8]) {
    const uint32_t um = (uint32_t)m, q0 = 0xFFFFFFFFu / um, r0 = 0xFFFFFFFFu - q0 * um;
    const uint32_t M = r0 + 1u == um ? q0 + 1u : q0, c32 = r0 + 1u == um ? 0u : r0 + 1u;  // floor(2^32 / m), 2^32 mod m
    const bool small = m <= 65536;
    auto mod32 = [um, M](uint32_t x) {
        uint32_t r = x - __umulhi(x, M) * um;
        return r >= um ? r - um : r;
    };
    uint64_t s = seed + (uint64_t)(h + 1) * 0xD1B54A32D192ED03ull;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int c;
        bool dup;
        do {
            const uint64_t z = splitmix64(s);
            c = small ? (int)mod32(mod32((uint32_t)(z >> 32)) * c32 + mod32((uint32_t)z)) : (int)(z % (uint64_t)um);
            dup = false;
#pragma unroll
            for (int j = 0; j < k; j++) dup |= idx[j] == c;
        } while (dup);
        idx[k] = c;
    }
}

// Null vector of the 8x9 epipolar constraint matrix: Householder QR of its transpose M (9x8, column j = constraint j);
// the last column of Q = H0 H1 .. H7 e8 spans the orthogonal complement of the 8 constraints.  No pivoting is needed
// for an orthonormal Q, every index is a compile-time constant (the whole factorisation lives in registers), and
// it stays well defined when E has vanishing entries (pure sideways translation: e33 = 0).
__device__ bool eight_point(const double* pts /* [8][4] */, double* E) {
    double M[9][8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        double x1 = pts[k * 4], y1 = pts[k * 4 + 1], x2 = pts[k * 4 + 2], y2 = pts[k * 4 + 3];
        M[0][k] = x2 * x1; M[1][k] = x2 * y1; M[2][k] = x2;
        M[3][k] = y2 * x1; M[4][k] = y2 * y1; M[5][k] = y2;
        M[6][k] = x1;      M[7][k] = y1;      M[8][k] = 1.0;
    }
    double beta[8];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        double nrm2 = 0;
#pragma unroll
        for (int i = k; i < 9; i++) nrm2 += M[i][k] * M[i][k];
        double nrm = sqrt(nrm2);
        ok = ok && (nrm > 1e-13);
        double alpha = M[k][k] >= 0 ? -nrm : nrm;
        double v0 = M[k][k] - alpha;           // v = x - alpha e1, stored in place of column k (rows k..8)
        M[k][k] = v0;
        double vtv = nrm2 - 2.0 * alpha * (v0 + alpha) + alpha * alpha;  // = |x|^2 - 2 alpha x0 + alpha^2
        vtv = vtv > 0 ? vtv : 1.0;
        beta[k] = 2.0 / vtv;
#pragma unroll
        for (int j = k + 1; j < 8; j++) {
            double dot = 0;
#pragma unroll
            for (int i = k; i < 9; i++) dot += M[i][k] * M[i][j];
            dot *= beta[k];
#pragma unroll
            for (int i = k; i < 9; i++) M[i][j] -= dot * M[i][k];
        }
    }
    double n[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1.0};
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        double dot = 0;
#pragma unroll
        for (int i = k; i < 9; i++) dot += M[i][k] * n[i];
        dot *= beta[k];
#pragma unroll
        for (int i = k; i < 9; i++) n[i] -= dot * M[i][k];
    }
    double nn = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) nn += n[j] * n[j];
    nn = 1.0 / sqrt(nn);
#pragma unroll
    for (int j = 0; j < 9; j++) E[j] = n[j] * nn;
    return ok;
}

// Sampson threshold in the coordinates the hypotheses live in
__device__ __forceinline__ double tv_thr(const TwoViewArgs& a, const TvWork& w, int pair) {
    return a.model ? a.thr_px * w.norm[(size_t)pair * 8] : a.thr_px / ((a.K[0] + a.K[4]) / 2.0);
}

// nearest rank-2 matrix (smallest singular value -> 0): the fundamental-matrix constraint.  F - sigma3 u3 v3' = F (I - v3 v3')
__device__ bool project_rank2(double* F) {
    double v[3];
    if (!smallest_right3(F, v)) return false;
    for (int i = 0; i < 3; i++) {
        const double fv = F[i * 3] * v[0] + F[i * 3 + 1] * v[1] + F[i * 3 + 2] * v[2];
        for (int j = 0; j < 3; j++) F[i * 3 + j] -= fv * v[j];
    }
    return true;
}

// ---------------------------------------------------------------- prep ----------------------------
// Fundamental-matrix model: replace the K-normalised coordinates by Hartley-normalised pixel coordinates x_n = s (x - c) with
// the centroid of each image and ONE scale for both (mean distance from the centroids -> sqrt 2), so that a Sampson distance
// in these coordinates is s^2 times the distance in pixels and the scoring kernels need no per-image weights.
// Called by all threads of the prep block after px[0 .. m) has been written.
__device__ void tv_hartley(const TwoViewArgs& a, const TvWork& w, int pair, int m) {
    __shared__ double s_h[TV_BLOCK / 64][5];
    __shared__ double s_nrm[5];
    const int tid = threadIdx.x;
    const float* px = w.px + (size_t)pair * a.cap * 4;
    double* xn = w.xn + (size_t)pair * a.cap * 4;
    __syncthreads();  // px of this block is complete
    double acc[4] = {0, 0, 0, 0};
    for (int i = tid; i < m; i += TV_BLOCK)
        for (int k = 0; k < 4; k++) acc[k] += (double)px[4 * i + k];
    for (int k = 0; k < 4; k++) {
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
        if ((tid & 63) == 0) s_h[tid >> 6][k] = acc[k];
    }
    __syncthreads();
    if (tid < 4) s_nrm[1 + tid] = (s_h[0][tid] + s_h[1][tid] + s_h[2][tid] + s_h[3][tid]) / (double)max(m, 1);
    __syncthreads();
    const double c1x = s_nrm[1], c1y = s_nrm[2], c2x = s_nrm[3], c2y = s_nrm[4];
    double dsum = 0;
    for (int i = tid; i < m; i += TV_BLOCK) {
        const double a1 = (double)px[4 * i] - c1x, b1 = (double)px[4 * i + 1] - c1y, a2 = (double)px[4 * i + 2] - c2x, b2 = (double)px[4 * i + 3] - c2y;
        dsum += sqrt(a1 * a1 + b1 * b1) + sqrt(a2 * a2 + b2 * b2);
    }
    for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o, 64);
    if ((tid & 63) == 0) s_h[tid >> 6][4] = dsum;
    __syncthreads();
    if (tid == 0) {
        const double mean = (s_h[0][4] + s_h[1][4] + s_h[2][4] + s_h[3][4]) / (2.0 * (double)max(m, 1));
        s_nrm[0] = mean > 1e-12 ? 1.4142135623730951 / mean : 1.0;
        double* o = w.norm + (size_t)pair * 8;
        o[0] = s_nrm[0]; o[1] = c1x; o[2] = c1y; o[3] = c2x; o[4] = c2y;
    }
    __syncthreads();
    const double sc = s_nrm[0];
    for (int i = tid; i < m; i += TV_BLOCK) {
        xn[4 * i] = sc * ((double)px[4 * i] - c1x); xn[4 * i + 1] = sc * ((double)px[4 * i + 1] - c1y);
        xn[4 * i + 2] = sc * ((double)px[4 * i + 2] - c2x); xn[4 * i + 3] = sc * ((double)px[4 * i + 3] - c2y);
    }
}

// float32 copy of the scoring coordinates and their largest magnitude (first scoring stage); called by all threads of the prep
// block once xn[0 .. m) is final
__device__ void tv_f32_copy(const TwoViewArgs& a, const TvWork& w, int pair, int m) {
    __shared__ float s_mx[TV_BLOCK / 64];
    const int tid = threadIdx.x;
    const double* xn = w.xn + (size_t)pair * a.cap * 4;
    float* xf = w.xf + (size_t)pair * a.cap * 4;
    __syncthreads();  // xn of this block is complete
    float mx = 1.0f;
    for (int i = tid; i < 4 * m; i += TV_BLOCK) {
        const float v = (float)xn[i];
        xf[i] = v;
        mx = fmaxf(mx, fabsf(v));  // (a NaN coordinate is dropped here and poisons nothing: see k_tv_hyp)
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) s_mx[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) w.norm[(size_t)pair * 8 + 5] = (double)fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3])) * 1.0000002;
}

__global__ __launch_bounds__(TV_BLOCK) void k_tv_prep(TwoViewArgs a, TvWork w) {
    const int pair = blockIdx.x, tid = threadIdx.x;
    __shared__ int s_w[TV_BLOCK / 64];
    __shared__ int s_base;
    const double fx = a.K[0], fy = a.K[4], cx = a.K[2], cy = a.K[5];
    double* xn = w.xn + (size_t)pair * a.cap * 4;
    float* px = w.px + (size_t)pair * a.cap * 4;
    int* qidx = w.qidx + (size_t)pair * a.cap;
    if (tid == 0) {
        s_base = 0; w.best[pair] = ~0ull; w.n_alive[pair] = 0; if (pair == 0) *w.n_tasks = 0;
        *(unsigned*)&w.norm[(size_t)pair * 8 + 6] = 0x7F800000u;  // the pair-wide bound of k_tv_hyp's first stage: none yet
    }
    for (int j = tid; j < (a.n_hyp + 63) / 64; j += TV_BLOCK) w.wkey[(size_t)pair * ((a.n_hyp + 63) / 64) + j] = ~0ull;
    __syncthreads();
    if (a.d_p1) {  // explicit correspondences
        int m = a.m_fixed;
        for (int i = tid; i < m; i += TV_BLOCK) {
            float u1 = a.d_p1[(size_t)pair * m * 2 + 2 * i], v1 = a.d_p1[(size_t)pair * m * 2 + 2 * i + 1];
            float u2 = a.d_p2[(size_t)pair * m * 2 + 2 * i], v2 = a.d_p2[(size_t)pair * m * 2 + 2 * i + 1];
            px[4 * i] = u1; px[4 * i + 1] = v1; px[4 * i + 2] = u2; px[4 * i + 3] = v2;
            xn[4 * i] = ((double)u1 - cx) / fx; xn[4 * i + 1] = ((double)v1 - cy) / fy;
            xn[4 * i + 2] = ((double)u2 - cx) / fx; xn[4 * i + 3] = ((double)v2 - cy) / fy;
            qidx[i] = i;
        }
        if (tid == 0) w.m[pair] = m;
        if (a.model) tv_hartley(a, w, pair, m);
        tv_f32_copy(a, w, pair, m);
        return;
    }
    // pair p = frame p (query) vs frame p + 1 (train), or the caller's (query frame, train frame) arrays (keyframe mode)
    const int qfr = a.d_qf ? a.d_qf[pair] : pair, tfr = a.d_tf ? a.d_tf[pair] : pair + 1;
    const mo_keypoint* k1 = a.d_kps + (size_t)qfr * a.cap;
    const mo_keypoint* k2 = a.d_kps + (size_t)tfr * a.cap;
    if (a.d_sel) {  // tracking mode: the filtered match list in the reference's order (track_kernels.hip)
        const int m = min(a.d_sel_n[pair], a.cap);
        const int32_t* sl = a.d_sel + (size_t)pair * a.cap * 2;
        for (int o = tid; o < m; o += TV_BLOCK) {
            const int i = sl[2 * o], j = sl[2 * o + 1];
            float u1 = k1[i].x, v1 = k1[i].y, u2 = k2[j].x, v2 = k2[j].y;
            px[4 * o] = u1; px[4 * o + 1] = v1; px[4 * o + 2] = u2; px[4 * o + 3] = v2;
            xn[4 * o] = ((double)u1 - cx) / fx; xn[4 * o + 1] = ((double)v1 - cy) / fy;
            xn[4 * o + 2] = ((double)u2 - cx) / fx; xn[4 * o + 3] = ((double)v2 - cy) / fy;
            qidx[o] = i;
        }
        if (tid == 0) w.m[pair] = m;
        if (a.model) tv_hartley(a, w, pair, m);
        tv_f32_copy(a, w, pair, m);
        return;
    }
    // from matcher output: survivors of the ratio test in query order (need_two: only queries with a second neighbour,
    // local_mapper.py:123 `if len(match_pair) >= 2`)
    const int nq = min(a.d_counts[qfr], a.cap);
    const int32_t* midx = a.d_match_idx + (size_t)pair * a.cap * 2;
    const uint8_t* pass = a.d_match_pass + (size_t)pair * a.cap;
    for (int base = 0; base < nq; base += TV_BLOCK) {
        int i = base + tid;
        bool ok = i < nq && pass[i] && (!a.need_two || midx[2 * i + 1] >= 0);
        unsigned long long bal = __ballot(ok);
        int lane = tid & 63, wv = tid >> 6;
        int wpre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wv] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int k = 0; k < wv; k++) off += s_w[k];
        if (ok) {
            int o = off + wpre;
            int j = midx[2 * i];
            float u1 = k1[i].x, v1 = k1[i].y, u2 = k2[j].x, v2 = k2[j].y;
            px[4 * o] = u1; px[4 * o + 1] = v1; px[4 * o + 2] = u2; px[4 * o + 3] = v2;
            xn[4 * o] = ((double)u1 - cx) / fx; xn[4 * o + 1] = ((double)v1 - cy) / fy;
            xn[4 * o + 2] = ((double)u2 - cx) / fx; xn[4 * o + 3] = ((double)v2 - cy) / fy;
            qidx[o] = i;
        }
        __syncthreads();
        if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    const int m = s_base;
    if (tid == 0) w.m[pair] = m;
    if (a.model) tv_hartley(a, w, pair, m);
    tv_f32_copy(a, w, pair, m);
}

// ---------------------------------------------------------------- hypotheses ----------------------

// correspondences summed by the first scoring stage: TV_FIRST_NUM eighths of them, at least 32
#ifndef TV_FIRST_NUM
#define TV_FIRST_NUM 2
#endif
__device__ __forceinline__ int tv_first(int m, int num) { return min(m, max(32, (m * num + 7) >> 3)); }

// ---- THE CANONICAL MSAC COST (the value the keys are defined on): the correspondences are cut into TV_PARTS contiguous parts of
// q = ceil(m / TV_PARTS); each part is summed sequentially in fp64, the parts are combined as (P0 + P1) + (P2 + P3).  One lane
// can do all of it (tv_cost_lane, few hypotheses), or the four wavefronts of a workgroup one part each (k_tv_score: four times the wavefronts for
// the same work - the complete sums are chains of dependent fp64 additions and the launch is otherwise short of wavefronts).
__device__ __forceinline__ int tv_part_len(int m) { return (m + TV_PARTS - 1) / TV_PARTS; }
__device__ __forceinline__ double tv_cost_part(const double* E, const double* xn, int i0, int i1, double thr2) {
    typedef const __attribute__((address_space(4))) double* cdp;  // wave-uniform scalar loads: the coordinates feed the FMAs as scalar operands
    const cdp pts = (cdp)(uintptr_t)xn;
    double s = 0.0;
#pragma unroll 4
    for (int i = i0; i < i1; i++) s += fmin(sampson_fast(E, pts[4 * i], pts[4 * i + 1], pts[4 * i + 2], pts[4 * i + 3]), thr2);
    return s;
}
__device__ __forceinline__ double tv_combine(const double (&P)[TV_PARTS]) { return (P[0] + P[1]) + (P[2] + P[3]); }
__device__ __forceinline__ double tv_cost_lane(const double* E, const double* xn, int m, double thr2) {
    const int q = tv_part_len(m);
    double P[TV_PARTS];
    for (int k = 0; k < TV_PARTS; k++) P[k] = tv_cost_part(E, xn, min(m, k * q), min(m, (k + 1) * q), thr2);
    return tv_combine(P);
}

__global__ __launch_bounds__(TV_BLOCK) void k_tv_hyp(TwoViewArgs a, TvWork w, int staged, int first_num) {
    __shared__ unsigned long long s_best[TV_BLOCK / 64];
    // (the blocks of a pair are dispatched side by side; a pair-major grid - a pair's blocks spread over the launch's lifetime so that the
    //  later ones prune with the earlier ones' bounds - measured slower: 0.397 against 0.385 ms, profiles/r04_ab_tv_bound.txt)
    const int pair = blockIdx.y, hb = blockIdx.x, tid = threadIdx.x;
    const int h = hb * TV_BLOCK + tid;
    const int m = w.m[pair];
    if (m < 8) return;
    const double* xn = w.xn + (size_t)pair * a.cap * 4;
    const double thr = tv_thr(a, w, pair);
    const double thr2 = thr * thr;
    double E[9];
    bool valid = h < a.n_hyp;
    if (valid) {
        int idx[8];
        sample8(a.seed + ((uint64_t)pair + a.pair_base) * 0x632BE59BD9B4E019ull, h, m, idx);  // GLOBAL pair index: results do not depend on the sharding
        double pts[32];
        for (int k = 0; k < 8; k++)
            for (int j = 0; j < 4; j++) pts[k * 4 + j] = xn[(size_t)idx[k] * 4 + j];
        valid = eight_point(pts, E) && (a.model ? project_rank2(E) : project_essential(E));
    }
    if (!valid) for (int j = 0; j < 9; j++) E[j] = 0.0;
    // MSAC score: sum of Sampson distances truncated at thr^2 (a pure inlier count prefers slightly perturbed
    // models that catch more chance inliers).  Compared as float32, ties -> lowest hypothesis index.
    // The correspondences are the same for every lane: they arrive through wave-uniform scalar loads (constant address
    // space; k_tv_prep wrote them in an earlier launch) and feed the FMAs as scalar operands, one per instruction.
    const int lane = tid & 63, wv = tid >> 6;
    const size_t ntask_max = (size_t)((a.n_hyp + 63) / 64);
    unsigned long long key = ~0ull;
    if (!staged) {  // few hypotheses: every one is scored completely here, the block's best goes straight to the pair's minimum
        const double cost = valid ? tv_cost_lane(E, xn, m, thr2) : 0.0;
        if (valid && cost == cost) key = ((unsigned long long)__float_as_uint((float)cost) << 32) | (unsigned long long)(unsigned)h;
        const unsigned long long own = key;
        for (int o = 32; o > 0; o >>= 1) {
            unsigned long long other = __shfl_xor(key, o, 64);
            key = other < key ? other : key;
        }
        if (lane == 0) s_best[wv] = key;
        __syncthreads();
        for (int k = 0; k < TV_BLOCK / 64; k++) key = s_best[k] < key ? s_best[k] : key;  // block minimum, in every thread
        if (key != ~0ull && own == key) {  // (a block index is a valid slot: there are at least as many 64-tasks as 256-blocks)
            double* bE = w.wE + ((size_t)pair * ntask_max + hb) * 9;
            for (int j = 0; j < 9; j++) bE[j] = E[j];
            w.wkey[(size_t)pair * ntask_max + hb] = key;
            atomicMin(&w.best[pair], key);
        }
        return;
    }
    // Staged scoring (exact).  Stage 1, here, in float32: a LOWER BOUND of the cost over the first F correspondences.
    // With eps = 2^-24, X = the largest |coordinate| (>= 1, from k_tv_prep) and s1 = sum |E_ij|, float32 evaluation of
    // (E x1)_k is off by at most a = 4e-7 X s1 (five roundings of terms <= X s1, inputs rounded once), the residual
    // r = x2' E x1 by at most delta = 2e-6 X^2 s1, and the true denominator is at most (sqrt(den) + 2a)^2 <=
    // 1.001 den + 4004 a^2.  So  max(|r| - delta, 0)^2 / (1.0011 den + 4004 a^2)  never exceeds the exact Sampson
    // distance (the extra 1e-4 covers v_rcp_f32 and the roundings of this line), the truncated sum of F of them times
    // (1 - 1e-4) (float32 accumulation of non-negative terms) never exceeds the exact partial cost, and costs only grow
    // with more correspondences.  The block's hypothesis with the smallest bound is scored completely, in fp64, by the
    // whole block: its total (rounded up) is an upper bound of the best total of the pair, and a hypothesis whose lower
    // bound exceeds it cannot win.  Only the others - typically 5-20 % - are scored in fp64, by k_tv_score.
    typedef const __attribute__((address_space(4))) float* cfp;
    const cfp pf = (cfp)(uintptr_t)(w.xf + (size_t)pair * a.cap * 4);
    const int F = tv_first(m, first_num);
    const float X = (float)w.norm[(size_t)pair * 8 + 5];
    float lb = 0.0f;
    if (valid && X < 1e6f) {  // (absurd coordinates: no float32 pruning, every hypothesis goes to the fp64 stage)
        float e[9], s1 = 0.0f;
        for (int j = 0; j < 9; j++) { e[j] = (float)E[j]; s1 += fabsf(e[j]); }
        s1 *= 1.000001f;
        const float ae = 4e-7f * X * s1, delta = 2e-6f * X * X * s1, cden = 4004.0f * ae * ae;
        const float thr2f = __uint_as_float(__float_as_uint((float)thr2) - 1u);  // below thr^2 (thr2 > 0: a normal number)
#pragma unroll 4
        for (int i = 0; i < F; i++) {
            const float x1 = pf[4 * i], y1 = pf[4 * i + 1], x2 = pf[4 * i + 2], y2 = pf[4 * i + 3];
            const float ex0 = fmaf(e[0], x1, fmaf(e[1], y1, e[2]));
            const float ex1 = fmaf(e[3], x1, fmaf(e[4], y1, e[5]));
            const float ex2 = fmaf(e[6], x1, fmaf(e[7], y1, e[8]));
            const float et0 = fmaf(e[0], x2, fmaf(e[3], y2, e[6]));
            const float et1 = fmaf(e[1], x2, fmaf(e[4], y2, e[7]));
            const float r = fmaf(x2, ex0, fmaf(y2, ex1, ex2));
            const float den = fmaf(ex0, ex0, fmaf(ex1, ex1, fmaf(et0, et0, et1 * et1)));
            const float rl = fmaxf(fabsf(r) - delta, 0.0f);
            const float t = rl * rl * __builtin_amdgcn_rcpf(fmaf(den, 1.0011f, cden));
            lb += fminf(t, thr2f);
        }
        lb *= 0.9999f;
        if (!(lb >= 0.0f)) lb = 0.0f;  // NaN (overflow somewhere): no information, keep the hypothesis
    }
    if (valid) key = ((unsigned long long)__float_as_uint(lb) << 32) | (unsigned long long)(unsigned)h;
    const unsigned long long own = key;
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long other = __shfl_xor(key, o, 64);
        key = other < key ? other : key;
    }
    if (lane == 0) s_best[wv] = key;
    __syncthreads();
    for (int k = 0; k < TV_BLOCK / 64; k++) key = s_best[k] < key ? s_best[k] : key;  // block minimum, in every thread
    if (key == ~0ull) return;  // no valid hypothesis in this block (block-uniform)
    // the block's candidate: total cost by all threads -> bound
    __shared__ double s_cand[9];
    __shared__ double s_sum[TV_BLOCK / 64];
    __shared__ unsigned s_bound;
    if (own == key)
        for (int j = 0; j < 9; j++) s_cand[j] = E[j];
    __syncthreads();
    {
        double Ec[9];
        for (int j = 0; j < 9; j++) Ec[j] = s_cand[j];
        double tot = 0.0;
        for (int i = tid; i < m; i += TV_BLOCK)
            tot += fmin(sampson_fast(Ec, xn[4 * i], xn[4 * i + 1], xn[4 * i + 2], xn[4 * i + 3]), thr2);
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
        if (lane == 0) s_sum[wv] = tot;
    }
    __syncthreads();
    if (tid == 0) {
        double total = 0.0;
        for (int k = 0; k < TV_BLOCK / 64; k++) total += s_sum[k];
        // this sum is associated differently from the canonical (sequential) one: two float32 ulps upwards cover that
        // and the rounding to float32, so the bound never undercuts the candidate's canonical cost.  A NaN total (the
        // canonical sum of the candidate is then NaN too and it cannot win) gives no bound: everything survives.
        unsigned bound = total == total ? __float_as_uint((float)total) + 2u : 0x7F800000u;
        // the bound is pair-wide: every block posts its candidate's total and prunes with the smallest one posted so far (whichever
        // blocks of the pair got there first - the set of survivors varies from run to run, the minimum over them does not: the best
        // hypothesis of the pair lies below every candidate's total and always survives).  Non-negative floats order like their bits.
        const unsigned seen = atomicMin((unsigned*)&w.norm[(size_t)pair * 8 + 6], bound);
        s_bound = min(bound, seen);
    }
    __syncthreads();
    // survivors (lower bound <= bound; both >= 0, so bit order = value order) -> the pair's dense list in HBM (order is
    // irrelevant for a minimum; the candidate itself always survives): one atomic per block
    __shared__ int s_wcnt[TV_BLOCK / 64];
    __shared__ int s_gbase;
    const bool alive = own != ~0ull && __float_as_uint(lb) <= s_bound;
    const unsigned long long mk = __ballot(alive);
    if (lane == 0) s_wcnt[wv] = __popcll(mk);
    __syncthreads();
    int sbase = 0, n_alive = 0;
    for (int k = 0; k < TV_BLOCK / 64; k++) { if (k < wv) sbase += s_wcnt[k]; n_alive += s_wcnt[k]; }
    if (tid == 0) s_gbase = atomicAdd(&w.n_alive[pair], n_alive);
    __syncthreads();
    if (alive) {
        double* rec = w.surv + ((size_t)pair * a.n_hyp + s_gbase + sbase + __popcll(mk & ((1ull << lane) - 1ull))) * TV_REC;
        for (int j = 0; j < 9; j++) rec[j] = E[j];
        rec[9] = __longlong_as_double((long long)h);
    }
}

// One task per 64 survivors of a pair, appended to a launch-wide dense table: k_tv_score's workgroup t takes task t, so the busy
// wavefronts are the FIRST ones of its grid and spread evenly over the chip (the empty tail exits at once).
__global__ __launch_bounds__(TV_BLOCK) void k_tv_tasks(TwoViewArgs a, TvWork w) {
    __shared__ int s_off[TV_BLOCK + 1];
    const int tid = threadIdx.x;
    int total = 0;
    for (int p0 = 0; p0 < a.n_pairs; p0 += TV_BLOCK) {  // block-uniform trip count
        const int p = p0 + tid;
        const int nt = p < a.n_pairs ? (w.n_alive[p] + 63) / 64 : 0;
        s_off[tid + 1] = nt;
        __syncthreads();
        if (tid == 0) {
            s_off[0] = total;
            for (int k = 0; k < TV_BLOCK; k++) s_off[k + 1] += s_off[k];
        }
        __syncthreads();
        for (int j = 0; j < nt; j++) w.task[s_off[tid] + j] = make_int2(p, j);
        total = s_off[TV_BLOCK];
        __syncthreads();
    }
    if (tid == 0) *w.n_tasks = total;
}

// Second stage: the canonical cost of every survivor.  One workgroup per task of 64 survivors (one per lane), wavefront k sums part k;
// the parts meet in LDS, wavefront 0 combines them and its best key goes to the pair's minimum and, with its matrix, to the slot
// k_tv_finish looks it up in.  (Four wavefronts of different workgroups with a device-scope fence and a counter instead: 2.4x slower
// - a release fence writes the L2 of the XCD back.)
__global__ __launch_bounds__(64 * TV_PARTS) void k_tv_score(TwoViewArgs a, TvWork w, int one_pair) {
    __shared__ double s_part[TV_PARTS][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    // one pair (the single-frame calls): workgroup t IS task t of pair 0, no task table (k_tv_tasks is not launched)
    if (one_pair ? (int)blockIdx.x * 64 >= w.n_alive[0] : (int)blockIdx.x >= *w.n_tasks) return;  // block-uniform
    const int2 tk = one_pair ? make_int2(0, (int)blockIdx.x) : w.task[blockIdx.x];
    const int pair = tk.x, t0 = tk.y * 64;
    const int total = w.n_alive[pair];
    const int m = w.m[pair];
    const double thr = tv_thr(a, w, pair);
    const double thr2 = thr * thr;
    const double* xn = w.xn + (size_t)pair * a.cap * 4;
    const bool on = t0 + lane < total;
    const double* rec = w.surv + ((size_t)pair * a.n_hyp + (on ? t0 + lane : t0)) * TV_REC;
    double E[9];
    for (int j = 0; j < 9; j++) E[j] = rec[j];
    const int q = tv_part_len(m);
    s_part[part][lane] = tv_cost_part(E, xn, min(m, part * q), min(m, (part + 1) * q), thr2);
    __syncthreads();
    if (part != 0) return;
    double P[TV_PARTS];
    for (int k = 0; k < TV_PARTS; k++) P[k] = s_part[k][lane];
    const double cost = tv_combine(P);
    const unsigned h = (unsigned)__double_as_longlong(rec[9]);
    unsigned long long key = ~0ull;
    if (on && cost == cost) key = ((unsigned long long)__float_as_uint((float)cost) << 32) | (unsigned long long)h;
    const unsigned long long own = key;
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long other = __shfl_xor(key, o, 64);
        key = other < key ? other : key;
    }
    if (key != ~0ull && own == key) {  // exactly one lane (keys are unique)
        const size_t slot = (size_t)pair * ((a.n_hyp + 63) / 64) + tk.y;
        for (int j = 0; j < 9; j++) w.wE[slot * 9 + j] = E[j];
        w.wkey[slot] = key;
        atomicMin(&w.best[pair], key);
    }
}

// ---------------------------------------------------------------- finish --------------------------
// DLT null vector of the 4x4 system [x*P3-P1; y*P3-P2] for two views: smallest eigenvector of A^T A by inverse
// iteration (the matrix is rank 3 up to noise, so two or three steps reach machine precision)
__device__ void dlt_point(const double* P1, const double* P2, double x1, double y1, double x2, double y2, double* X) {
    double A[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = x1 * P1[8 + k] - P1[k];
        A[1 * 4 + k] = y1 * P1[8 + k] - P1[4 + k];
        A[2 * 4 + k] = x2 * P2[8 + k] - P2[k];
        A[3 * 4 + k] = y2 * P2[8 + k] - P2[4 + k];
    }
    double S[10];
#pragma unroll
    for (int i = 0, k = 0; i < 4; i++)
#pragma unroll
        for (int j = i; j < 4; j++, k++) S[k] = A[i] * A[j] + A[4 + i] * A[4 + j] + A[8 + i] * A[8 + j] + A[12 + i] * A[12 + j];
    X[0] = 0.5; X[1] = 0.5; X[2] = 0.5; X[3] = 0.5;
    smallest_eigvec<4>(S, X, 8);
}

__device__ inline double block_sum(double v, double* s_red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int k = 0; k < TVF_BLOCK / 64; k++) r += s_red[k];
    return r;
}

__device__ inline int block_sum_i(int v, int* s_red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = 0;
    for (int k = 0; k < TVF_BLOCK / 64; k++) r += s_red[k];
    return r;
}

// Smallest eigenvector of the 9x9 normal matrix of a refit on ONE wavefront (rounds 1 - 3: thread 0 alone, LDL^T + at most 16 steps of
// inverse iteration - 16 k cycles per call, 46 % of this kernel, and with an eigenvalue ratio of 0.5 between the two smallest
// eigenvalues 16 steps leave an error of 1.5e-5: the iteration ran into its cap on every pair of the benchmark).  Here the matrix is
// inverted explicitly and the inverse is squared ten times: (M^-1)^(2^10) has the wanted vector as its dominant eigenvector with
// the eigenvalue ratio raised to the 1024th power (0.99 -> 3e-5, 0.9 -> 1e-47).  Both steps work on the 45 entries of the packed upper
// triangle, one per lane:
//   * inverse by the symmetric sweep operator (sweep k: M[k][k] -> -1/d, M[i][k] -> M[i][k]/d, M[i][j] -> M[i][j] - M[i][k] M[k][j]/d;
//     all nine sweeps give -M^-1 and every intermediate matrix is symmetric), on M / trace + 1e-15 I, pivots clamped at 1e-17;
//   * B <- (B / trace B)^2, nine products per lane out of LDS.
// sN: the 45 packed sums (p <= q, row-major) in LDS; sM: 48 doubles of LDS scratch; x[9] in LDS: previous estimate in (its orientation
// is kept), eigenvector out.  Lanes 0..63 of one wavefront, convergent; returns false (x untouched) when the arithmetic broke down.
__device__ __forceinline__ int tv_pk(int p, int q) { return p <= q ? p * 9 - (p * (p - 1)) / 2 + (q - p) : q * 9 - (q * (q - 1)) / 2 + (p - q); }
__device__ __forceinline__ void tv_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ bool wave_smallest_eigvec9(const double* sN, double* sM, double* x, int lane) {
    int p = 0, q = lane < 45 ? lane : 0;
    while (q >= 9 - p) { q -= 9 - p; p++; }  // packed index -> (p, p + q)
    q += p;
    const bool on = lane < 45;
    double tr = 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) tr += sN[tv_pk(k, k)];
    if (!(tr > 0.0)) return false;  // (wave-uniform: every lane read the same nine values)
    const double itr = 1.0 / tr;
    if (on) sM[lane] = sN[lane] * itr + (p == q ? 1e-15 : 0.0);
    tv_wave_sync();
    for (int k = 0; k < 9; k++) {  // sweep k
        const double d = fmax(sM[tv_pk(k, k)], 1e-17), id = 1.0 / d;
        const double a = sM[tv_pk(p, k)], b = sM[tv_pk(k, q)], c = sM[lane < 45 ? lane : 0];
        const double nv = p == k && q == k ? -id : p == k ? b * id : q == k ? a * id : c - a * b * id;
        tv_wave_sync();
        if (on) sM[lane] = nv;
        tv_wave_sync();
    }
    // sM = -(M^-1); its square is the square of M^-1
    for (int it = 0; it < 10; it++) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) t += sM[tv_pk(k, k)];
        const double sc = 1.0 / t;  // (trace of a definite matrix: never 0; negative in the first round, squared away)
        // After a round the matrix is (previous / trace)^2: its trace is the sum of the squared normalised eigenvalues, 1 - 2 mu2 / mu1 to first
        // order.  Within 1e-13 of 1 the second direction's share of every column is below 5e-14 and further rounds change nothing that the
        // 1e-4 comparison with the oracle - or this scheme's own 1e-13 distance from eigh - could see: typical normal matrices get there
        // in 3 - 5 rounds, ill-separated ones still take all ten (wave-uniform: every lane read the same nine values).
        // (A/B against the fixed ten rounds: two-view stage 0.385 - 0.389 against 0.387 - 0.392 ms, profiles/r04_ab_tv_bound.txt)
        if (it > 0 && fabs(1.0 - t) < 1e-13) break;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) acc += sM[tv_pk(p, k)] * sM[tv_pk(k, q)];
        acc *= sc * sc;
        tv_wave_sync();
        if (on) sM[lane] = acc;
        tv_wave_sync();
    }
    // the column with the largest diagonal entry, normalised, oriented like the previous estimate
    int jb = 0;
    double best = sM[tv_pk(0, 0)];
#pragma unroll
    for (int k = 1; k < 9; k++) { const double v = sM[tv_pk(k, k)]; if (v > best) { best = v; jb = k; } }
    double nn = 0.0, dot = 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) { const double v = sM[tv_pk(k, jb)]; nn += v * v; dot += v * x[k]; }
    if (!(nn > 0.0) || !(nn < 1e300)) return false;  // NaN / overflow somewhere (wave-uniform)
    const double sgn = (dot < 0 ? -1.0 : 1.0) / sqrt(nn);
    const double mine = lane < 9 ? sM[tv_pk(lane, jb)] * sgn : 0.0;
    tv_wave_sync();
    if (lane < 9) x[lane] = mine;
    tv_wave_sync();
    return true;
}

__global__ __launch_bounds__(TVF_BLOCK) void k_tv_finish(TwoViewArgs a, TvWork w) {
    __shared__ double s_red[TVF_BLOCK / 64];
    __shared__ int s_redi[TVF_BLOCK / 64];
    __shared__ double s_E[9];
    __shared__ double s_P[4][12];   // candidate [R|t] in normalised coordinates
    __shared__ double s_Ppix[2][12];
    __shared__ int s_ok, s_win;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int m = w.m[pair];
    double* pose = a.d_pose ? a.d_pose + (size_t)pair * 12 : nullptr;
    float* Xout = a.d_points + (size_t)pair * a.cap * 3;
    uint8_t* inl_out = a.d_inlier ? a.d_inlier + (size_t)pair * a.cap : nullptr;
    uint8_t* ran_out = a.d_ransac ? a.d_ransac + (size_t)pair * a.cap : nullptr;
    const int out_n = a.d_p1 ? a.m_fixed : a.cap;
    const float qnan = __uint_as_float(0x7FC00000u);
    for (int i = tid; i < out_n; i += TVF_BLOCK) {
        Xout[3 * i] = qnan; Xout[3 * i + 1] = qnan; Xout[3 * i + 2] = qnan;
        if (inl_out) inl_out[i] = 0;
        if (ran_out) ran_out[i] = 0;
    }
    const bool given = a.d_E_in != nullptr;  // recoverPose mode: E and the consensus mask come from the caller
    const unsigned long long best = given ? 0ull : w.best[pair];
    if (given ? m < 1 : (m < 8 || best == ~0ull)) {
        if (tid == 0) {
            a.d_n_points[pair] = 0;
            if (pose) for (int j = 0; j < 12; j++) pose[j] = __longlong_as_double(0x7FF8000000000000ll);
            if (a.d_E) for (int j = 0; j < 9; j++) a.d_E[(size_t)pair * 9 + j] = __longlong_as_double(0x7FF8000000000000ll);
        }
        return;
    }
    const double* xn = w.xn + (size_t)pair * a.cap * 4;
    const float* px = w.px + (size_t)pair * a.cap * 4;
    const int* qidx = w.qidx + (size_t)pair * a.cap;
    const double thr = tv_thr(a, w, pair);
    const double thr2 = thr * thr;
    if (given) {
        if (tid < 9) s_E[tid] = a.d_E_in[(size_t)pair * 9 + tid];
    } else {  // the matrix of the winning key: every scoring wavefront (or unstaged block) left its best in a slot of the pair
        const int nslot = (a.n_hyp + 63) / 64;
        for (int j = tid; j < nslot; j += TVF_BLOCK)
            if (w.wkey[(size_t)pair * nslot + j] == best)
                for (int q = 0; q < 9; q++) s_E[q] = w.wE[((size_t)pair * nslot + j) * 9 + q];
    }
    __syncthreads();

    // ---- local optimisation: least-squares 8-point refits (9x9 normal matrix, 45 unique sums, Jacobi) on an
    // adaptively tightened consensus set.  The selection threshold follows a 3-sigma rule on the mean Sampson
    // residual of the previous selection, clamped to [thr/64, thr], so chance inliers of the loose RANSAC threshold
    // do not bias the algebraic fit; a refit is only accepted while >= half of the original consensus is selected.
    __shared__ double s_N[45], s_M[48], s_x[9];
    __shared__ double s_R[TVF_BLOCK * 9];          // constraint rows of one pass
    __shared__ double s_tile[TVF_BLOCK / 64][256];  // per-wavefront partial Gram tiles
    __shared__ int s_stop;
    if (!given) {  // block-uniform
    const double lo2 = thr2 / 4096.0;
    int n0;
    double tau2;
    {
        double E[9];
        for (int j = 0; j < 9; j++) E[j] = s_E[j];
        int cnt = 0;
        double sd = 0;
        for (int i = tid; i < m; i += TVF_BLOCK) {
            double d = sampson(E, xn[4 * i], xn[4 * i + 1], xn[4 * i + 2], xn[4 * i + 3]);
            if (d <= thr2) { cnt++; sd += d; }
        }
        n0 = block_sum_i(cnt, s_redi);
        double sds = block_sum(sd, s_red);
        tau2 = n0 > 0 ? fmin(fmax(9.0 * sds / n0, lo2), thr2) : thr2;
    }
    if (n0 < 8) {  // block-uniform: the best model does not even explain a minimal sample
        if (tid == 0) {
            a.d_n_points[pair] = 0;
            if (pose) for (int j = 0; j < 12; j++) pose[j] = __longlong_as_double(0x7FF8000000000000ll);
            if (a.d_E) for (int j = 0; j < 9; j++) a.d_E[(size_t)pair * 9 + j] = __longlong_as_double(0x7FF8000000000000ll);
        }
        return;
    }
    int c_prev = -1;
    double tau2_prev = -1.0;
    const int lane = tid & 63, wv = tid >> 6, gc = lane & 15, gk = lane >> 4;
    for (int it = 0; it < 5; it++) {
        // Normal matrix N = R^T R of the selected constraint rows r = [x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1]
        // as a Gram product on the fp64 matrix core: v_mfma_f64_16x16x4_f64 takes A[i][k] and B[k][j] from lane
        // (i or j = lane & 15, k = lane >> 4); with A = R^T and B = R both operands are the same register, R[k][lane & 15].
        // Each thread stages the row of its correspondence in LDS (zeros when it is not selected), each wavefront
        // multiplies its 64 rows in 16 steps; the cross-lane sums happen inside the MFMA.
        v4d gram = {0.0, 0.0, 0.0, 0.0};
        int cnt = 0;
        double sd = 0;
        {
            double E[9];
            for (int j = 0; j < 9; j++) E[j] = s_E[j];
            for (int base = 0; base < m; base += TVF_BLOCK) {  // block-uniform trip count
                const int i = base + tid;
                double r[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (i < m) {
                    double x1 = xn[4 * i], y1 = xn[4 * i + 1], x2 = xn[4 * i + 2], y2 = xn[4 * i + 3];
                    double d = sampson(E, x1, y1, x2, y2);
                    if (d <= tau2) {
                        r[0] = x2 * x1; r[1] = x2 * y1; r[2] = x2; r[3] = y2 * x1; r[4] = y2 * y1; r[5] = y2;
                        r[6] = x1; r[7] = y1; r[8] = 1.0;
                        cnt++;
                        sd += d;
                    }
                }
                __syncthreads();  // the previous pass has been consumed
#pragma unroll
                for (int j = 0; j < 9; j++) s_R[tid * 9 + j] = r[j];
                __syncthreads();
#pragma unroll 4
                for (int s4 = 0; s4 < 16; s4++) {
                    const double v = gc < 9 ? s_R[(wv * 64 + 4 * s4 + gk) * 9 + gc] : 0.0;
                    gram = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, gram, 0, 0, 0);
                }
            }
        }
        const int c = block_sum_i(cnt, s_redi);
        if (c < 8 || 2 * c < n0) break;  // block-uniform
        if (c == c_prev && tau2 == tau2_prev) break;  // same selection size at the same threshold: converged
        c_prev = c; tau2_prev = tau2;
        const double sds = block_sum(sd, s_red);
        // per-wavefront 16x16 partial Gram tiles (result row = (lane >> 4) + 4 reg, column = lane & 15) -> 45 sums
#pragma unroll
        for (int rg = 0; rg < 4; rg++) s_tile[wv][(gk + 4 * rg) * 16 + gc] = gram[rg];
        __syncthreads();
        if (tid < 45) {
            int p = 0, q = tid;
            while (q >= 9 - p) { q -= 9 - p; p++; }  // packed upper-triangle index -> (p, p + q)
            double v = 0;
            for (int k = 0; k < TVF_BLOCK / 64; k++) v += s_tile[k][p * 16 + p + q];
            s_N[tid] = v;
        }
        __syncthreads();
        // smallest eigenvector of the normal matrix on wavefront 0 (the current estimate only fixes the sign), then the projection on
        // the essential / rank-2 manifold by one lane
        if (tid < 64) {
            if (tid < 9) {
                double en = 0;
                for (int i = 0; i < 9; i++) en += s_E[i] * s_E[i];
                s_x[tid] = s_E[tid] / sqrt(en);
            }
            tv_wave_sync();
            const bool conv = wave_smallest_eigvec9(s_N, s_M, s_x, tid);
            if (tid == 0) {
                double E[9];
                for (int i = 0; i < 9; i++) E[i] = s_x[i];
                const bool ok = conv && (a.model ? project_rank2(E) : project_essential(E));
                if (ok) for (int i = 0; i < 9; i++) s_E[i] = E[i];  // else keep the previous estimate
                s_stop = ok ? 0 : 1;
            }
        }
        __syncthreads();
        if (s_stop) break;
        tau2 = fmin(fmax(9.0 * sds / c, lo2), thr2);
    }
    }
    __syncthreads();
    if (a.model) {  // fundamental matrix: mask = Sampson distance within thr (pixels), F back in pixel coordinates; no pose
        double Fn[9];
        for (int j = 0; j < 9; j++) Fn[j] = s_E[j];
        int cnt = 0;
        for (int i = tid; i < m; i += TVF_BLOCK) {
            const bool in = sampson(Fn, xn[4 * i], xn[4 * i + 1], xn[4 * i + 2], xn[4 * i + 3]) <= thr2;
            if (in) {
                cnt++;
                if (ran_out) ran_out[qidx[i]] = 1;
                if (inl_out) inl_out[qidx[i]] = 1;
                if (a.d_P1) {  // keyframe map growth (local_mapper.py:148-149): triangulate the inliers with the caller's two projection matrices
                    double Pa[12], Pb[12], X[4];
                    for (int j = 0; j < 12; j++) { Pa[j] = a.d_P1[(size_t)pair * 12 + j]; Pb[j] = a.d_P2[(size_t)pair * 12 + j]; }
                    dlt_point(Pa, Pb, (double)px[4 * i], (double)px[4 * i + 1], (double)px[4 * i + 2], (double)px[4 * i + 3], X);
                    const float xf = (float)X[0], yf = (float)X[1], zf = (float)X[2], wf = (float)X[3];
                    const int o = qidx[i];
                    Xout[3 * o] = xf / wf; Xout[3 * o + 1] = yf / wf; Xout[3 * o + 2] = zf / wf;
                }
            }
        }
        const int total = block_sum_i(cnt, s_redi);
        if (tid == 0) {
            a.d_n_points[pair] = total;
            if (pose) for (int j = 0; j < 12; j++) pose[j] = __longlong_as_double(0x7FF8000000000000ll);
            if (a.d_E) {  // F = T2^T Fn T1 with T = [s 0 -s cx; 0 s -s cy; 0 0 1], scaled to F33 = 1 like cv2 (unit norm if F33 ~ 0)
                const double* nm = w.norm + (size_t)pair * 8;
                const double sc = nm[0];
                const double T1[9] = {sc, 0, -sc * nm[1], 0, sc, -sc * nm[2], 0, 0, 1};
                const double T2[9] = {sc, 0, -sc * nm[3], 0, sc, -sc * nm[4], 0, 0, 1};
                double A[9], F[9], nn = 0;
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) { double v = 0; for (int q = 0; q < 3; q++) v += Fn[i * 3 + q] * T1[q * 3 + j]; A[i * 3 + j] = v; }
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) { double v = 0; for (int q = 0; q < 3; q++) v += T2[q * 3 + i] * A[q * 3 + j]; F[i * 3 + j] = v; nn += v * v; }
                const double sF = fabs(F[8]) > 1e-12 * sqrt(nn) ? 1.0 / F[8] : 1.0 / sqrt(nn);
                for (int j = 0; j < 9; j++) a.d_E[(size_t)pair * 9 + j] = F[j] * sF;
            }
        }
        return;
    }
    if (tid == 0) {
        // decompose: R1 = U W V^T, R2 = U W^T V^T, t = u3
        double U[9], Vm[9], sg[3], Ef[9];
        for (int i = 0; i < 9; i++) Ef[i] = s_E[i];
        s_ok = svd3_rank2(Ef, U, Vm, sg) ? 1 : 0;
        if (s_ok) {
            const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
            double UW[9], UWt[9];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    double s1 = 0, s2 = 0;
                    for (int q = 0; q < 3; q++) { s1 += U[i * 3 + q] * W[q * 3 + j]; s2 += U[i * 3 + q] * W[j * 3 + q]; }
                    UW[i * 3 + j] = s1; UWt[i * 3 + j] = s2;
                }
            for (int cnd = 0; cnd < 4; cnd++) {
                const double* M = (cnd & 1) ? UWt : UW;   // candidates: (R1,t) (R2,t) (R1,-t) (R2,-t)
                double sgn = cnd >= 2 ? -1.0 : 1.0;
                for (int i = 0; i < 3; i++) {
                    for (int j = 0; j < 3; j++) {
                        double s = 0;
                        for (int q = 0; q < 3; q++) s += M[i * 3 + q] * Vm[j * 3 + q];
                        s_P[cnd][i * 4 + j] = s;
                    }
                    s_P[cnd][i * 4 + 3] = sgn * U[i * 3 + 2];
                }
            }
        }
    }
    __syncthreads();
    if (!s_ok) {
        if (tid == 0) {
            a.d_n_points[pair] = 0;
            if (pose) for (int j = 0; j < 12; j++) pose[j] = __longlong_as_double(0x7FF8000000000000ll);
        }
        return;
    }
    // ---- final RANSAC mask + cheirality vote (recoverPose: depth in (0, 50) in both cameras)
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const double dist = 50.0;
    int good[4] = {0, 0, 0, 0};
    uint8_t* cbits = w.cbits + (size_t)pair * a.cap;  // per correspondence: bit cnd = in front of both cameras under candidate cnd
    double E[9];
    for (int j = 0; j < 9; j++) E[j] = s_E[j];
    for (int i = tid; i < m; i += TVF_BLOCK) {
        double x1 = xn[4 * i], y1 = xn[4 * i + 1], x2 = xn[4 * i + 2], y2 = xn[4 * i + 3];
        unsigned bits = 0;
        const bool consider = given ? (!a.d_mask_in || a.d_mask_in[(size_t)pair * a.cap + qidx[i]] != 0) : sampson(E, x1, y1, x2, y2) <= thr2;
        if (consider) {
            for (int cnd = 0; cnd < 4; cnd++) {
                double Pc[12], X[4];
                for (int j = 0; j < 12; j++) Pc[j] = s_P[cnd][j];
                dlt_point(P0, Pc, x1, y1, x2, y2, X);
                bool ok = X[2] * X[3] > 0;
                double iw = 1.0 / X[3];
                double qx = X[0] * iw, qy = X[1] * iw, qz = X[2] * iw;
                ok = ok && qz < dist;
                double z2 = Pc[8] * qx + Pc[9] * qy + Pc[10] * qz + Pc[11];
                ok = ok && z2 > 0 && z2 < dist;
                if (ok) { bits |= 1u << cnd; good[cnd]++; }
            }
            bits |= 16u;
            if (ran_out) ran_out[qidx[i]] = 1;
        }
        cbits[i] = (uint8_t)bits;  // (read back by the same thread below)
    }
    int g[4];
    for (int cnd = 0; cnd < 4; cnd++) g[cnd] = block_sum_i(good[cnd], s_redi);
    if (tid == 0) {
        int win;
        if (g[0] >= g[1] && g[0] >= g[2] && g[0] >= g[3]) win = 0;
        else if (g[1] >= g[0] && g[1] >= g[2] && g[1] >= g[3]) win = 1;
        else if (g[2] >= g[0] && g[2] >= g[1] && g[2] >= g[3]) win = 2;
        else win = 3;
        s_win = win;
        // pixel-space projection matrices P1 = K [I|0], P2 = K [R|t]  (utils.py:137-160)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 4; j++) {
                double s1 = 0, s2 = 0;
                for (int q = 0; q < 3; q++) { s1 += a.K[i * 3 + q] * P0[q * 4 + j]; s2 += a.K[i * 3 + q] * s_P[win][q * 4 + j]; }
                s_Ppix[0][i * 4 + j] = s1; s_Ppix[1][i * 4 + j] = s2;
            }
        if (pose) {
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) pose[i * 3 + j] = s_P[win][i * 4 + j];
                pose[9 + i] = s_P[win][i * 4 + 3];
            }
        }
        if (a.d_E) for (int j = 0; j < 9; j++) a.d_E[(size_t)pair * 9 + j] = s_E[j];
        a.d_n_points[pair] = g[win];
    }
    __syncthreads();
    const int win = s_win;
    double Pa[12], Pb[12];
    for (int j = 0; j < 12; j++) { Pa[j] = s_Ppix[0][j]; Pb[j] = s_Ppix[1][j]; }
    for (int i = tid; i < m; i += TVF_BLOCK) {
        const unsigned bits = cbits[i];
        if (!((bits >> win) & 1u)) continue;
        double X[4];
        dlt_point(Pa, Pb, (double)px[4 * i], (double)px[4 * i + 1], (double)px[4 * i + 2], (double)px[4 * i + 3], X);
        float xf = (float)X[0], yf = (float)X[1], zf = (float)X[2], wf = (float)X[3];
        int o = qidx[i];
        Xout[3 * o] = xf / wf; Xout[3 * o + 1] = yf / wf; Xout[3 * o + 2] = zf / wf;
        if (inl_out) inl_out[o] = 1;
    }
}

struct TriArgs { double P1[12], P2[12]; };

__global__ void k_triangulate(TriArgs t, const float* __restrict__ p1, const float* __restrict__ p2, int n, float* __restrict__ X4) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double X[4];
    dlt_point(t.P1, t.P2, (double)p1[2 * i], (double)p1[2 * i + 1], (double)p2[2 * i], (double)p2[2 * i + 1], X);
    for (int k = 0; k < 4; k++) X4[4 * i + k] = (float)X[k];
}

int triangulate_launch(mo_ctx* c, const double* P1, const double* P2, const float* d_p1, const float* d_p2, int n, float* d_X4) {
    if (n <= 0) return MO_OK;
    TriArgs t;
    for (int i = 0; i < 12; i++) { t.P1[i] = P1[i]; t.P2[i] = P2[i]; }
    hipLaunchKernelGGL(k_triangulate, dim3((n + 127) / 128), dim3(128), 0, c->stream, t, d_p1, d_p2, n, d_X4);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

int twoview_launch(mo_ctx* c, const TwoViewArgs& a_in) {
    if (a_in.n_pairs <= 0) return MO_OK;
    // (a pair takes any number of correspondences: the sampler is exact for every m, the per-correspondence state lives in the workspace)
    TwoViewArgs a = a_in;
    a.flags = c->flags_cur ? c->flags_cur : c->d_flags;
    if (!a.d_E_in && (a.n_hyp < 1 || a.n_hyp > (1 << 20))) return mo_fail(c, MO_ERR_ARG, "n_hyp out of range");
    size_t need = twoview_workspace_bytes(a.n_pairs, a.cap, a.n_hyp);
    int rc = mo_reserve(c, c->d_tv, c->tv_bytes, need);
    if (rc) return rc;
    TvWork w = carve(c->d_tv, a.n_pairs, a.cap, a.n_hyp);
    hipLaunchKernelGGL(k_tv_prep, dim3(a.n_pairs), dim3(TV_BLOCK), 0, c->stream, a, w);
    if (a.d_E_in) {  // recoverPose on a given E: decomposition + cheirality vote + triangulation only
        hipLaunchKernelGGL(k_tv_finish, dim3(a.n_pairs), dim3(TVF_BLOCK), 0, c->stream, a, w);
        HIPCHK(c, hipGetLastError());
        return MO_OK;
    }
    const int staged = a.n_hyp >= 512;  // below that the bound of one or two blocks prunes too little to pay for the second stage
    const int first_num = TV_FIRST_NUM;
    const dim3 hyp_grid((a.n_hyp + TV_BLOCK - 1) / TV_BLOCK, a.n_pairs);
    hipLaunchKernelGGL(k_tv_hyp, hyp_grid, dim3(TV_BLOCK), 0, c->stream, a, w, staged, first_num);
    if (staged) {
        const int one_pair = a.n_pairs == 1;
        if (!one_pair) hipLaunchKernelGGL(k_tv_tasks, dim3(1), dim3(TV_BLOCK), 0, c->stream, a, w);
        hipLaunchKernelGGL(k_tv_score, dim3((unsigned)((a.n_hyp + 63) / 64) * a.n_pairs), dim3(64 * TV_PARTS), 0, c->stream, a, w, one_pair);
    }
    hipLaunchKernelGGL(k_tv_finish, dim3(a.n_pairs), dim3(TVF_BLOCK), 0, c->stream, a, w);
    HIPCHK(c, hipGetLastError());
    return MO_OK;
}

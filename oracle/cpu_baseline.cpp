// oracle/cpu_baseline.cpp -- TEST / BENCH INFRASTRUCTURE ONLY: the CPU baseline leg of bench.py (kind "port").
//
// The hot path (extract -> match -> two-view) of the repo's own CPU restatement, compiled -O3 for the host and run with one
// OpenMP thread per core over frames / frame pairs, so that bench.py can state "this is what all host cores of the GPU box do"
// next to the MI355X number.  cv2 is not installed on the box, hence "port", not "reference".
// Extraction and matching are the functions of orb_oracle.cpp (included below, same arithmetic as the parity oracle);
// the two-view stage is a C++ port of oracle/geom_oracle.py (8-point E RANSAC over n_hyp hypotheses, MSAC ranking,
// adaptive-threshold refits, recoverPose-style cheirality vote, DLT) with Jacobi eigen-solvers instead of numpy's LAPACK.
// Only bench.py's cpu_baseline() and tests/test_cpu_baseline.py call this library; the product path never does.
#include "orb_oracle.cpp"

#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>

namespace {

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (a destroyed: eigenvalues on its diagonal; v: eigenvectors in columns)
void jacobi_sym(int n, double* a, double* v) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) v[i * n + j] = i == j;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) (i == j ? diag : off) += a[i * n + j] * a[i * n + j];
        if (off <= 1e-32 * diag) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = a[p * n + q];
                if (apq == 0.0) continue;
                double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq; a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk; a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq; v[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

// smallest eigenvector of the normal matrix of `rows` constraint rows of 9
void null9(const double* rows, int n, double* out) {
    double N[81] = {0}, V[81];
    for (int r = 0; r < n; r++)
        for (int i = 0; i < 9; i++)
            for (int j = i; j < 9; j++) N[i * 9 + j] += rows[r * 9 + i] * rows[r * 9 + j];
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < i; j++) N[i * 9 + j] = N[j * 9 + i];
    jacobi_sym(9, N, V);
    int k = 0;
    for (int i = 1; i < 9; i++) if (N[i * 9 + i] < N[k * 9 + k]) k = i;
    for (int i = 0; i < 9; i++) out[i] = V[i * 9 + k];
}

struct Svd3 { double U[9], V[9], s[3]; bool ok; };

Svd3 svd3(const double* E) {  // rank >= 2 assumed: right vectors from eig(E^T E), u_i = E v_i / s_i, third by cross products
    Svd3 r;
    double A[9], Vv[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A[i * 3 + j] = E[i] * E[j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
    jacobi_sym(3, A, Vv);
    int o[3] = {0, 1, 2};
    std::sort(o, o + 3, [&](int a, int b) { return A[a * 3 + a] > A[b * 3 + b]; });
    double v[3][3], u[3][3];
    for (int k = 0; k < 3; k++) {
        r.s[k] = std::sqrt(std::max(A[o[k] * 3 + o[k]], 0.0));
        for (int i = 0; i < 3; i++) v[k][i] = Vv[i * 3 + o[k]];
    }
    r.ok = r.s[0] > 0 && r.s[1] > 1e-12 * r.s[0];
    if (!r.ok) return r;
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < 3; i++) u[k][i] = (E[i * 3] * v[k][0] + E[i * 3 + 1] * v[k][1] + E[i * 3 + 2] * v[k][2]) / r.s[k];
    double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2];
    for (int i = 0; i < 3; i++) u[1][i] -= d * u[0][i];
    for (int k = 0; k < 2; k++) {
        double n = std::sqrt(u[k][0] * u[k][0] + u[k][1] * u[k][1] + u[k][2] * u[k][2]);
        for (int i = 0; i < 3; i++) u[k][i] /= n;
    }
    auto cross = [](const double* a, const double* b, double* c) {
        c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
    };
    cross(u[0], u[1], u[2]);
    cross(v[0], v[1], v[2]);
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) { r.U[i * 3 + k] = u[k][i]; r.V[i * 3 + k] = v[k][i]; }
    return r;
}

bool project_essential(double* E) {
    Svd3 s = svd3(E);
    if (!s.ok) return false;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) E[i * 3 + j] = s.U[i * 3] * s.V[j * 3] + s.U[i * 3 + 1] * s.V[j * 3 + 1];
    return true;
}

inline double sampson(const double* E, const double* x) {
    double ex0 = E[0] * x[0] + E[1] * x[1] + E[2], ex1 = E[3] * x[0] + E[4] * x[1] + E[5], ex2 = E[6] * x[0] + E[7] * x[1] + E[8];
    double et0 = E[0] * x[2] + E[3] * x[3] + E[6], et1 = E[1] * x[2] + E[4] * x[3] + E[7];
    double num = x[2] * ex0 + x[3] * ex1 + ex2;
    return num * num / (ex0 * ex0 + ex1 * ex1 + et0 * et0 + et1 * et1);
}

inline void design_row(const double* x, double* r) {
    r[0] = x[2] * x[0]; r[1] = x[2] * x[1]; r[2] = x[2]; r[3] = x[3] * x[0]; r[4] = x[3] * x[1]; r[5] = x[3];
    r[6] = x[0]; r[7] = x[1]; r[8] = 1.0;
}

inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void dlt_point(const double* P1, const double* P2, double x1, double y1, double x2, double y2, double* X) {
    double A[16], S[16], V[16];
    for (int k = 0; k < 4; k++) {
        A[k] = x1 * P1[8 + k] - P1[k]; A[4 + k] = y1 * P1[8 + k] - P1[4 + k];
        A[8 + k] = x2 * P2[8 + k] - P2[k]; A[12 + k] = y2 * P2[8 + k] - P2[4 + k];
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) S[i * 4 + j] = A[i] * A[j] + A[4 + i] * A[4 + j] + A[8 + i] * A[8 + j] + A[12 + i] * A[12 + j];
    jacobi_sym(4, S, V);
    int k = 0;
    for (int i = 1; i < 4; i++) if (S[i * 4 + i] < S[k * 4 + k]) k = i;
    for (int i = 0; i < 4; i++) X[i] = V[i * 4 + k];
}

// geom_oracle.init_two_view without the map points' float32 casts: returns the number of pose inliers, R (9), t (3)
int two_view(const float* p1, const float* p2, int m, const double* K, double thr_px, int n_hyp, uint64_t seed, double* R, double* t) {
    if (m < 8) return 0;
    std::vector<double> xn((size_t)m * 4);
    for (int i = 0; i < m; i++) {
        xn[4 * i] = ((double)p1[2 * i] - K[2]) / K[0]; xn[4 * i + 1] = ((double)p1[2 * i + 1] - K[5]) / K[4];
        xn[4 * i + 2] = ((double)p2[2 * i] - K[2]) / K[0]; xn[4 * i + 3] = ((double)p2[2 * i + 1] - K[5]) / K[4];
    }
    const double thr = thr_px / ((K[0] + K[4]) / 2.0), thr2 = thr * thr;
    double bestE[9] = {0};
    float best_cost = INFINITY;
    for (int h = 0; h < n_hyp; h++) {
        uint64_t s = seed + (uint64_t)(h + 1) * 0xD1B54A32D192ED03ull;
        int idx[8];
        for (int k = 0; k < 8;) {
            int c = (int)(splitmix64(s) % (uint64_t)m);
            bool dup = false;
            for (int j = 0; j < k; j++) dup |= idx[j] == c;
            if (!dup) idx[k++] = c;
        }
        double rows[72], E[9];
        for (int k = 0; k < 8; k++) design_row(&xn[(size_t)idx[k] * 4], rows + 9 * k);
        null9(rows, 8, E);
        if (!project_essential(E)) continue;
        double cost = 0;
        for (int i = 0; i < m; i++) cost += std::min(sampson(E, &xn[(size_t)i * 4]), thr2);
        if ((float)cost < best_cost) { best_cost = (float)cost; std::memcpy(bestE, E, sizeof(E)); }
    }
    if (!(best_cost < INFINITY)) return 0;
    double E[9];
    std::memcpy(E, bestE, sizeof(E));
    int n0 = 0;
    double sd = 0;
    for (int i = 0; i < m; i++) { double d = sampson(E, &xn[(size_t)i * 4]); if (d <= thr2) { n0++; sd += d; } }
    if (n0 < 8) return 0;
    const double lo2 = thr2 / 4096.0;
    double tau2 = std::min(std::max(9.0 * sd / n0, lo2), thr2), tau2_prev = -1;
    int c_prev = -1;
    std::vector<double> rows((size_t)m * 9);
    for (int it = 0; it < 5; it++) {
        int c = 0;
        sd = 0;
        for (int i = 0; i < m; i++) {
            double d = sampson(E, &xn[(size_t)i * 4]);
            if (d <= tau2) { design_row(&xn[(size_t)i * 4], &rows[(size_t)c * 9]); c++; sd += d; }
        }
        if (c < 8 || 2 * c < n0) break;
        if (c == c_prev && tau2 == tau2_prev) break;
        c_prev = c; tau2_prev = tau2;
        double En[9];
        null9(rows.data(), c, En);
        if (!project_essential(En)) break;
        std::memcpy(E, En, sizeof(E));
        tau2 = std::min(std::max(9.0 * sd / c, lo2), thr2);
    }
    Svd3 s = svd3(E);
    if (!s.ok) return 0;
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double P[4][12];
    for (int cnd = 0; cnd < 4; cnd++)
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                double v = 0;
                for (int q = 0; q < 3; q++)
                    for (int r = 0; r < 3; r++) v += s.U[i * 3 + q] * ((cnd & 1) ? W[r * 3 + q] : W[q * 3 + r]) * s.V[j * 3 + r];
                P[cnd][i * 4 + j] = v;
            }
            P[cnd][i * 4 + 3] = (cnd >= 2 ? -1.0 : 1.0) * s.U[i * 3 + 2];
        }
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    int good[4] = {0, 0, 0, 0};
    for (int i = 0; i < m; i++) {
        const double* x = &xn[(size_t)i * 4];
        if (sampson(E, x) > thr2) continue;
        for (int cnd = 0; cnd < 4; cnd++) {
            double X[4];
            dlt_point(P0, P[cnd], x[0], x[1], x[2], x[3], X);
            bool ok = X[2] * X[3] > 0;
            double qx = X[0] / X[3], qy = X[1] / X[3], qz = X[2] / X[3];
            ok = ok && qz < 50.0;
            double z2 = P[cnd][8] * qx + P[cnd][9] * qy + P[cnd][10] * qz + P[cnd][11];
            good[cnd] += ok && z2 > 0 && z2 < 50.0;
        }
    }
    int w = 0;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) w = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) w = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) w = 2;
    else w = 3;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[i * 3 + j] = P[w][i * 4 + j];
        t[i] = P[w][i * 4 + 3];
    }
    // final DLT triangulation of the pose inliers in pixel space (timed like the device path does it)
    double Pa[12], Pb[12];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double a = 0, b = 0;
            for (int q = 0; q < 3; q++) { a += K[i * 3 + q] * P0[q * 4 + j]; b += K[i * 3 + q] * P[w][q * 4 + j]; }
            Pa[i * 4 + j] = a; Pb[i * 4 + j] = b;
        }
    volatile double sink = 0;
    for (int i = 0; i < m; i++)
        if (sampson(E, &xn[(size_t)i * 4]) <= thr2) {
            double X[4];
            dlt_point(Pa, Pb, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], X);
            sink = sink + X[0];
        }
    return good[w];
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" {

int orc_baseline_threads(void) { return omp_get_max_threads(); }

// single two-view problem (tests/test_cpu_baseline.py checks it against geom_oracle.py)
int orc_two_view(const float* p1, const float* p2, int m, const double* K, double thr_px, int n_hyp, uint64_t seed, double* R, double* t) {
    return two_view(p1, p2, m, K, thr_px, n_hyp, seed, R, t);
}

// The hot path over a batch of gray frames with `threads` OpenMP threads (0 = all): extract every frame, match consecutive
// pairs (2-NN + ratio), two-view on the first n_pose pairs.  times[3] = wall seconds of the three stages;
// counts[3] = total keypoints, total ratio-test matches, total pose inliers.  Returns the number of threads used.
int orc_baseline_run(const uint8_t* frames, int n, int w, int h, int nfeatures, double ratio, int n_pose, const double* K, int n_hyp,
                     int threads, double* times, long long* counts) {
    if (threads > 0) omp_set_num_threads(threads);
    const int used = omp_get_max_threads();
    orc_orb_params prm = {nfeatures, 1.2f, 8, 31, 7};
    const int cap = 4 * nfeatures + 1024;
    std::vector<std::vector<orc_keypoint>> kps(n);
    std::vector<std::vector<uint8_t>> desc(n);
    std::vector<int> cnt(n, 0);
    double t0 = now_s();
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n; i++) {
        kps[i].resize(cap); desc[i].resize((size_t)cap * 32);
        int k = orc_orb_detect_compute(frames + (size_t)i * w * h, w, h, &prm, kps[i].data(), desc[i].data(), cap);
        cnt[i] = std::max(k, 0);
    }
    double t1 = now_s();
    std::vector<std::vector<int32_t>> idx(std::max(n - 1, 0)), dist(std::max(n - 1, 0));
    std::vector<std::vector<uint8_t>> pass(std::max(n - 1, 0));
    long long n_match = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : n_match)
    for (int i = 0; i < n - 1; i++) {
        idx[i].resize((size_t)cnt[i] * 2 + 2); dist[i].resize((size_t)cnt[i] * 2 + 2); pass[i].resize(cnt[i] + 1);
        orc_match_knn2(desc[i].data(), cnt[i], desc[i + 1].data(), cnt[i + 1], idx[i].data(), dist[i].data());
        orc_ratio_test(idx[i].data(), dist[i].data(), cnt[i], ratio, 1, pass[i].data());
        for (int q = 0; q < cnt[i]; q++) n_match += pass[i][q];
    }
    double t2 = now_s();
    long long n_inl = 0;
    n_pose = std::min(n_pose, n - 1);
#pragma omp parallel for schedule(dynamic) reduction(+ : n_inl)
    for (int i = 0; i < n_pose; i++) {
        std::vector<float> p1, p2;
        for (int q = 0; q < cnt[i]; q++)
            if (pass[i][q]) {
                const orc_keypoint& a = kps[i][q];
                const orc_keypoint& b = kps[i + 1][idx[i][2 * q]];
                p1.push_back(a.x); p1.push_back(a.y); p2.push_back(b.x); p2.push_back(b.y);
            }
        double R[9], t[3];
        n_inl += two_view(p1.data(), p2.data(), (int)p1.size() / 2, K, 3.0, n_hyp, 4096 + (uint64_t)i * 0x632BE59BD9B4E019ull, R, t);
    }
    double t3 = now_s();
    times[0] = t1 - t0; times[1] = t2 - t1; times[2] = t3 - t2;
    long long nk = 0;
    for (int i = 0; i < n; i++) nk += cnt[i];
    counts[0] = nk; counts[1] = n_match; counts[2] = n_inl;
    return used;
}

}  // extern "C"

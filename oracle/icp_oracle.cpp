// =====================================================================================
// icp_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code)
//
// Dependency-free C++14 restatement of the ICP hot path of PetropoulakisPanagiotis/ICP-Variants
// (reference mounted at /root/reference, cited below as file:line relative to
// /root/reference/icp-variants/).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the shipped HIP path never calls it.
//
// PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this path
//  and cannot be built here (Eigen/FLANN/Ceres/PCL absent).  The only numeric
// anchor it holds is the 4 hand-picked bunny ground-truth correspondences (main.cpp:110-120),
// which tests/test_oracle_bunny.py checks.  Everything else is a restatement of the source.
//
// Arithmetic contract (build with -O2 -ffp-contract=off, no -ffast-math, x86-64 SSE2):
//   * all geometry is IEEE fp32, one rounding per operation, NO fused multiply-add;
//   * k-NN distance (3-D and 6-D): FLANN L2 functor order, sequential  ((d0^2+d1^2)+d2^2)+...
//     (NearestNeighbor.h:136,174 use flann::L2<float>; squared distance, squared threshold :182);
//   * projective / brute-force-class distance: Eigen fixed-size squaredNorm() reduction tree
//     d0^2 + (d1^2 + d2^2)   (NearestNeighbor.h:396; Eigen 3.3 redux_novec_unroller<0,3>);
//   * transformPoints: sequential  ((R_i0*x + R_i1*y) + R_i2*z) + t_i   (utils.h:114; the
//     rotation is a dynamic-size Block, so Eigen's coefficient product is a sequential loop);
//   * first (lowest-index) minimum wins: strict `minDist > dist` (NearestNeighbor.h:87,399).
// Two solver flavours are provided for every linear solve:
//   mode 0 "faithful": fp32 accumulations in the reference's order (sequential sums, fp32 system)
//   mode 1 "exact"   : identical fp32 rows/inputs, but sums and factorisations in fp64.
// Eigen's own blocked GEMM / Householder / Jacobi rounding cannot be reproduced bit-for-bit
// without Eigen; both flavours are documented restatements, the difference between them bounds
// the reference's own rounding noise.
// =====================================================================================
#include <cmath>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <vector>
#include <algorithm>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {
struct OrcMatch { int idx; float weight; };   // NearestNeighbor.h:7-10
}

namespace {

const float MINF_ = -std::numeric_limits<float>::infinity();   // Eigen.h:20-22

inline bool finite3(const float* p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }

// Pose is a column-major 4x4 fp32 matrix exactly like Eigen::Matrix4f::data().
inline float P(const float* pose, int r, int c) { return pose[c * 4 + r]; }

// ---- Eigen fixed-size-3 reductions: e0 + (e1 + e2) -----------------------------------
inline float dot3_tree(const float* a, const float* b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
inline float sqnorm3_tree(const float* a) { return a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]); }

// ---- (R^-1)^T of the pose's rotation block --------------------------------------------
// utils.h:129 recomputes rotation.inverse().transpose() per normal through Eigen's dynamic-size
// PartialPivLU in fp32.  That rounding is not reproducible without Eigen; the contract used by
// oracle AND device is the fp64 cofactor inverse rounded once to fp32 (what Eigen's result
// approximates to ~1-2 ulp).  Fixed operation order, no contraction.
void normal_matrix(const float* pose, float* N /*row-major 3x3*/) {
    double a = P(pose,0,0), b = P(pose,0,1), c = P(pose,0,2);
    double d = P(pose,1,0), e = P(pose,1,1), f = P(pose,1,2);
    double g = P(pose,2,0), h = P(pose,2,1), i = P(pose,2,2);
    double c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    double c10 = c * h - b * i, c11 = a * i - c * g, c12 = b * g - a * h;
    double c20 = b * f - c * e, c21 = c * d - a * f, c22 = a * e - b * d;
    double det = (a * c00 + b * c01) + c * c02;
    // inverse = adj/det with adj = cof^T ; (inverse)^T = cof/det
    N[0] = (float)(c00 / det); N[1] = (float)(c01 / det); N[2] = (float)(c02 / det);
    N[3] = (float)(c10 / det); N[4] = (float)(c11 / det); N[5] = (float)(c12 / det);
    N[6] = (float)(c20 / det); N[7] = (float)(c21 / det); N[8] = (float)(c22 / det);
}

inline void xform_point(const float* pose, const float* p, float* o) {   // utils.h:113-115
    for (int r = 0; r < 3; r++)
        o[r] = ((P(pose,r,0) * p[0] + P(pose,r,1) * p[1]) + P(pose,r,2) * p[2]) + P(pose,r,3);
}
inline void xform_normal(const float* N, const float* n, float* o) {     // utils.h:128-130
    for (int r = 0; r < 3; r++)
        o[r] = (N[r*3+0] * n[0] + N[r*3+1] * n[1]) + N[r*3+2] * n[2];
}

// ---- small dense linear algebra, templated on the working precision --------------------
// Column-pivoted Householder QR of A (m x n, row-major, n<=6) applied in place, rhs b too.
// After the call the leading n x n block of A holds R, b[0..n) holds (Q^T b), perm the pivots.
template <class T>
void householder_qr(std::vector<T>& A, std::vector<T>& b, size_t m, int n, int* perm) {
    std::vector<T> cn(n);
    for (int j = 0; j < n; j++) { perm[j] = j; T s = 0; for (size_t i = 0; i < m; i++) s += A[i*n+j] * A[i*n+j]; cn[j] = s; }
    for (int k = 0; k < n && (size_t)k < m; k++) {
        int piv = k; for (int j = k + 1; j < n; j++) if (cn[j] > cn[piv]) piv = j;
        if (piv != k) { for (size_t i = 0; i < m; i++) std::swap(A[i*n+k], A[i*n+piv]); std::swap(cn[k], cn[piv]); std::swap(perm[k], perm[piv]); }
        T nrm = 0; for (size_t i = k; i < m; i++) nrm += A[i*n+k] * A[i*n+k];
        nrm = std::sqrt(nrm);
        if (nrm == T(0)) continue;
        T alpha = A[(size_t)k*n+k] > 0 ? -nrm : nrm;
        std::vector<T> v(m - k);
        for (size_t i = k; i < m; i++) v[i-k] = A[i*n+k];
        v[0] -= alpha;
        T vn = 0; for (size_t i = 0; i < v.size(); i++) vn += v[i] * v[i];
        if (vn == T(0)) continue;
        for (int j = k; j < n; j++) {
            T s = 0; for (size_t i = k; i < m; i++) s += v[i-k] * A[i*n+j];
            s = T(2) * s / vn;
            for (size_t i = k; i < m; i++) A[i*n+j] -= s * v[i-k];
        }
        { T s = 0; for (size_t i = k; i < m; i++) s += v[i-k] * b[i]; s = T(2) * s / vn; for (size_t i = k; i < m; i++) b[i] -= s * v[i-k]; }
        for (int j = k + 1; j < n; j++) { T s = 0; for (size_t i = k + 1; i < m; i++) s += A[i*n+j] * A[i*n+j]; cn[j] = s; }
    }
}

// One-sided (Hestenes) Jacobi SVD of a small n x n matrix M (row-major): M V = U S.
// Returns singular values (unsorted) in s, U columns in M (normalised), V in V.
template <class T>
void jacobi_svd_small(T* M, int n, T* V, T* s) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i*n+j] = (i == j) ? T(1) : T(0);
    const T eps = std::numeric_limits<T>::epsilon();
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < n - 1; p++) for (int q = p + 1; q < n; q++) {
            T a = 0, b = 0, g = 0;
            for (int i = 0; i < n; i++) { a += M[i*n+p] * M[i*n+p]; b += M[i*n+q] * M[i*n+q]; g += M[i*n+p] * M[i*n+q]; }
            if (g == T(0) || std::fabs(g) <= eps * std::sqrt(a * b)) continue;
            rotated = true;
            T zeta = (b - a) / (T(2) * g);
            T t = (zeta >= 0 ? T(1) : T(-1)) / (std::fabs(zeta) + std::sqrt(T(1) + zeta * zeta));
            T c = T(1) / std::sqrt(T(1) + t * t), sn = c * t;
            for (int i = 0; i < n; i++) {
                T mp = M[i*n+p], mq = M[i*n+q]; M[i*n+p] = c * mp - sn * mq; M[i*n+q] = sn * mp + c * mq;
                T vp = V[i*n+p], vq = V[i*n+q]; V[i*n+p] = c * vp - sn * vq; V[i*n+q] = sn * vp + c * vq;
            }
        }
        if (!rotated) break;
    }
    for (int j = 0; j < n; j++) {
        T nr = 0; for (int i = 0; i < n; i++) nr += M[i*n+j] * M[i*n+j];
        nr = std::sqrt(nr); s[j] = nr;
        if (nr > T(0)) for (int i = 0; i < n; i++) M[i*n+j] /= nr;
    }
}

// Least squares min ||A x - b|| through QR + Jacobi SVD with Eigen's JacobiSVD::solve rank
// rule (ICPOptimizer.h:757-758): singular values <= diagSize*eps_f32*sigma_max are dropped.
template <class T>
void lstsq_svd(std::vector<T>& A, std::vector<T>& b, size_t m, int n, T* x) {
    int perm[8];
    householder_qr<T>(A, b, m, n, perm);
    T R[36], V[36], s[6], c[6];
    for (int i = 0; i < n; i++) { c[i] = (size_t)i < m ? b[i] : T(0); for (int j = 0; j < n; j++) R[i*n+j] = (j >= i && (size_t)i < m) ? A[(size_t)i*n+j] : T(0); }
    jacobi_svd_small<T>(R, n, V, s);
    T smax = 0; for (int j = 0; j < n; j++) smax = std::max(smax, s[j]);
    const T thr = T(n) * (T)std::numeric_limits<float>::epsilon() * smax;
    T y[6] = {0,0,0,0,0,0};
    for (int j = 0; j < n; j++) {
        if (!(s[j] > thr)) continue;
        T utc = 0; for (int i = 0; i < n; i++) utc += R[i*n+j] * c[i];   // U^T c
        T coef = utc / s[j];
        for (int i = 0; i < n; i++) y[i] += V[i*n+j] * coef;
    }
    for (int j = 0; j < n; j++) x[perm[j]] = y[j];
}

// Full-pivot LU solve with Eigen's FullPivLU rank rule (ICPOptimizer.h:866-868):
// pivots with |p| <= |maxpivot| * eps_f32 * n are treated as zero, their unknowns set to 0.
template <class T>
void fullpiv_lu_solve(T* M, T* rhs, int n, T* x) {
    int rowp[8], colp[8];
    for (int i = 0; i < n; i++) { rowp[i] = i; colp[i] = i; }
    T maxpiv = 0; int rank = n;
    const T thrf = (T)std::numeric_limits<float>::epsilon() * T(n);
    for (int k = 0; k < n; k++) {
        int pr = k, pc = k; T best = -1;
        for (int i = k; i < n; i++) for (int j = k; j < n; j++) { T v = std::fabs(M[i*n+j]); if (v > best) { best = v; pr = i; pc = j; } }
        if (k == 0) maxpiv = best;
        if (best > maxpiv) maxpiv = best;
        if (best == T(0)) { rank = k; break; }
        if (pr != k) { for (int j = 0; j < n; j++) std::swap(M[k*n+j], M[pr*n+j]); std::swap(rhs[k], rhs[pr]); std::swap(rowp[k], rowp[pr]); }
        if (pc != k) { for (int i = 0; i < n; i++) std::swap(M[i*n+k], M[i*n+pc]); std::swap(colp[k], colp[pc]); }
        for (int i = k + 1; i < n; i++) {
            T f = M[i*n+k] / M[k*n+k]; M[i*n+k] = f;
            for (int j = k + 1; j < n; j++) M[i*n+j] -= f * M[k*n+j];
            rhs[i] -= f * rhs[k];
        }
    }
    int r = 0; for (int k = 0; k < rank; k++) if (std::fabs(M[k*n+k]) > maxpiv * thrf) r++; else break;
    T y[8] = {0,0,0,0,0,0,0,0};
    for (int k = r - 1; k >= 0; k--) { T sacc = rhs[k]; for (int j = k + 1; j < r; j++) sacc -= M[k*n+j] * y[j]; y[k] = sacc / M[k*n+k]; }
    for (int k = 0; k < n; k++) x[colp[k]] = (k < r) ? y[k] : T(0);
}

template <class T> T det3(const T* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

void mat4_identity(float* M) { for (int i = 0; i < 16; i++) M[i] = (i % 5 == 0) ? 1.f : 0.f; }
// C = A*B, column-major 4x4 fp32, Eigen fixed 4x4 packet product: sequential over k (ICPOptimizer.h:614-620)
void mat4_mul(const float* A, const float* B, float* C) {
    float T4[16];
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) {
        float acc = A[0*4+r] * B[c*4+0];
        acc = acc + A[1*4+r] * B[c*4+1];
        acc = acc + A[2*4+r] * B[c*4+2];
        acc = acc + A[3*4+r] * B[c*4+3];
        T4[c*4+r] = acc;
    }
    std::memcpy(C, T4, sizeof(T4));
}
// 3x3 row-major fp32 product with the Eigen fixed-size coefficient tree e0+(e1+e2)
void mat3_mul(const float* A, const float* B, float* C) {
    float T9[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        T9[r*3+c] = A[r*3+0] * B[0*3+c] + (A[r*3+1] * B[1*3+c] + A[r*3+2] * B[2*3+c]);
    std::memcpy(C, T9, sizeof(T9));
}
inline void mat3_vec(const float* A, const float* v, float* o) {
    for (int r = 0; r < 3; r++) o[r] = A[r*3+0] * v[0] + (A[r*3+1] * v[1] + A[r*3+2] * v[2]);
}
void set_pose(float* pose, const float* R /*row-major*/, const float* t) {
    mat4_identity(pose);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) pose[c*4+r] = R[r*3+c]; pose[3*4+r] = t[r]; }
}

// computeMean, utils.h:136-145 / ProcrustesAligner.h:32-41 : fp32 running sum, then / n
template <class T>
void compute_mean(const float* pts, size_t n, float* mean) {
    T s0 = 0, s1 = 0, s2 = 0;
    for (size_t i = 0; i < n; i++) { s0 += pts[i*3+0]; s1 += pts[i*3+1]; s2 += pts[i*3+2]; }
    T den = (T)(float)n;
    mean[0] = (float)(s0 / den); mean[1] = (float)(s1 / den); mean[2] = (float)(s2 / den);
}

// ---- ProcrustesAligner::estimatePose, ProcrustesAligner.h:6-72 ---------------------------
template <class T>
int solve_p2p_t(const float* s, const float* d, const float* w, size_t n, float* pose) {
    if (n == 0) return 1;                                     // reference ASSERT hangs (ICPOptimizer.h:668)
    float sm[3], dm[3];
    compute_mean<T>(s, n, sm); compute_mean<T>(d, n, dm);     // :13-14 (unweighted)
    T A[9] = {0,0,0,0,0,0,0,0,0};                             // A = targetMatrix^T * sourceMatrix  (:55)
    for (size_t i = 0; i < n; i++) {
        float sr[3], dr[3];
        for (int k = 0; k < 3; k++) { sr[k] = w[i] * (s[i*3+k] - sm[k]); dr[k] = d[i*3+k] - dm[k]; }   // :51-52 fp32 rows
        for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) A[j*3+k] += (T)dr[j] * (T)sr[k];
    }
    // JacobiSVD(A, FullU|FullV) (:56): here A^T A = V S^2 V^T by one-sided Jacobi, U = A V S^-1
    T M[9], V[9], sv[3];
    for (int i = 0; i < 9; i++) M[i] = A[i];
    jacobi_svd_small<T>(M, 3, V, sv);                         // M columns = U, sv unsorted
    int ord[3] = {0,1,2};
    std::sort(ord, ord + 3, [&](int a, int b) { return sv[a] > sv[b]; });
    T U[9], Vs[9];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { U[r*3+c] = M[r*3+ord[c]]; Vs[r*3+c] = V[r*3+ord[c]]; }
    // a zero singular value leaves its U column undefined: complete U to an orthonormal basis
    if (!(sv[ord[2]] > T(0))) {
        if (!(sv[ord[1]] > T(0))) {
            if (!(sv[ord[0]] > T(0))) { for (int i = 0; i < 9; i++) U[i] = (i % 4 == 0) ? T(1) : T(0); }
            else {
                T u0[3] = {U[0], U[3], U[6]}; int k = std::fabs(u0[0]) < std::fabs(u0[1]) ? (std::fabs(u0[0]) < std::fabs(u0[2]) ? 0 : 2) : (std::fabs(u0[1]) < std::fabs(u0[2]) ? 1 : 2);
                T e[3] = {0,0,0}; e[k] = 1; T dp = u0[k];
                T u1[3] = {e[0] - dp*u0[0], e[1] - dp*u0[1], e[2] - dp*u0[2]};
                T nr = std::sqrt(u1[0]*u1[0] + u1[1]*u1[1] + u1[2]*u1[2]);
                for (int r = 0; r < 3; r++) U[r*3+1] = u1[r] / nr;
            }
        }
        T a0[3] = {U[0], U[3], U[6]}, a1[3] = {U[1], U[4], U[7]};
        U[2] = a0[1]*a1[2] - a0[2]*a1[1]; U[5] = a0[2]*a1[0] - a0[0]*a1[2]; U[8] = a0[0]*a1[1] - a0[1]*a1[0];
    }
    T UVt[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { T acc = 0; for (int k = 0; k < 3; k++) acc += U[r*3+k] * Vs[c*3+k]; UVt[r*3+c] = acc; }
    T dd = det3<T>(UVt);                                      // :60-62
    float R[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        T acc = 0; for (int k = 0; k < 3; k++) acc += U[r*3+k] * (k == 2 ? dd : T(1)) * Vs[c*3+k];
        R[r*3+c] = (float)acc;                                // :64
    }
    float tr[3] = {dm[0] - sm[0], dm[1] - sm[1], dm[2] - sm[2]};   // :70
    float Rt[3], Rd[3], t[3];
    mat3_vec(R, tr, Rt); mat3_vec(R, dm, Rd);
    for (int k = 0; k < 3; k++) t[k] = (Rt[k] - Rd[k]) + dm[k];    // :26
    set_pose(pose, R, t);
    return 0;
}

// Rows of the reference's 4n x 6 system.  kind 0: point-to-plane (ICPOptimizer.h:687-751),
// kind 1: symmetric (ICPOptimizer.h:800-853; s,d already centred by the caller, n = n_t + n_s).
inline void build_rows(int kind, const float* s, const float* d, const float* n, float w, float A[4][6], float b[4]) {
    const float LAMBDA_POINT = 0.1f, LAMBDA_MAIN = 1.0f;         // :737-738 / :839-840 (constraints.h:46,91,142)
    if (kind == 0) {
        A[0][0] = n[2]*s[1] - n[1]*s[2]; A[0][1] = n[0]*s[2] - n[2]*s[0]; A[0][2] = n[1]*s[0] - n[0]*s[1];   // :698-700
        A[0][3] = n[0]; A[0][4] = n[1]; A[0][5] = n[2];
        b[0] = ((n[0]*d[0] + n[1]*d[1]) + n[2]*d[2]) - ((n[0]*s[0] + n[1]*s[1]) + n[2]*s[2]);               // :708-710
    } else {
        float sd[3] = {s[0] + d[0], s[1] + d[1], s[2] + d[2]};      // (s~ + d~)
        float ds[3] = {d[0] - s[0], d[1] - s[1], d[2] - s[2]};      // (d~ - s~)
        A[0][0] = sd[1]*n[2] - sd[2]*n[1]; A[0][1] = sd[2]*n[0] - sd[0]*n[2]; A[0][2] = sd[0]*n[1] - sd[1]*n[0];   // :814 cross
        A[0][3] = n[0]; A[0][4] = n[1]; A[0][5] = n[2];                                                          // :815
        b[0] = ds[0]*n[0] + (ds[1]*n[1] + ds[2]*n[2]);                                                           // :811 Eigen dot tree
    }
    A[1][0] = 0;     A[1][1] = s[2];  A[1][2] = -s[1]; A[1][3] = 1; A[1][4] = 0; A[1][5] = 0; b[1] = d[0] - s[0];   // :718-721
    A[2][0] = -s[2]; A[2][1] = 0;     A[2][2] = s[0];  A[2][3] = 0; A[2][4] = 1; A[2][5] = 0; b[2] = d[1] - s[1];   // :724-727
    A[3][0] = s[1];  A[3][1] = -s[0]; A[3][2] = 0;     A[3][3] = 0; A[3][4] = 0; A[3][5] = 1; b[3] = d[2] - s[2];   // :730-733
    const float f0 = LAMBDA_MAIN * w, f1 = LAMBDA_POINT * w;        // :740-750
    for (int c = 0; c < 6; c++) { A[0][c] *= f0; A[1][c] *= f1; A[2][c] *= f1; A[3][c] *= f1; }
    b[0] *= f0; b[1] *= f1; b[2] *= f1; b[3] *= f1;
}

// ---- estimatePosePointToPlane, ICPOptimizer.h:676-782 ------------------------------------
template <class T>
int solve_p2plane_t(const float* s, const float* d, const float* nt, const float* w, size_t n, float* pose, double* x_out) {
    if (n == 0) return 1;
    std::vector<T> A(4 * n * 6), b(4 * n);
    for (size_t i = 0; i < n; i++) {
        float Ar[4][6], br[4];
        build_rows(0, s + i*3, d + i*3, nt + i*3, w[i], Ar, br);
        for (int r = 0; r < 4; r++) { for (int c = 0; c < 6; c++) A[(4*i+r)*6+c] = (T)Ar[r][c]; b[4*i+r] = (T)br[r]; }
    }
    T x[6];
    lstsq_svd<T>(A, b, 4 * n, 6, x);                                // :757-758
    if (x_out) for (int k = 0; k < 6; k++) x_out[k] = (double)x[k];
    float al = (float)x[0], be = (float)x[1], ga = (float)x[2];   // :768
    float ca = std::cos(al), sa = std::sin(al), cb = std::cos(be), sb = std::sin(be), cg = std::cos(ga), sg = std::sin(ga);
    float Rx[9] = {1,0,0, 0,ca,-sa, 0,sa,ca}, Ry[9] = {cb,0,sb, 0,1,0, -sb,0,cb}, Rz[9] = {cg,-sg,0, sg,cg,0, 0,0,1};
    float Rxy[9], R[9];
    mat3_mul(Rx, Ry, Rxy); mat3_mul(Rxy, Rz, R);                    // :771-773
    float t[3] = {(float)x[3], (float)x[4], (float)x[5]};           // :775
    set_pose(pose, R, t);
    return 0;
}

// ---- estimatePoseSymmetricICP, ICPOptimizer.h:784-898 --------------------------------------
template <class T>
int solve_symmetric_t(const float* s, const float* d, const float* ns, const float* nt, const float* w, size_t n, float* pose, double* x_out) {
    if (n == 0) return 1;
    float sm[3], dm[3];
    compute_mean<T>(s, n, sm); compute_mean<T>(d, n, dm);          // :797-798
    T M[36], g[6];
    for (int i = 0; i < 36; i++) M[i] = 0;
    for (int i = 0; i < 6; i++) g[i] = 0;
    for (size_t i = 0; i < n; i++) {
        float sc[3], dc[3], nn[3];
        for (int k = 0; k < 3; k++) { sc[k] = s[i*3+k] - sm[k]; dc[k] = d[i*3+k] - dm[k]; nn[k] = nt[i*3+k] + ns[i*3+k]; }   // :806-809
        float Ar[4][6], br[4];
        build_rows(1, sc, dc, nn, w[i], Ar, br);
        for (int r = 0; r < 4; r++) for (int a = 0; a < 6; a++) {
            for (int c = 0; c < 6; c++) M[a*6+c] += (T)Ar[r][a] * (T)Ar[r][c];     // A^T A (:858)
            g[a] += (T)Ar[r][a] * (T)br[r];                                       // A^T b (:859)
        }
    }
    const float lambda = 0.0001f;                                    // :863-864
    const float l2 = lambda * lambda;
    for (int k = 0; k < 6; k++) M[k*6+k] += (T)l2;
    T x[6];
    fullpiv_lu_solve<T>(M, g, 6, x);                                 // :866-868
    if (x_out) for (int k = 0; k < 6; k++) x_out[k] = (double)x[k];
    float at[3] = {(float)x[0], (float)x[1], (float)x[2]}, tt[3] = {(float)x[3], (float)x[4], (float)x[5]};
    float tan_theta = std::sqrt(sqnorm3_tree(at));                   // :878
    float a[3] = {at[0] / tan_theta, at[1] / tan_theta, at[2] / tan_theta};                   // :879
    float sin_theta = (float)((double)tan_theta / std::sqrt(1.0 + (double)(tan_theta * tan_theta)));   // :884 (1.0 is double)
    float cos_theta = sin_theta / tan_theta;                        // :885
    float t[3] = {tt[0] * cos_theta, tt[1] * cos_theta, tt[2] * cos_theta};                   // :887
    // getRodriguesMatrix, utils.h:171-176 : I + sin*K + (1-cos)*K*K
    float K[9] = {0,-a[2],a[1], a[2],0,-a[0], -a[1],a[0],0}, KK[9], Rod[9];
    // Eigen evaluates ((1 - cos) * K) * K : scale K first, then the product
    float omc = 1 - cos_theta, Ks[9]; for (int i = 0; i < 9; i++) Ks[i] = omc * K[i];
    mat3_mul(Ks, K, KK);
    for (int i = 0; i < 9; i++) Rod[i] = ((i % 4 == 0) ? 1.f : 0.f) + (sin_theta * K[i] + KK[i]);
    float Tm[16], Tt[16], Ts[16], Rm[16], tmp[16], tmp2[16];
    float zero[3] = {0,0,0}, I3[9] = {1,0,0,0,1,0,0,0,1}, nsm[3] = {-sm[0], -sm[1], -sm[2]};
    set_pose(Tm, I3, dm); set_pose(Tt, I3, t); set_pose(Ts, I3, nsm); set_pose(Rm, Rod, zero);
    mat4_mul(Tm, Rm, tmp); mat4_mul(tmp, Tt, tmp2); mat4_mul(tmp2, Rm, tmp); mat4_mul(tmp, Ts, pose);   // :894-895
    return 0;
}


// ---- exact kd-tree 1-NN (CPU stand-in for the FLANN index the optimizer instantiates) ----------
// NearestNeighbor.h:122-141 builds flann::KDTreeIndexParams(1) and :172-174 queries it with 16 checks:
// approximate and randomised, hence not reproducible.  This tree is EXACT: leaves evaluate the same fp32
// distance as orc_knn3 and candidates are compared lexicographically (d2, index), so the result is
// bit-identical to the brute-force scan; pruning uses a plane-distance bound with a 1e-5 relative safety
// margin (fp32 rounding of d2 is < 4e-7 relative).  Used to check full-size GPU results and as the CPU baseline.
struct KdNode { int left, right; int begin, end; int dim; float split; float lo[3], hi[3]; };
struct KdTree {
    std::vector<KdNode> nodes; std::vector<int> perm; std::vector<float> pts;   // pts: reordered copy, 3 per point
    int m = 0;
};
static int kd_build(KdTree& t, const float* tgt, int begin, int end) {
    KdNode nd; nd.begin = begin; nd.end = end; nd.left = nd.right = -1; nd.dim = 0; nd.split = 0;
    for (int k = 0; k < 3; k++) { nd.lo[k] = std::numeric_limits<float>::infinity(); nd.hi[k] = -std::numeric_limits<float>::infinity(); }
    for (int i = begin; i < end; i++) for (int k = 0; k < 3; k++) { float v = tgt[(size_t)t.perm[i]*3+k]; if (v < nd.lo[k]) nd.lo[k] = v; if (v > nd.hi[k]) nd.hi[k] = v; }
    int id = (int)t.nodes.size(); t.nodes.push_back(nd);
    if (end - begin > 16) {
        int dim = 0; float ext = -1;
        for (int k = 0; k < 3; k++) { float e = nd.hi[k] - nd.lo[k]; if (e > ext) { ext = e; dim = k; } }
        if (ext > 0 && std::isfinite(ext)) {
            int mid = (begin + end) / 2;
            std::nth_element(t.perm.begin() + begin, t.perm.begin() + mid, t.perm.begin() + end,
                             [&](int a, int b) { return tgt[(size_t)a*3+dim] < tgt[(size_t)b*3+dim]; });
            t.nodes[id].dim = dim; t.nodes[id].split = tgt[(size_t)t.perm[mid]*3+dim];
            int l = kd_build(t, tgt, begin, mid); int r = kd_build(t, tgt, mid, end);
            t.nodes[id].left = l; t.nodes[id].right = r;
        }
    }
    return id;
}
static inline double kd_box_d2(const KdNode& nd, const float* p) {   // squared distance to the node's bounding box (fp64, lower bound)
    double s = 0;
    for (int k = 0; k < 3; k++) { double d = 0; if (p[k] < nd.lo[k]) d = (double)nd.lo[k] - p[k]; else if (p[k] > nd.hi[k]) d = (double)p[k] - nd.hi[k]; s += d * d; }
    return s;
}
static void kd_query(const KdTree& t, int id, const float* p, float& best, int& bi) {
    const KdNode& nd = t.nodes[id];
    if (nd.left < 0) {
        for (int i = nd.begin; i < nd.end; i++) {
            const float* q = &t.pts[(size_t)i*3];
            float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
            float dist = (dx*dx + dy*dy) + dz*dz;
            int j = t.perm[i];
            if (dist < best || (dist == best && j < bi && bi >= 0)) { best = dist; bi = j; }
        }
        return;
    }
    const KdNode& L = t.nodes[nd.left]; const KdNode& R = t.nodes[nd.right];
    double dl = kd_box_d2(L, p), dr = kd_box_d2(R, p);
    int first = nd.left, second = nd.right; double d1 = dl, d2 = dr;
    if (dr < dl) { first = nd.right; second = nd.left; d1 = dr; d2 = dl; }
    if (!(d1 * (1.0 - 1e-5) > (double)best)) kd_query(t, first, p, best, bi);
    if (!(d2 * (1.0 - 1e-5) > (double)best)) kd_query(t, second, p, best, bi);
}

}  // namespace

// =====================================================================================
//                                    C ABI (ctypes)
// =====================================================================================
extern "C" {

void orc_normal_matrix(const float* pose, float* N9) { normal_matrix(pose, N9); }

// transformPoints, utils.h:106-118
void orc_transform_points(const float* pts, int n, const float* pose, float* out) {
    for (int i = 0; i < n; i++) xform_point(pose, pts + (size_t)i*3, out + (size_t)i*3);
}
// transformNormals, utils.h:122-133
void orc_transform_normals(const float* nrm, int n, const float* pose, float* out) {
    float N[9]; normal_matrix(pose, N);
    for (int i = 0; i < n; i++) xform_normal(N, nrm + (size_t)i*3, out + (size_t)i*3);
}

// Exact 1-NN, first (lowest index) minimum: NearestNeighbor.h:81-97 semantics with the squared
// distance + squared threshold the optimizer really runs (:181-185).  q: transformed queries.
// d2_out (optional) receives the winning squared distance (FLT_MAX if none).
void orc_knn3(const float* q, int n, const float* tgt, int m, float max_dist, OrcMatch* out, float* d2_out) {
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        const float px = q[(size_t)i*3], py = q[(size_t)i*3+1], pz = q[(size_t)i*3+2];
        float best = std::numeric_limits<float>::max(); int bi = -1;
        for (int j = 0; j < m; j++) {
            float dx = px - tgt[(size_t)j*3], dy = py - tgt[(size_t)j*3+1], dz = pz - tgt[(size_t)j*3+2];
            float dist = (dx*dx + dy*dy) + dz*dz;              // flann::L2<float> order
            if (best > dist) { bi = j; best = dist; }          // strict: first minimum
        }
        if (best <= max_dist) { out[i].idx = bi; out[i].weight = 1.f; } else { out[i].idx = -1; out[i].weight = 0.f; }
        if (d2_out) d2_out[i] = best;
    }
}


// buildIndex + queryMatches through the exact kd-tree; results are bit-identical to orc_knn3.
void* orc_kdtree_build(const float* tgt, int m) {
    KdTree* t = new KdTree(); t->m = m; t->perm.resize(m);
    // non-finite targets can never win the strict-< argmin (their distance is inf/NaN): leave them out of the tree
    int cnt = 0; for (int i = 0; i < m; i++) if (finite3(tgt + (size_t)i*3)) t->perm[cnt++] = i;
    t->perm.resize(cnt);
    if (cnt > 0) kd_build(*t, tgt, 0, cnt);
    t->pts.resize((size_t)cnt*3);
    for (int i = 0; i < cnt; i++) std::memcpy(&t->pts[(size_t)i*3], tgt + (size_t)t->perm[i]*3, 12);
    return t;
}
void orc_kdtree_free(void* h) { delete (KdTree*)h; }
void orc_kdtree_query(const void* h, const float* q, int n, float max_dist, OrcMatch* out, float* d2_out) {
    const KdTree& t = *(const KdTree*)h;
    #pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++) {
        float best = std::numeric_limits<float>::max(); int bi = -1;
        const float* p = q + (size_t)i*3;
        if (!t.nodes.empty() && finite3(p)) kd_query(t, 0, p, best, bi);
        if (bi < 0) best = std::numeric_limits<float>::max();
        if (best <= max_dist) { out[i].idx = bi; out[i].weight = 1.f; } else { out[i].idx = -1; out[i].weight = 0.f; }
        if (d2_out) d2_out[i] = best;
    }
}

// colour feature, NearestNeighbor.h:212-221,245-254 : (color_scale*color_normalize) * float(c)
static inline float color_feat(unsigned char c) { const float cn = 1 / float(255); const float cs = 1; return cs * cn * c; }

void orc_color_features(const unsigned char* rgba, int n, float* out3) {
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) out3[(size_t)i*3+k] = color_feat(rgba[(size_t)i*4+k]);
}

// 6-D (xyz + rgb/255) exact 1-NN, NearestNeighbor.h:209-303 ; alpha ignored
void orc_knn6(const float* q, const unsigned char* qrgba, int n, const float* tgt, const unsigned char* trgba, int m,
              float max_dist, OrcMatch* out, float* d2_out) {
    std::vector<float> tf((size_t)m*3), qf((size_t)n*3);
    orc_color_features(trgba, m, tf.data()); orc_color_features(qrgba, n, qf.data());
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        const float* p = q + (size_t)i*3; const float* pc = qf.data() + (size_t)i*3;
        float best = std::numeric_limits<float>::max(); int bi = -1;
        for (int j = 0; j < m; j++) {
            const float* t = tgt + (size_t)j*3; const float* tc = tf.data() + (size_t)j*3;
            float d0 = p[0]-t[0], d1 = p[1]-t[1], d2 = p[2]-t[2], d3 = pc[0]-tc[0], d4 = pc[1]-tc[1], d5 = pc[2]-tc[2];
            float dist = ((((d0*d0 + d1*d1) + d2*d2) + d3*d3) + d4*d4) + d5*d5;   // flann::L2 : 4-unrolled head then tail
            if (best > dist) { bi = j; best = dist; }
        }
        if (best <= max_dist) { out[i].idx = bi; out[i].weight = 1.f; } else { out[i].idx = -1; out[i].weight = 0.f; }
        if (d2_out) d2_out[i] = best;
    }
}

// NearestNeighborSearchProjective::queryMatches, NearestNeighbor.h:333-421.
// Unsigned-underflow quirk (:385-386): a window whose start would be negative never runs =>
// no match.  Projections that are NaN / negative / >= 2^31 are undefined behaviour in the
// reference (float -> unsigned); they are fenced here as "no match" (what x86-64 produces for
// everything below 2^32).  Query x == MINF => value-initialised Match{0, 0.f} (:353,372-373).
void orc_projective(const float* q, int n, const float* tgt, int width, int height, float fx, float fy, float mx, float my,
                    float max_dist, int window, OrcMatch* out, float* d2_out) {
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        const float* p = q + (size_t)i*3;
        if (d2_out) d2_out[i] = std::numeric_limits<float>::max();
        if (p[0] == MINF_) { out[i].idx = 0; out[i].weight = 0.f; continue; }
        float uf = std::round(((p[0] * fx) / p[2]) + mx);          // :378
        float vf = std::round(((p[1] * fy) / p[2]) + my);          // :379
        float best = std::numeric_limits<float>::max(); long long bi = -1;
        bool ok = (uf >= (float)window) && (vf >= (float)window) && (uf < 2147483648.f) && (vf < 2147483648.f);
        if (ok) {
            long long u0 = (long long)uf - window, v0 = (long long)vf - window, u1 = (long long)uf + window, v1 = (long long)vf + window;
            for (long long v = v0; v < height && v <= v1; v++) {
                for (long long u = u0; u < width && u <= u1; u++) {
                    const float* t = tgt + ((size_t)width * v + u) * 3;
                    if (t[0] == MINF_) continue;                  // :392
                    float d[3] = {p[0] - t[0], p[1] - t[1], p[2] - t[2]};
                    float dist = sqnorm3_tree(d);                 // :396 squaredNorm
                    if (best > dist) { bi = (long long)width * v + u; best = dist; }
                }
            }
        }
        if (best <= max_dist) { out[i].idx = (int)bi; out[i].weight = 1.f; } else { out[i].idx = -1; out[i].weight = 0.f; }
        if (d2_out) d2_out[i] = best;
    }
}

// WeightingMethod::applyWeights, weighting.h:39-99.  method: 0 constant,1 distances,2 normals,3 colours
void orc_apply_weights(int method, float max_distance, const float* sp, const float* tp, const float* sn, const float* tn,
                       const unsigned char* sc, const unsigned char* tc, OrcMatch* matches, int n) {
    if (method == 0) return;                                      // :44
    for (int i = 0; i < n; i++) {
        if (matches[i].idx < 0) continue;                         // :51
        const size_t j = (size_t)matches[i].idx;
        float wnew = 0.0f;
        if (method == 1 || method == 3) {
            if (finite3(sp + (size_t)i*3) && finite3(tp + j*3)) {
                float d0 = sp[(size_t)i*3] - tp[j*3], d1 = sp[(size_t)i*3+1] - tp[j*3+1], d2 = sp[(size_t)i*3+2] - tp[j*3+2];
                float dw = (float)(1.0 - (double)(((d0*d0 + d1*d1) + d2*d2) / max_distance));   // :19 (double subtraction)
                wnew += dw;
            }
        }
        if (method == 2) {
            if (finite3(sn + (size_t)i*3) && finite3(tn + j*3)) wnew += dot3_tree(sn + (size_t)i*3, tn + j*3);   // :24
        }
        if (method == 3) {
            unsigned char e0 = (unsigned char)(sc[(size_t)i*4] - tc[j*4]), e1 = (unsigned char)(sc[(size_t)i*4+1] - tc[j*4+1]), e2 = (unsigned char)(sc[(size_t)i*4+2] - tc[j*4+2]);   // :28 uint8 wrap
            float cw = (float)(1.0 - (double)(float(e0*e0 + e1*e1 + e2*e2) / float(195075)));   // :29
            wnew *= cw;
        }
        matches[i].weight = wnew;                                 // :90
    }
}

// ICPOptimizer::pruneCorrespondences, ICPOptimizer.h:157-174
void orc_prune(const float* sn, const float* tn, OrcMatch* matches, int n) {
    const double threshold = 60 * 3.141592653589793238462643383279502884 / 180.0;    // :161 EIGEN_PI
    for (int i = 0; i < n; i++) {
        if (matches[i].idx < 0) continue;
        const float* a = sn + (size_t)i*3; const float* b = tn + (size_t)matches[i].idx*3;
        float c = dot3_tree(a, b) / (std::sqrt(sqnorm3_tree(a)) * std::sqrt(sqnorm3_tree(b)));   // :170
        if ((double)std::acos(c) > threshold) matches[i].idx = -1;           // NaN => kept
    }
}

// The predicate above as a function of the cosine alone (used to derive the device constant).
int orc_prune_predicate(float c) { const double threshold = 60 * 3.141592653589793238462643383279502884 / 180.0; return (double)std::acos(c) > threshold ? 1 : 0; }

// Compaction, ICPOptimizer.h:594-610.  Returns count; arrays sized n by the caller.
int orc_compact(const float* sp, const float* sn, const float* tp, const float* tn, const OrcMatch* matches, int n,
                float* cs, float* cd, float* cw, float* cnt, float* cns) {
    int k = 0;
    for (int j = 0; j < n; j++) {
        if (matches[j].idx < 0) continue;
        const size_t t = (size_t)matches[j].idx;
        if (!finite3(sp + (size_t)j*3) || !finite3(tp + t*3)) continue;
        std::memcpy(cs + (size_t)k*3, sp + (size_t)j*3, 12); std::memcpy(cd + (size_t)k*3, tp + t*3, 12);
        cw[k] = matches[j].weight;
        if (cnt) std::memcpy(cnt + (size_t)k*3, tn + t*3, 12);
        if (cns) std::memcpy(cns + (size_t)k*3, sn + (size_t)j*3, 12);
        k++;
    }
    return k;
}

int orc_solve_p2p(const float* s, const float* d, const float* w, int n, int mode, float* pose) {
    return mode ? solve_p2p_t<double>(s, d, w, (size_t)n, pose) : solve_p2p_t<float>(s, d, w, (size_t)n, pose);
}
int orc_solve_p2plane(const float* s, const float* d, const float* nt, const float* w, int n, int mode, float* pose, double* x6) {
    return mode ? solve_p2plane_t<double>(s, d, nt, w, (size_t)n, pose, x6) : solve_p2plane_t<float>(s, d, nt, w, (size_t)n, pose, x6);
}
int orc_solve_symmetric(const float* s, const float* d, const float* ns, const float* nt, const float* w, int n, int mode, float* pose, double* x6) {
    return mode ? solve_symmetric_t<double>(s, d, ns, nt, w, (size_t)n, pose, x6) : solve_symmetric_t<float>(s, d, ns, nt, w, (size_t)n, pose, x6);
}

void orc_mat4_mul(const float* A, const float* B, float* C) { mat4_mul(A, B, C); }

// ConvergenceMeasure::rmseAlignmentError, ConvergenceMeasure.h:50-66 (fp32 running sum, tree squaredNorm)
float orc_rmse(const float* src, const float* ref, int n, const float* pose) {
    int counter = 0; float rmse = 0.0f;
    for (int i = 0; i < n; i++) {
        float t[3]; xform_point(pose, src + (size_t)i*3, t);
        if (finite3(t) && finite3(ref + (size_t)i*3)) {
            float d[3] = {t[0] - ref[(size_t)i*3], t[1] - ref[(size_t)i*3+1], t[2] - ref[(size_t)i*3+2]};
            rmse += sqnorm3_tree(d); counter++;
        }
    }
    rmse /= counter;
    return std::sqrt(rmse);
}

// ConvergenceMeasure::benchmarkError -> calculate_error, ConvergenceMeasure.h:104-151 (pcl::compute3DCentroid accumulates
// in double and the centroid is narrowed to a float PointXYZ :113-114; pcl::euclideanDistance is an fp32 norm; the
// quotient and the running sum are double :117-121).
double orc_benchmark_error(const float* src, const float* ref, int n, const float* pose) {
    std::vector<float> t((size_t)n * 3);
    double c[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) { xform_point(pose, src + (size_t)i*3, &t[(size_t)i*3]); for (int k = 0; k < 3; k++) c[k] += (double)t[(size_t)i*3+k]; }
    float cf[3] = {(float)(c[0] / n), (float)(c[1] / n), (float)(c[2] / n)};
    double error = 0;
    for (int i = 0; i < n; i++) {
        float e[3] = {t[(size_t)i*3] - ref[(size_t)i*3], t[(size_t)i*3+1] - ref[(size_t)i*3+1], t[(size_t)i*3+2] - ref[(size_t)i*3+2]};
        float g[3] = {t[(size_t)i*3] - cf[0], t[(size_t)i*3+1] - cf[1], t[(size_t)i*3+2] - cf[2]};
        double centroid_distance = std::sqrt(sqnorm3_tree(g));
        error += (double)std::sqrt(sqnorm3_tree(e)) / centroid_distance;
    }
    return error / n;
}

// depthExtrinsics.inverse() (PointCloud.h:88-90).  Eigen's fp32 4x4 inverse is not reproducible without Eigen; the contract
// shared with the device library is the fp64 cofactor inverse of the affine [R|t] (column-major fp32 input) rounded once.
void orc_invert_extrinsics(const float* E /* column-major 4x4 */, float* inv12 /* R^-1 row-major (9), then -R^-1 t (3) */) {
    double R[9], t[3];
    for (int r = 0; r < 3; r++) { for (int k = 0; k < 3; k++) R[r*3+k] = E[k*4+r]; t[r] = E[12+r]; }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    double Ri[9] = {(R[4] * R[8] - R[5] * R[7]) / det, (R[2] * R[7] - R[1] * R[8]) / det, (R[1] * R[5] - R[2] * R[4]) / det,
                    (R[5] * R[6] - R[3] * R[8]) / det, (R[0] * R[8] - R[2] * R[6]) / det, (R[2] * R[3] - R[0] * R[5]) / det,
                    (R[3] * R[7] - R[4] * R[6]) / det, (R[1] * R[6] - R[0] * R[7]) / det, (R[0] * R[4] - R[1] * R[3]) / det};
    for (int i = 0; i < 9; i++) inv12[i] = (float)Ri[i];
    for (int r = 0; r < 3; r++) inv12[9 + r] = (float)(-(Ri[r*3] * t[0] + Ri[r*3+1] * t[1] + Ri[r*3+2] * t[2]));
}

// PointCloud(float* depthMap, BYTE* colorFrame, K, extrinsics, width, height, keepOriginalSize = true), PointCloud.h:78-165.
// inv: 3x3 row-major inverse rotation followed by the inverse translation (12 floats), supplied by the caller.
void orc_backproject(const float* depth, const unsigned char* rgbx, int width, int height, float fx, float fy, float cx, float cy, const float* inv,
                     float max_distance, int fix_color_index, float* xyz, float* nrm, unsigned char* rgba, unsigned char* valid) {
    const float half = max_distance / 2.f;                          // :85
    const int n = width * height;
    for (int v = 0; v < height; v++) for (int u = 0; u < width; u++) {
        const int idx = v * width + u;
        float* p = xyz + (size_t)idx*3; float* q = nrm + (size_t)idx*3;
        const float d = depth[idx];
        if (d == MINF_) { p[0] = p[1] = p[2] = MINF_; }                // :104-106
        else {
            float c[3] = {(u - cx) / fx * d, (v - cy) / fy * d, d};    // :109
            for (int r = 0; r < 3; r++) p[r] = (inv[r*3] * c[0] + (inv[r*3+1] * c[1] + inv[r*3+2] * c[2])) + inv[9 + r];
        }
        q[0] = q[1] = q[2] = MINF_;
        if (v >= 1 && v < height - 1 && u >= 1 && u < width - 1) {     // :117-131
            const float du = 0.5f * (depth[idx + 1] - depth[idx - 1]);
            const float dv = 0.5f * (depth[idx + width] - depth[idx - width]);
            if (std::isfinite(du) && std::isfinite(dv) && !(std::fabs(du) > half) && !(std::fabs(dv) > half)) {
                float nn[3] = {-du, -dv, 1.f};
                const float len = std::sqrt(sqnorm3_tree(nn));        // normalize(), :129
                q[0] = nn[0] / len; q[1] = nn[1] / len; q[2] = nn[2] / len;
            }
        }
        if (rgba && rgbx) {
            const size_t base = fix_color_index ? (size_t)idx * 4 : (size_t)idx;      // :156-157 reads colorFrame[i .. i+3]
            const size_t last = (size_t)n * 4 - 1;
            for (int k = 0; k < 4; k++) rgba[(size_t)idx*4 + k] = rgbx[base + k <= last ? base + k : last];
        }
        if (valid) valid[idx] = (finite3(p) && finite3(q)) ? 1 : 0;     // :152
    }
}

// PointCloud::getCoarseResolution, PointCloud.h:325-343 : stride decimation keeping finite pts+normals.
int orc_coarse(const float* pts, const float* nrm, const unsigned char* rgba, int n, int factor, float* opts, float* onrm, unsigned char* orgba, int* oidx) {
    int k = 0;
    for (int i = 0; i < n; i += factor) {
        if (finite3(pts + (size_t)i*3) && finite3(nrm + (size_t)i*3)) {
            if (opts) std::memcpy(opts + (size_t)k*3, pts + (size_t)i*3, 12);
            if (onrm) std::memcpy(onrm + (size_t)k*3, nrm + (size_t)i*3, 12);
            if (orgba && rgba) std::memcpy(orgba + (size_t)k*4, rgba + (size_t)i*4, 4);
            if (oidx) oidx[k] = i;
            k++;
        }
    }
    return k;
}

struct OrcParams {
    int metric;          // 0 p2p, 1 p2plane, 2 symmetric          ICPOptimizer.h:46-48
    int matching;        // 0 k-NN, 1 projective                   :71-78
    int weighting;       // 0..3                                    weighting.h:8
    int rejection;       // 1 = normals angle (default)             ICPOptimizer.h:30,63-65
    int color_icp;       // 6-D k-NN                                 :54-56
    int multires;        // :50-52
    int n_iterations;    // :84-86
    int solver_mode;     // 0 faithful fp32 sums, 1 fp64 sums (oracle knob, not in the reference)
    float max_distance;  // squared metres                           :41-44
    float fx, fy, cx, cy; int width, height;                      // setCameraParams
    int window;          // 12                                       NearestNeighbor.h:319
    int knn_kdtree;      // oracle knob: 1 = exact kd-tree matcher (same results as the brute-force scan, faster)
    void* kdtree;        // cached tree handle (built by orc_estimate_pose / the caller)
    int selection;       // 0 SELECT_ALL, 1 RANDOM_SAMPLING                      selection.h:9, ICPOptimizer.h:58-61
    float selection_proba;
    unsigned selection_seed;   // the reference seeds mt19937 from random_device (selection.h:76-79); see orc_select_hash
};

struct OrcIterRecord { int n_src; int n_valid; float pose[16]; double seconds_match; double seconds_rest; };

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// One ICP iteration on (possibly decimated) source arrays: stages 2-5 of ICPOptimizer.h:553-621.
// matches_out (n entries) receives the matches after weighting and pruning.
int orc_iterate(const OrcParams* prm, const float* sp0, const float* sn0, const unsigned char* sc, int n,
                const float* tp, const float* tn, const unsigned char* tc, int m, float* pose, OrcMatch* matches_out,
                int* n_valid_out, double* t_match, double* t_rest) {
    std::vector<float> sp((size_t)n*3), sn((size_t)n*3);
    orc_transform_points(sp0, n, pose, sp.data());                 // :553
    orc_transform_normals(sn0, n, pose, sn.data());                // :554
    std::vector<OrcMatch> mt(n);
    double t0 = now_s();
    if (prm->matching == 1) orc_projective(sp.data(), n, tp, prm->width, prm->height, prm->fx, prm->fy, prm->cx, prm->cy, prm->max_distance, prm->window, mt.data(), nullptr);
    else if (prm->color_icp) orc_knn6(sp.data(), sc, n, tp, tc, m, prm->max_distance, mt.data(), nullptr);   // :562-563
    else if (prm->knn_kdtree && prm->kdtree) orc_kdtree_query(prm->kdtree, sp.data(), n, prm->max_distance, mt.data(), nullptr);
    else orc_knn3(sp.data(), n, tp, m, prm->max_distance, mt.data(), nullptr);                                // :565
    double t1 = now_s();
    orc_apply_weights(prm->weighting, prm->max_distance, sp.data(), tp, sn.data(), tn, sc, tc, mt.data(), n);   // :571
    if (prm->rejection == 1) orc_prune(sn.data(), tn, mt.data(), n);                                           // :578-579
    std::vector<float> cs((size_t)n*3), cd((size_t)n*3), cw(n), cnt((size_t)n*3), cns((size_t)n*3);
    int k = orc_compact(sp.data(), sn.data(), tp, tn, mt.data(), n, cs.data(), cd.data(), cw.data(), cnt.data(), cns.data());
    if (matches_out) std::memcpy(matches_out, mt.data(), sizeof(OrcMatch) * (size_t)n);
    if (n_valid_out) *n_valid_out = k;
    float dT[16]; int rc;
    if (prm->metric == 1) rc = orc_solve_p2plane(cs.data(), cd.data(), cnt.data(), cw.data(), k, prm->solver_mode, dT, nullptr);
    else if (prm->metric == 0) rc = orc_solve_p2p(cs.data(), cd.data(), cw.data(), k, prm->solver_mode, dT);
    else rc = orc_solve_symmetric(cs.data(), cd.data(), cns.data(), cnt.data(), cw.data(), k, prm->solver_mode, dT, nullptr);
    if (rc) return rc;
    mat4_mul(dT, pose, pose);                                     // :614-620  pose <- dT * pose
    double t2 = now_s();
    if (t_match) *t_match = t1 - t0;
    if (t_rest) *t_rest = (t2 - t1) + 0.0;
    return 0;
}

// RANDOM_SAMPLING predicate.  selection.h:88-106 keeps point i iff ureal(rng) < prob with an mt19937 seeded from
// random_device -- irreproducible by construction.  The contract shared with the device library is a counter-based hash
// of (seed, resample number, original point index): kept iff hash < prob * 2^32.  Same distribution, reproducible.
static inline unsigned orc_fmix32(unsigned h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
unsigned orc_select_hash(unsigned seed, unsigned iteration, unsigned index) {
    return orc_fmix32(index * 0x9E3779B9u + orc_fmix32(seed + iteration * 0x7F4A7C15u + 0x165667B1u));
}

// LinearICPOptimizer::estimatePose, ICPOptimizer.h:493-663 (SELECT_ALL only; RANDOM_SAMPLING is seeded
// from random_device in the reference, selection.h:76-79, and is therefore out of parity scope).
// records: capacity max_records; returns number of iterations run, or -1 on "no correspondences".
int orc_estimate_pose(const OrcParams* prm, const float* sp, const float* sn, const unsigned char* sc, int n,
                      const float* tp, const float* tn, const unsigned char* tc, int m, float* pose,
                      OrcIterRecord* records, int max_records) {
    OrcParams local = *prm; void* owned = nullptr;
    if (local.knn_kdtree && !local.kdtree && local.matching == 0 && !local.color_icp) { owned = orc_kdtree_build(tp, m); local.kdtree = owned; }   // buildIndex :532-535
    prm = &local;
    struct Guard { void* h; ~Guard() { if (h) orc_kdtree_free(h); } } guard{owned};
    float currentResolution = 1.0f; int originalSize = n;
    if (prm->multires) {                                          // :505-516
        while (1) { originalSize = (int)(originalSize / 2.0); if (originalSize < 100) break; currentResolution *= 2.0f; }
    }
    std::vector<float> cp, cn; std::vector<unsigned char> cc; std::vector<int> cidx;
    const float* P0 = sp; const float* N0 = sn; const unsigned char* C0 = sc; int cur_n = n;
    auto decimate = [&](int factor) {
        cp.assign((size_t)n*3, 0.f); cn.assign((size_t)n*3, 0.f); cc.assign((size_t)n*4, 0); cidx.assign((size_t)n, 0);
        cur_n = orc_coarse(sp, sn, sc, n, factor, cp.data(), cn.data(), sc ? cc.data() : nullptr, cidx.data());
        P0 = cp.data(); N0 = cn.data(); C0 = sc ? cc.data() : nullptr;
    };
    std::vector<float> rp, rn; std::vector<unsigned char> rc8;
    if (prm->multires) decimate((int)currentResolution);          // :520-523
    int it = 0;
    for (int i = 0; i < prm->n_iterations || prm->multires; ++i) {   // :540
        int nv = 0; double tm = 0, tr = 0;
        const float* Pi = P0; const float* Ni = N0; const unsigned char* Ci = C0; int ni = cur_n;
        if (prm->selection == 1) {                               // sourceSelection.resample(), :549-550 / selection.h:88-106
            double th = (double)prm->selection_proba * 4294967296.0;
            rp.clear(); rn.clear(); rc8.clear();
            for (int k = 0; k < cur_n; k++) {
                const unsigned orig = (P0 == sp) ? (unsigned)k : (unsigned)cidx[k];
                if (th >= 4294967296.0 || (th > 0.0 && orc_select_hash(prm->selection_seed, (unsigned)i, orig) < (unsigned)th)) {
                    rp.insert(rp.end(), P0 + (size_t)k*3, P0 + (size_t)k*3 + 3); rn.insert(rn.end(), N0 + (size_t)k*3, N0 + (size_t)k*3 + 3);
                    if (C0) rc8.insert(rc8.end(), C0 + (size_t)k*4, C0 + (size_t)k*4 + 4);
                }
            }
            ni = (int)(rp.size() / 3); Pi = rp.data(); Ni = rn.data(); Ci = C0 ? rc8.data() : nullptr;
        }
        int rc = ni > 0 ? orc_iterate(prm, Pi, Ni, Ci, ni, tp, tn, tc, m, pose, nullptr, &nv, &tm, &tr) : 1;
        if (rc) return -1;
        if (records && it < max_records) { records[it].n_src = ni; records[it].n_valid = nv; std::memcpy(records[it].pose, pose, 64); records[it].seconds_match = tm; records[it].seconds_rest = tr; }
        it++;
        if (prm->multires) {                                      // :634-655
            if (currentResolution == 1.0f && i >= prm->n_iterations - 1) break;
            if (currentResolution == 1.0f) continue;
            currentResolution /= 2.0f; if (currentResolution < 1.0f) currentResolution = 1.0f;
            decimate((int)currentResolution);
        }
    }
    return it;
}

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}

}  // extern "C"

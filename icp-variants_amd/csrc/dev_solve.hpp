// dev_solve.hpp -- fp64 small dense solvers and k_reduce_solve.
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.
// ------------------------------------------------------------------------------------------------
// fp64 small dense solvers, run by one thread of k_reduce_solve.
template <int n>
__device__ inline void jacobi_eig_sym(double* A /* n x n, destroyed */, double* V, double* ev) {
#pragma unroll
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 50; sweep++) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int j = 0; j < n; j++) { if (i != j) off += A[i * n + j] * A[i * n + j]; else dg += A[i * n + j] * A[i * n + j]; }
        }
        if (off <= 1e-300 || off <= 1e-34 * dg) break;
#pragma unroll
        for (int p = 0; p < n - 1; p++) {
#pragma unroll
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq != 0.0) {
                    const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < n; k++) { const double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq; }
#pragma unroll
                    for (int k = 0; k < n; k++) { const double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk; }
#pragma unroll
                    for (int k = 0; k < n; k++) { const double vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq; }
                    // the rotation annihilates a_pq by construction; writing the exact zero (instead of the ~1e-16 |A| the formulas
                    // leave) lets the sweep loop see convergence -- otherwise all 50 sweeps run (57 us for a 3x3 on one lane)
                    A[p * n + q] = 0.0; A[q * n + p] = 0.0;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < n; i++) ev[i] = A[i * n + i];
}

// Fast path: when the 6x6 normal matrix is comfortably full rank (every LDL^T pivot > 1e-9 x its diagonal entry,
// i.e. far above the (6 eps_f32)^2 = 5e-13 relative eigenvalue cut of the SVD rule) the truncated-SVD solution IS the
// plain solution and an unrolled fp64 LDL^T gives it in ~100 flops.  Otherwise: Jacobi eigen-decomposition.
__device__ __forceinline__ bool solve_ldlt6(const double* sums, double* x) {
    double a00 = sums[0], a01 = sums[1], a02 = sums[2], a03 = sums[3], a04 = sums[4], a05 = sums[5];
    double a11 = sums[6], a12 = sums[7], a13 = sums[8], a14 = sums[9], a15 = sums[10];
    double a22 = sums[11], a23 = sums[12], a24 = sums[13], a25 = sums[14];
    double a33 = sums[15], a34 = sums[16], a35 = sums[17];
    double a44 = sums[18], a45 = sums[19];
    double a55 = sums[20];
    const double g0 = sums[21], g1 = sums[22], g2 = sums[23], g3 = sums[24], g4 = sums[25], g5 = sums[26];
    const double tol = 1e-9;
    const double o00 = a00, o11 = a11, o22 = a22, o33 = a33, o44 = a44, o55 = a55;
    // column 0
    const double d0 = a00; if (!(d0 > tol * o00) || !(o00 > 0)) return false;
    const double l10 = a01 / d0, l20 = a02 / d0, l30 = a03 / d0, l40 = a04 / d0, l50 = a05 / d0;
    a11 -= l10 * a01; a12 -= l10 * a02; a13 -= l10 * a03; a14 -= l10 * a04; a15 -= l10 * a05;
    a22 -= l20 * a02; a23 -= l20 * a03; a24 -= l20 * a04; a25 -= l20 * a05;
    a33 -= l30 * a03; a34 -= l30 * a04; a35 -= l30 * a05;
    a44 -= l40 * a04; a45 -= l40 * a05;
    a55 -= l50 * a05;
    const double d1 = a11; if (!(d1 > tol * o11)) return false;
    const double l21 = a12 / d1, l31 = a13 / d1, l41 = a14 / d1, l51 = a15 / d1;
    a22 -= l21 * a12; a23 -= l21 * a13; a24 -= l21 * a14; a25 -= l21 * a15;
    a33 -= l31 * a13; a34 -= l31 * a14; a35 -= l31 * a15;
    a44 -= l41 * a14; a45 -= l41 * a15;
    a55 -= l51 * a15;
    const double d2 = a22; if (!(d2 > tol * o22)) return false;
    const double l32 = a23 / d2, l42 = a24 / d2, l52 = a25 / d2;
    a33 -= l32 * a23; a34 -= l32 * a24; a35 -= l32 * a25;
    a44 -= l42 * a24; a45 -= l42 * a25;
    a55 -= l52 * a25;
    const double d3 = a33; if (!(d3 > tol * o33)) return false;
    const double l43 = a34 / d3, l53 = a35 / d3;
    a44 -= l43 * a34; a45 -= l43 * a35;
    a55 -= l53 * a35;
    const double d4 = a44; if (!(d4 > tol * o44)) return false;
    const double l54 = a45 / d4;
    a55 -= l54 * a45;
    const double d5 = a55; if (!(d5 > tol * o55)) return false;
    // L z = g
    const double z0 = g0;
    const double z1 = g1 - l10 * z0;
    const double z2 = g2 - l20 * z0 - l21 * z1;
    const double z3 = g3 - l30 * z0 - l31 * z1 - l32 * z2;
    const double z4 = g4 - l40 * z0 - l41 * z1 - l42 * z2 - l43 * z3;
    const double z5 = g5 - l50 * z0 - l51 * z1 - l52 * z2 - l53 * z3 - l54 * z4;
    // D y = z ; L^T x = y
    const double x5 = z5 / d5;
    const double x4 = z4 / d4 - l54 * x5;
    const double x3 = z3 / d3 - l43 * x4 - l53 * x5;
    const double x2 = z2 / d2 - l32 * x3 - l42 * x4 - l52 * x5;
    const double x1 = z1 / d1 - l21 * x2 - l31 * x3 - l41 * x4 - l51 * x5;
    const double x0 = z0 / d0 - l10 * x1 - l20 * x2 - l30 * x3 - l40 * x4 - l50 * x5;
    x[0] = x0; x[1] = x1; x[2] = x2; x[3] = x3; x[4] = x4; x[5] = x5;
    return true;
}

__device__ inline void solve_normal_svd(const double* sums /* 21 + 6 */, double* x) {
    if (solve_ldlt6(sums, x)) return;
    // workspaces in LDS, not in registers / scratch: this runs on ONE thread, and inside the fused matcher it must not raise the
    // kernel's register or scratch footprint (a dispatch with a large scratch demand stalls the queue)
    __shared__ double A[36], V[36], ev[6];
    int q = 0;
    for (int a = 0; a < 6; a++) for (int c = a; c < 6; c++) { A[a * 6 + c] = sums[q]; A[c * 6 + a] = sums[q]; q++; }
    const double* g = sums + 21;
    jacobi_eig_sym<6>(A, V, ev);
    double emax = 0.0;
    for (int i = 0; i < 6; i++) emax = fmax(emax, ev[i]);
    const double thr = 6.0 * 1.1920928955078125e-07;
    for (int i = 0; i < 6; i++) x[i] = 0.0;
    for (int j = 0; j < 6; j++) {
        if (!(ev[j] > thr * thr * emax)) continue;
        double vg = 0.0;
        for (int i = 0; i < 6; i++) vg += V[i * 6 + j] * g[i];
        const double coef = vg / ev[j];
        for (int i = 0; i < 6; i++) x[i] += V[i * 6 + j] * coef;
    }
}

// FullPivLU::solve with its rank rule in fp64 (ICPOptimizer.h:866-868).
// M, rhs, x, colp and y are caller-provided workspaces (LDS in k_reduce_solve: dynamically indexed local arrays would live in
// scratch memory, two orders of magnitude slower per access for this single-thread code).
__device__ inline void solve_fullpiv_lu6(double* M, double* rhs, double* x, int* colp, double* y) {
    const int n = 6;
    for (int i = 0; i < n; i++) colp[i] = i;
    double maxpiv = 0.0; int rank = n;
    for (int k = 0; k < n; k++) {
        int pr = k, pc = k; double best = -1.0;
        for (int i = k; i < n; i++) for (int j = k; j < n; j++) { const double v = fabs(M[i * n + j]); if (v > best) { best = v; pr = i; pc = j; } }
        if (best > maxpiv) maxpiv = best;
        if (best == 0.0) { rank = k; break; }
        if (pr != k) { for (int j = 0; j < n; j++) { const double t = M[k * n + j]; M[k * n + j] = M[pr * n + j]; M[pr * n + j] = t; } const double t = rhs[k]; rhs[k] = rhs[pr]; rhs[pr] = t; }
        if (pc != k) { for (int i = 0; i < n; i++) { const double t = M[i * n + k]; M[i * n + k] = M[i * n + pc]; M[i * n + pc] = t; } const int t = colp[k]; colp[k] = colp[pc]; colp[pc] = t; }
        for (int i = k + 1; i < n; i++) {
            const double f = M[i * n + k] / M[k * n + k];
            for (int j = k + 1; j < n; j++) M[i * n + j] -= f * M[k * n + j];
            rhs[i] -= f * rhs[k];
        }
    }
    const double thr = 1.1920928955078125e-07 * 6.0;
    int r = 0;
    for (int k = 0; k < rank; k++) { if (fabs(M[k * n + k]) > maxpiv * thr) r++; else break; }
    for (int k = 0; k < n; k++) y[k] = 0.0;
    for (int k = r - 1; k >= 0; k--) { double s = rhs[k]; for (int j = k + 1; j < r; j++) s -= M[k * n + j] * y[j]; y[k] = s / M[k * n + k]; }
    for (int k = 0; k < n; k++) x[colp[k]] = (k < r) ? y[k] : 0.0;
}

// Rotation of the weighted Procrustes problem: R = U diag(1,1,det(UV^T)) V^T of A = U S V^T
// (ProcrustesAligner.h:55-64).  V from the fp64 eigen-decomposition of A^T A (descending), U_c = A v_c/|A v_c|
// for the two leading columns.  With c = U_0 x U_1 the reference's product collapses to
//   R = U_0 V_0^T + U_1 V_1^T + det(V) * c * V_2^T
// (flipping the sign of the third left vector flips det(UV^T) too), which stays well defined when sigma_3 -> 0.
__device__ inline void procrustes_rotation(const double* A /* 3x3 row-major */, double* R) {
    __shared__ double B[9], V[9], ev[3], Vs[9], U[9];     // LDS workspaces (single thread): see solve_normal_svd
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[k * 3 + i] * A[k * 3 + j]; B[i * 3 + j] = s; }
    jacobi_eig_sym<3>(B, V, ev);
    int o[3] = {0, 1, 2};
    for (int a = 0; a < 2; a++) for (int b = a + 1; b < 3; b++) if (ev[o[b]] > ev[o[a]]) { const int t = o[a]; o[a] = o[b]; o[b] = t; }
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) Vs[r * 3 + c] = V[r * 3 + o[c]];
    const double s0 = sqrt(fmax(ev[o[0]], 0.0));
    for (int c = 0; c < 2; c++) {
        double u[3];
        for (int r = 0; r < 3; r++) { u[r] = 0; for (int k = 0; k < 3; k++) u[r] += A[r * 3 + k] * Vs[k * 3 + c]; }
        if (c == 1) {   // re-orthogonalise against column 0 (exact in exact arithmetic)
            const double dp = u[0] * U[0] + u[1] * U[3] + u[2] * U[6];
            u[0] -= dp * U[0]; u[1] -= dp * U[3]; u[2] -= dp * U[6];
        }
        double nr = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (!(nr > 1e-13 * s0) || !(nr > 0.0)) {   // rank-deficient A: any unit vector orthogonal to what we have
            if (c == 0) { u[0] = 1; u[1] = 0; u[2] = 0; }
            else {
                const double a0 = U[0], a1 = U[3], a2 = U[6];
                const int kmin = fabs(a0) < fabs(a1) ? (fabs(a0) < fabs(a2) ? 0 : 2) : (fabs(a1) < fabs(a2) ? 1 : 2);
                const double dp = (kmin == 0 ? a0 : (kmin == 1 ? a1 : a2));
                u[0] = (kmin == 0 ? 1.0 : 0.0) - dp * a0; u[1] = (kmin == 1 ? 1.0 : 0.0) - dp * a1; u[2] = (kmin == 2 ? 1.0 : 0.0) - dp * a2;
            }
            nr = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        }
        for (int r = 0; r < 3; r++) U[r * 3 + c] = u[r] / nr;
    }
    U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
    const double detV = Vs[0] * (Vs[4] * Vs[8] - Vs[5] * Vs[7]) - Vs[1] * (Vs[3] * Vs[8] - Vs[5] * Vs[6]) + Vs[2] * (Vs[3] * Vs[7] - Vs[4] * Vs[6]);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        R[r * 3 + c] = (U[r * 3] * Vs[c * 3] + U[r * 3 + 1] * Vs[c * 3 + 1]) + detV * U[r * 3 + 2] * Vs[c * 3 + 2];
}

// fp32 helpers with the Eigen fixed-size evaluation orders used by the reference's pose algebra
__device__ inline void mat4_mul_f32(const float* A, const float* B, float* C) {   // column-major, sequential over k
    float T[16];
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) {
        float acc = A[0 * 4 + r] * B[c * 4 + 0];
        acc = acc + A[1 * 4 + r] * B[c * 4 + 1];
        acc = acc + A[2 * 4 + r] * B[c * 4 + 2];
        acc = acc + A[3 * 4 + r] * B[c * 4 + 3];
        T[c * 4 + r] = acc;
    }
    for (int i = 0; i < 16; i++) C[i] = T[i];
}
__device__ inline void mat3_mul_f32(const float* A, const float* B, float* C) {   // row-major, e0 + (e1 + e2)
    float T[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) T[r * 3 + c] = A[r * 3] * B[c] + (A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c]);
    for (int i = 0; i < 9; i++) C[i] = T[i];
}
__device__ inline void set_pose_f32(float* pose, const float* R, const float* t) {
    for (int i = 0; i < 16; i++) pose[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) pose[c * 4 + r] = R[r * 3 + c]; pose[12 + r] = t[r]; }
}
// (R^-1)^T by fp64 cofactors rounded once (same operation order as the oracle's normal_matrix)
__device__ __host__ inline void normal_matrix_from_pose(const float* pose, float* N) {
    const double a = pose[0], b = pose[4], c = pose[8];
    const double d = pose[1], e = pose[5], f = pose[9];
    const double g = pose[2], h = pose[6], i = pose[10];
    const double c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const double c10 = c * h - b * i, c11 = a * i - c * g, c12 = b * g - a * h;
    const double c20 = b * f - c * e, c21 = c * d - a * f, c22 = a * e - b * d;
    const double det = (a * c00 + b * c01) + c * c02;
    N[0] = (float)(c00 / det); N[1] = (float)(c01 / det); N[2] = (float)(c02 / det);
    N[3] = (float)(c10 / det); N[4] = (float)(c11 / det); N[5] = (float)(c12 / det);
    N[6] = (float)(c20 / det); N[7] = (float)(c21 / det); N[8] = (float)(c22 / det);
}

// one entry of the same matrix (entry e = 3 row + column): the same operations on the same values as normal_matrix_from_pose, so the
// nine lanes that each compute one give the nine floats it gives
__device__ inline float normal_matrix_entry(const float* pose, int e) {
    const double a = pose[0], b = pose[4], c = pose[8];
    const double d = pose[1], e_ = pose[5], f = pose[9];
    const double g = pose[2], h = pose[6], i = pose[10];
    const double c00 = e_ * i - f * h, c01 = f * g - d * i, c02 = d * h - e_ * g;
    const double det = (a * c00 + b * c01) + c * c02;
    double cof;
    switch (e) {
        case 0: cof = c00; break;  case 1: cof = c01; break;  case 2: cof = c02; break;
        case 3: cof = c * h - b * i; break;  case 4: cof = a * i - c * g; break;  case 5: cof = b * g - a * h; break;
        case 6: cof = b * f - c * e_; break;  case 7: cof = c * d - a * f; break;  default: cof = a * e_ - b * d; break;
    }
    return (float)(cof / det);
}

struct SolveParams {
    const double* partials; int nblocks;   // [NSUM][nblocks]
    double* totals; unsigned* ticket;      // NSUM reduced sums; arrival counter (0 between launches)
    PoseState* ps;
    int metric; int phase;         // phase 0: full solve (p2p / p2plane) or means only (symmetric); phase 1: symmetric solve
    icp_iter_stats* stats;         // record slot of this iteration (may be null)
    int n_src;
    double* sums_out;              // optional copy of the reduced sums (NSUM doubles)
    int update_pose;               // 0: only reduce (icp_correspond)
    const double* rmse_partials; int rmse_blocks;   // unused here
    int spin;                      // 1: block 0 waits for the other blocks' totals (self-validating 8-byte values) instead of the ticket hand-over
};

// Point-to-plane fast path of k_reduce_solve on the lanes of the (last) block instead of one thread: the same LDL^T recurrences
// as solve_ldlt6 (every element sees the same operations in the same order, so the result is bit-identical to it), the three
// sincos on three lanes, the pose product on sixteen.  A single lane runs ~13 cycles per dependent fp64 operation with nothing
// to overlap; this cuts the serial tail of every iteration from ~5 us to ~2 us.
// p2plane_lanes_core: m = the 27 point-to-plane sums (shared memory), pose_in = the pose the iteration searched at (16 floats, global or
// shared).  Returns the composed pose dT * pose_in (16 floats in shared memory, valid for every thread after the call) or nullptr when
// a pivot fails the rank test (nothing computed).  All threads of the block must call it (>= 64 threads).
// One wave does it: its LDS operations execute in program order, so the hand-overs between its lanes need no workgroup barrier -- only
// the compiler must keep the order (wave_sync).  Measured in the merged launch (tools/dev_ring_times.py): 2.8 us with twelve block barriers,
// six sequential divisions in the back substitution and the rotation composed by one thread.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ const float* p2plane_lanes_core(const double* m /* shared */, const float* pose_in) {
    __shared__ double A[6][7], Lm[6][6], od[6], zs[6];
    __shared__ float npose[16];
    __shared__ int okflag;
    const int tid = threadIdx.x;
    if (tid < WAVE) {
        const int i = tid / 7, j = tid % 7;
        if (tid < 42) A[i][j] = (j == 6) ? m[21 + i] : (j >= i ? m[i * 6 - i * (i - 1) / 2 + (j - i)] : 0.0);
        if (tid < 6) od[tid] = m[tid * 6 - tid * (tid - 1) / 2];
        wave_sync();
        bool ok_all = true;
        for (int k = 0; k < 6; k++) {
            const double d = A[k][k];
            const bool ok = (d > 1e-9 * od[k]) && (k > 0 || od[0] > 0);          // solve_ldlt6's pivot tests
            if (!ok) { ok_all = false; break; }                                   // every lane sees the same values
            if (tid < 42 && i > k && j >= i) {
                const double lik = A[k][i] / d;
                if (j == i) Lm[i][k] = lik;
                A[i][j] = A[i][j] - lik * A[k][j];
            }
            wave_sync();
        }
        if (tid == 0) okflag = ok_all ? 1 : 0;
        if (ok_all) {
            if (tid < 6) zs[tid] = A[tid][6] / A[tid][tid];                      // D y = z: the six quotients side by side ...
            wave_sync();
            // ... L^T x = y: the same subtractions in the same order as ever, by EVERY lane for itself (a wave pays per instruction, not per
            // lane: what all lanes hold in registers needs no further trip through LDS -- each of those costs more than the arithmetic
            // between two of them)
            double x[6];
#pragma unroll
            for (int r = 5; r >= 0; r--) {
                double v = zs[r];
#pragma unroll
                for (int c = r + 1; c < 6; c++) v = v - Lm[c][r] * x[c];
                x[r] = v;
            }
            // angles -> sines and cosines: lanes 0..2 take one angle each, the six values travel as scalars (ICPOptimizer.h:768)
            const float ang = (float)(tid == 0 ? x[0] : tid == 1 ? x[1] : x[2]);
            double sd, cd;
            sincos((double)ang, &sd, &cd);
            const float sf = (float)sd, cf = (float)cd;
            const float ca = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cf), 0)), sa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sf), 0));
            const float cb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cf), 1)), sb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sf), 1));
            const float cg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cf), 2)), sg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sf), 2));
            if (tid < 16) {
                // entry (r, c) of dT * pose, dT = [Rx Ry Rz | t] (:771-773, set_pose_f32, mat4_mul_f32): row r of the two 3 x 3 products with
                // mat3_mul_f32's operation order, in registers
                const int c = tid >> 2, r = tid & 3;
                const float one = 1.f, zero = 0.f;
                const float rx0 = r == 0 ? one : zero, rx1 = r == 1 ? ca : r == 2 ? sa : zero, rx2 = r == 1 ? -sa : r == 2 ? ca : zero;      // row r of Rx
                const float ry[9] = {cb, zero, sb, zero, one, zero, -sb, zero, cb}, rz[9] = {cg, -sg, zero, sg, cg, zero, zero, zero, one};
                float q[3], rr[3];
#pragma unroll
                for (int e = 0; e < 3; e++) q[e] = rx0 * ry[e] + (rx1 * ry[3 + e] + rx2 * ry[6 + e]);
#pragma unroll
                for (int e = 0; e < 3; e++) rr[e] = q[0] * rz[e] + (q[1] * rz[3 + e] + q[2] * rz[6 + e]);
                const float tr = (float)(r == 0 ? x[3] : r == 1 ? x[4] : x[5]);
                const float d0 = r < 3 ? rr[0] : zero, d1 = r < 3 ? rr[1] : zero, d2 = r < 3 ? rr[2] : zero, d3 = r < 3 ? tr : one;      // row r of dT
                float acc = d0 * pose_in[c * 4 + 0];                               // mat4_mul_f32(dT, pose): column-major, sequential over k
                acc = acc + d1 * pose_in[c * 4 + 1];
                acc = acc + d2 * pose_in[c * 4 + 2];
                acc = acc + d3 * pose_in[c * 4 + 3];
                npose[tid] = acc;
            }
        }
    }
    __syncthreads();
    if (!okflag) return nullptr;
    return npose;
}
// The k_reduce_solve form: pose state updated in place.  Returns false (nothing written) when the fast path does not apply: other
// metrics, no valid pair, or a pivot fails the rank test -> the caller's single-thread path with the eigen fallback.
__device__ __forceinline__ bool solve_p2plane_lanes(const SolveParams& sp, const double* tot /* shared */) {
    if (!(sp.update_pose && sp.metric == ICP_METRIC_POINT_TO_PLANE && sp.phase == 0 && tot[SUM_N] > 0)) return false;   // uniform
    PoseState* ps = sp.ps;
    const float* npose = p2plane_lanes_core(tot + SUM_M, ps->pose);
    if (!npose) return false;
    const int tid = threadIdx.x;
    if (tid == 0 && sp.sums_out) for (int a = 0; a < NSUM; a++) sp.sums_out[a] = tot[a];
    if (tid >= 64 && tid < 70) {                           // means are not needed for this metric; keep the state defined (k_reduce_solve's phase 0 writes them too): six lanes, one quotient each
        const int k = tid - 64; const double n = tot[SUM_N];
        if (k < 3) ps->mean_s[k] = (float)(tot[SUM_S + k] / n); else ps->mean_d[k - 3] = (float)(tot[SUM_D + k - 3] / n);
    }
    if (tid < 16) { ps->pose[tid] = npose[tid]; if (sp.stats) sp.stats->pose[tid] = npose[tid]; }
    if (tid >= 32 && tid < 41) ps->nmat[tid - 32] = normal_matrix_entry(npose, tid - 32);      // nine lanes, one quotient each
    if (tid == 33 && sp.stats) {
        sp.stats->n_src = sp.n_src; sp.stats->n_valid = (int)tot[SUM_N];
        sp.stats->rmse = -1.f; sp.stats->benchmark_error = -1.f; sp.stats->status = ICP_OK;
    }
    return true;
}

// The single-thread part of the solve (every metric, and point-to-plane when a pivot fails the rank test): thread 0 of the block
// that holds the NSUM totals in `tot` (shared memory).  (A real call would not help the fused matcher, which inlines it too: on
// AMDGPU a kernel reserves the registers of everything it may call -- 254 here.  The matcher caps its own budget instead, so this
// cold code spills there rather than costing the tree walk a wave per SIMD.)
template <bool WITH_SYMMETRIC>
__device__ inline void solve_generic(const SolveParams& sp, const double* tot) {
    if (sp.sums_out) for (int a = 0; a < NSUM; a++) sp.sums_out[a] = tot[a];
    PoseState* ps = sp.ps;
    const double n = tot[SUM_N];
    if (sp.phase == 0) {
        // means of the valid pairs (utils.h:136-145 computes them as fp32 running sums; here fp64 sums rounded once)
        float ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
        if (n > 0) for (int k = 0; k < 3; k++) { ms[k] = (float)(tot[SUM_S + k] / n); md[k] = (float)(tot[SUM_D + k] / n); }
        for (int k = 0; k < 3; k++) { ps->mean_s[k] = ms[k]; ps->mean_d[k] = md[k]; }
    }
    if (!sp.update_pose) return;
    if (sp.metric == ICP_METRIC_SYMMETRIC && sp.phase == 0) return;      // wait for the second pass
    int status = ICP_OK;
    float dT[16];
    for (int i = 0; i < 16; i++) dT[i] = (i % 5 == 0) ? 1.f : 0.f;
    if (!(n > 0)) {
        status = ICP_ERR_NO_CORRESPONDENCES;
    } else if (sp.metric == ICP_METRIC_POINT_TO_PLANE) {
        double x[6];
        solve_normal_svd(tot + SUM_M, x);
        const float al = (float)x[0], be = (float)x[1], ga = (float)x[2];       // ICPOptimizer.h:768
        const float ca = (float)cos((double)al), sa = (float)sin((double)al);
        const float cb = (float)cos((double)be), sb = (float)sin((double)be);
        const float cg = (float)cos((double)ga), sg = (float)sin((double)ga);
        const float Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca}, Ry[9] = {cb, 0, sb, 0, 1, 0, -sb, 0, cb}, Rz[9] = {cg, -sg, 0, sg, cg, 0, 0, 0, 1};
        float Rxy[9], R[9];
        mat3_mul_f32(Rx, Ry, Rxy); mat3_mul_f32(Rxy, Rz, R);                   // :771-773
        const float t[3] = {(float)x[3], (float)x[4], (float)x[5]};
        set_pose_f32(dT, R, t);
    } else if (sp.metric == ICP_METRIC_POINT_TO_POINT) {
        // A = sum_i (d_i - dm)(w_i (s_i - sm))^T expanded in moments (ProcrustesAligner.h:50-55)
        const double* m = tot + SUM_M;
        const float msf[3] = {ps->mean_s[0], ps->mean_s[1], ps->mean_s[2]}, mdf[3] = {ps->mean_d[0], ps->mean_d[1], ps->mean_d[2]};
        __shared__ double A[9], Rd[9];
        for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++)
            A[j * 3 + k] = m[7 + j * 3 + k] - m[4 + j] * (double)msf[k] - (double)mdf[j] * m[1 + k] + m[0] * (double)mdf[j] * (double)msf[k];
        float R[9];
        procrustes_rotation(A, Rd);
        for (int i = 0; i < 9; i++) R[i] = (float)Rd[i];
        const float tr[3] = {mdf[0] - msf[0], mdf[1] - msf[1], mdf[2] - msf[2]};     // ProcrustesAligner.h:70
        float t[3];
        for (int r = 0; r < 3; r++) {
            const float Rt = R[r * 3] * tr[0] + (R[r * 3 + 1] * tr[1] + R[r * 3 + 2] * tr[2]);
            const float Rm = R[r * 3] * mdf[0] + (R[r * 3 + 1] * mdf[1] + R[r * 3 + 2] * mdf[2]);
            t[r] = (Rt - Rm) + mdf[r];                                             // ProcrustesAligner.h:26
        }
        set_pose_f32(dT, R, t);
    } else if constexpr (WITH_SYMMETRIC) {
        // symmetric: M = A^T A + lambda^2 I, FullPivLU (ICPOptimizer.h:858-868)
        __shared__ double M[36], g[6], x[6], ywork[6];
        __shared__ int colp[6];
        int q = 0;
        for (int a = 0; a < 6; a++) for (int c = a; c < 6; c++) { M[a * 6 + c] = tot[SUM_M + q]; M[c * 6 + a] = tot[SUM_M + q]; q++; }
        for (int a = 0; a < 6; a++) g[a] = tot[SUM_M + 21 + a];
        const float lambda = 0.0001f; const float l2 = lambda * lambda;
        for (int a = 0; a < 6; a++) M[a * 6 + a] += (double)l2;
        solve_fullpiv_lu6(M, g, x, colp, ywork);
        const float at[3] = {(float)x[0], (float)x[1], (float)x[2]}, tt[3] = {(float)x[3], (float)x[4], (float)x[5]};
        const float tan_theta = sqrtf(at[0] * at[0] + (at[1] * at[1] + at[2] * at[2]));     // :878
        const float ax[3] = {at[0] / tan_theta, at[1] / tan_theta, at[2] / tan_theta};      // :879
        const float sin_theta = (float)((double)tan_theta / sqrt(1.0 + (double)(tan_theta * tan_theta)));   // :884
        const float cos_theta = sin_theta / tan_theta;                                      // :885
        const float t[3] = {tt[0] * cos_theta, tt[1] * cos_theta, tt[2] * cos_theta};
        const float K[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0};
        float Ks[9], KK[9], Rod[9];
        const float omc = 1 - cos_theta;
        for (int i = 0; i < 9; i++) Ks[i] = omc * K[i];
        mat3_mul_f32(Ks, K, KK);
        for (int i = 0; i < 9; i++) Rod[i] = ((i % 4 == 0) ? 1.f : 0.f) + (sin_theta * K[i] + KK[i]);   // utils.h:171-176
        const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
        const float md[3] = {ps->mean_d[0], ps->mean_d[1], ps->mean_d[2]}, nms[3] = {-ps->mean_s[0], -ps->mean_s[1], -ps->mean_s[2]};
        float Tm[16], Tt[16], Ts[16], Rm[16], t1[16], t2[16];
        set_pose_f32(Tm, I3, md); set_pose_f32(Tt, I3, t); set_pose_f32(Ts, I3, nms); set_pose_f32(Rm, Rod, zero);
        mat4_mul_f32(Tm, Rm, t1); mat4_mul_f32(t1, Tt, t2); mat4_mul_f32(t2, Rm, t1); mat4_mul_f32(t1, Ts, dT);   // :894-895
    }
    if (status == ICP_OK) {
        float np[16];
        mat4_mul_f32(dT, ps->pose, np);                                           // ICPOptimizer.h:614-620
        for (int i = 0; i < 16; i++) ps->pose[i] = np[i];
        normal_matrix_from_pose(ps->pose, ps->nmat);
    }
    if (sp.stats) {
        sp.stats->n_src = sp.n_src;
        sp.stats->n_valid = (int)n;
        for (int i = 0; i < 16; i++) sp.stats->pose[i] = ps->pose[i];
        sp.stats->rmse = -1.f;
        sp.stats->benchmark_error = -1.f;
        sp.stats->status = status;
    }
}

// Reduced sums -> pose update.  All threads of the block call it (>= 64 threads); `tot` = NSUM totals in shared memory.
template <bool WITH_SYMMETRIC = true>
__device__ __forceinline__ void solve_tail(const SolveParams& sp, const double* tot) {
    if (solve_p2plane_lanes(sp, tot)) return;             // common case, spread over the lanes of this block
    if (threadIdx.x == 0) solve_generic<WITH_SYMMETRIC>(sp, tot);
}

// Grid of NSUM_USED blocks: block a folds the partials of sum a in a fixed order (lanes stride the producer blocks, shuffle tree,
// then the waves in order) -- identical on every run and independent of block scheduling.  The block that finishes last (ticket
// counter, release/acquire fences at agent scope) gathers the NSUM totals and runs the small fp64 solve + pose composition.
constexpr int SOLVE_THREADS = 256;
constexpr unsigned long long TOTAL_SENTINEL = 0xFFF8D1CEC0DE5EEDull;      // a NaN payload no arithmetic produces: "total not written yet"
constexpr int SPIN_LIMIT = 1 << 22;                        // polls of ~1 us: seconds, not a hang
constexpr int SOLVE_INFLIGHT = 12;                        // loads in flight per thread: 12 x 256 = 3072 partials in ONE memory round trip
__global__ __launch_bounds__(SOLVE_THREADS) void k_reduce_solve(const SolveParams sp) {
    __shared__ double tot[NSUM];
    __shared__ double wsum[SOLVE_THREADS / WAVE];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, a = blockIdx.x;
    {
        const double* __restrict__ row = sp.partials + (size_t)a * sp.nblocks;
        // The partials come from the producer kernel's write-back: every load is a trip to memory, and what this kernel costs is the
        // number of DEPENDENT trips.  All of a thread's loads are issued before the first add (fixed assignment b = t + 256 j, added in
        // the order of j: the same sum on every run); 2 895 partials are one round, not three.
        double x = 0.0;
        for (int b0 = 0; b0 < sp.nblocks; b0 += SOLVE_INFLIGHT * SOLVE_THREADS) {
            double v[SOLVE_INFLIGHT];
#pragma unroll
            for (int j = 0; j < SOLVE_INFLIGHT; j++) { const int b = b0 + j * SOLVE_THREADS + (int)threadIdx.x; v[j] = b < sp.nblocks ? row[b] : 0.0; }
#pragma unroll
            for (int j = 0; j < SOLVE_INFLIGHT; j++) { const int b = b0 + j * SOLVE_THREADS + (int)threadIdx.x; if (b < sp.nblocks) x += v[j]; }
        }
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
        if (lane == 0) wsum[w] = x;
    }
    __syncthreads();
    if (sp.spin) {
        // Hand-over without a ticket: every total is ONE naturally aligned 8-byte write-through store and validates itself (anything
        // but the sentinel the slots hold between launches), so nothing has to be ordered against anything: block 0 -- always
        // resident, like the other 33 -- polls the 34 slots with sc1 loads, one lane per slot, takes the values, puts the sentinels
        // back and solves.  Against store -> drain -> ticket -> re-load that is two dependent trips to memory less per launch.  The
        // wait is bounded: after SPIN_LIMIT polls (seconds) the launch gives up, raises PoseState::fault and the run reports
        // ICP_ERR_HIP instead of hanging or solving with a slot that was never written.
        if (threadIdx.x == 0) {
            double x = wsum[0];
            for (int k = 1; k < SOLVE_THREADS / WAVE; k++) x += wsum[k];
            unsigned long long bits = (unsigned long long)__double_as_longlong(x);
            if (bits == TOTAL_SENTINEL) bits ^= 1ull;         // (still a NaN: the solve's result is the same)
            __hip_atomic_store((unsigned long long*)sp.totals + a, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a != 0) return;
        if (threadIdx.x < NSUM) {
            double v = 0.0;
            if (threadIdx.x < NSUM_USED) {
                unsigned long long* slot = (unsigned long long*)sp.totals + threadIdx.x;
                unsigned long long bits = TOTAL_SENTINEL;
                for (int spin = 0; spin < SPIN_LIMIT; spin++) {
                    bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (bits != TOTAL_SENTINEL) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (bits == TOTAL_SENTINEL) sp.ps->fault = 1;         // never written within the bound
                v = __longlong_as_double((long long)bits);
                __hip_atomic_store(slot, TOTAL_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            }
            tot[threadIdx.x] = v;
        }
        __syncthreads();
        solve_tail(sp, tot);
        return;
    }
    if (threadIdx.x == 0) {
        double x = wsum[0];
        for (int k = 1; k < SOLVE_THREADS / WAVE; k++) x += wsum[k];
        // hand-over without fences (MI355X_MICROARCH.md, valid forms): write-through store of the total, drained, then the ticket; the
        // last arriver reads the totals with sc1 loads issued after its add has returned.  A release / acquire fence pair here is an L2
        // write-back plus an L1 invalidate per block, ~3 us of this kernel's ~9.
        __hip_atomic_store(sp.totals + a, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(sp.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == (unsigned)(NSUM_USED - 1));
    }
    __syncthreads();
    if (!is_last) return;
    if (threadIdx.x < NSUM) tot[threadIdx.x] = threadIdx.x < NSUM_USED ? __hip_atomic_load(sp.totals + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;   // rows NSUM_USED.. are padding
    __syncthreads();
    if (threadIdx.x == 0) *sp.ticket = 0u;                // ready for the next launch on this stream
    solve_tail(sp, tot);
}

// Measured and NOT adopted (round 2): the same reduction + solve as ONE block of 1024 threads for producers that leave few partials
// (k_post's 512; the fused matcher with 512-thread blocks, 724 partials) -- no hand-over between blocks, but 8.1-9.0 us against
// 7.4 us for the 34-block form (and the matcher itself is slower with 256- / 512-thread blocks: 22.1 k / 20.3 k vs 22.6 k it/s).

// ---- the ring form: reduce + solve of iteration i - 1 riding IN FRONT of the matcher launch of iteration i ---------------------
// k_reduce_solve is a launch of its own: 6.7 us per iteration, 2.4 of them the bare cost of a dependent launch, behind a matcher whose
// own launch costs the same -- a fifth of a converged run.  In the merged loop (point-to-plane through the fused BVH matcher, see
// run_loop) blocks 0..NSUM_USED-1 of the matcher grid of iteration i ARE the reducer of iteration i - 1: they are dispatched first and
// wait for nothing that comes after them, so the matcher blocks behind them may wait for their result without a deadlock.  The matcher
// blocks issue every load that does not need the pose (source point, normal, search state, previous neighbour's record), and only then
// wait for the pose: one launch per iteration, with reduce + solve hidden behind the matcher's own load burst.
//   Hand-overs, all of the "self-validating 8-byte granule" kind (MI355X_MICROARCH.md, hand-off price list: handoff-1to1 -- one
//   naturally aligned 8-byte sc1 store per granule, sc1-load polls, no fence, no flag, no ordering between granules):
//   * totals: reducer block a -> block 0, slot a of THIS iteration's row of the totals ring;
//   * pose:   block 0 -> every matcher block, the 16 granules (128 bytes) of THIS iteration's slot of the pose ring.
//   Both rings are filled with all-ones (hipMemsetAsync 0xFF) at the start of every run, and a slot is written exactly once per run:
//   "not written yet" is the one bit pattern no writer stores (a writer flips the lowest bit of a value that happens to equal it), and
//   nothing has to be re-armed -- a run that was cut short leaves nothing behind that the next run could mistake for a result.
//   Every wait is bounded (SPIN_LIMIT polls, seconds): a waiter that gives up raises *run_fault and the host repeats the run with the
//   separate k_reduce_solve launches.  The same happens (PoseState::fault = 2 in the published slot, the rest of the chain passes it on
//   without touching anything) when a pivot of the 6 x 6 system fails the rank test, i.e. when the solve needs the eigen fallback that
//   only k_reduce_solve carries (131 VGPRs: it must not ride in the matcher).
// Thousands of waves polling ONE 128-byte line is a hot spot, not a broadcast: sc1 loads are served behind the L2, one memory channel
// hands out a line every ~2 ns, so a round of 5 790 polls takes 11 us (measured: 24-28 us per converged iteration with a single slot,
// against 14 + 7 with one launch per iteration).  Every slot therefore exists POSE_REPLICAS times on lines far enough apart to land on
// different channels whatever the interleave; block 0 publishes all of them, a wave polls the one its index selects.
#ifndef ICP_POSE_REPLICAS
#define ICP_POSE_REPLICAS 16
#endif
constexpr int POSE_REPLICAS = ICP_POSE_REPLICAS;
constexpr size_t POSE_REPLICA_STRIDE = 4096 + 128;
__device__ __host__ __forceinline__ PoseState* loop_slot(PoseState* base, int g, int replica) {
    return (PoseState*)((char*)base + ((size_t)g * POSE_REPLICAS + (size_t)replica) * POSE_REPLICA_STRIDE);
}
#if ICP_DEBUG_TIMES
#define LOOP_STAMP(slot, j) do { if (L.dbg && g == L.dbg_iter && (threadIdx.x & 63) == 0) L.dbg[8 * (slot) + (j)] = (int)(unsigned int)wall_clock64(); } while (0)
#else
#define LOOP_STAMP(slot, j)
#endif
// One launch in front of a run of the merged / one-launch loop: slot 0 = the incoming pose in every replica, every other slot's granules
// and the totals rows "empty", the fault word and the clocks zero.  (Only the 128 bytes of a replica that are ever read are touched --
// the replicas sit 4 KB apart.)
__global__ void k_run_init(const PoseState* __restrict__ src, PoseState* slots, int n_slots, unsigned long long* totals, int n_totals, int* zero_words, int n_zero) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_slot_granules = n_slots * POSE_REPLICAS * 16;
    if (t < n_slot_granules) {
        const int g = t / (POSE_REPLICAS * 16), r = (t / 16) % POSE_REPLICAS, q = t & 15;
        unsigned long long* dst = (unsigned long long*)loop_slot(slots, g, r) + q;
        *dst = g == 0 ? ((const unsigned long long*)src)[q] : ~0ull;
    } else if (t < n_slot_granules + n_totals) totals[t - n_slot_granules] = ~0ull;
    else if (t < n_slot_granules + n_totals + n_zero) zero_words[t - n_slot_granules - n_totals] = 0;
}
constexpr unsigned long long GRANULE_EMPTY = ~0ull;
#ifndef ICP_RING_SLEEP
#define ICP_RING_SLEEP 4           // s_sleep argument (x 64 clocks) between two polls of the pose slot by a matcher wave
#endif
struct RingParams {
    const double* red_partials; int red_nblocks;   // partials of the iteration being reduced: [NSUM][red_nblocks]
    unsigned long long* totals_row;                // [NSUM] that iteration's row of the totals ring
    const PoseState* ps_in;                        // the pose that iteration searched at (replica 0 of its slot): complete since the previous launch
    PoseState* ps_out;                             // slot to publish, POSE_REPLICAS copies of it (loop_slot): the pose the matcher blocks of THIS launch wait for
    icp_iter_stats* stats; int n_src;              // record of the reduced iteration
    PoseState* final_out;                          // the closing launch of a run: the final pose state also goes here (beside the records: ONE copy back); else nullptr
    int* run_fault;                                // raised by any waiter that ran out of polls
    int n_red;                                     // reducer blocks in front of this grid: 0 (first launch of a run) or NSUM_USED
};
__device__ __forceinline__ unsigned long long granule_of(unsigned int lo, unsigned int hi) {
    const unsigned long long g = ((unsigned long long)hi << 32) | lo;
    return g == GRANULE_EMPTY ? g ^ 1ull : g;             // (all-ones is a NaN pair either way)
}
// All threads of reducer block a = blockIdx.x < NSUM_USED; blockDim.x == RING_THREADS (two waves).  The fold reproduces k_reduce_solve's
// summation order exactly -- thread t stands for the threads t and t + 128 of its 256-thread block: same strided assignment, same
// shuffle trees, the four wave sums added in the same order -- so the merged loop and the separate launches give bit-identical poses
// (tests/test_gpu_merged.py compares them).
constexpr int RING_THREADS = 128;
#ifndef ICP_DEBUG_TIMES
#define ICP_DEBUG_TIMES 0
#endif
#if ICP_DEBUG_TIMES
// development builds: thread 0 of reducer block a stamps the 100 MHz clock at [8 a + j]: j = 0 block start, 1 folded, 2 total published;
// block 0 also at [8 NSUM_USED + j]: 0 totals received, 1 solved, 2 pose published  (icp_debug_ring_times; the stamps of the last matcher launch of a run stay: the closing launch does not stamp)
__device__ int g_ring_dbg[(NSUM_USED + 1) * 8];
#define RING_STAMP(a, j) do { if (threadIdx.x == 0 && !rp.final_out) g_ring_dbg[8 * (a) + (j)] = (int)(unsigned int)wall_clock64(); } while (0)
#else
#define RING_STAMP(a, j)
#endif
__device__ __forceinline__ void ring_reduce_solve(const RingParams& rp) {
    __shared__ double tot[NSUM];
    __shared__ double wsum[4];
    __shared__ unsigned int slot_words[32];
    __shared__ int give_up;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, a = blockIdx.x;
    RING_STAMP(a, 0);
    if (a == 0 && rp.ps_in->fault) {                      // the chain was cut further up: pass it on, touch nothing else
        { const unsigned int* src = (const unsigned int*)rp.ps_in; for (int q = tid; q < 16 * POSE_REPLICAS; q += RING_THREADS) __hip_atomic_store((unsigned long long*)loop_slot(rp.ps_out, 0, q >> 4) + (q & 15), granule_of(src[2 * (q & 15)], src[2 * (q & 15) + 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (rp.final_out && tid < 32) ((unsigned int*)rp.final_out)[tid] = src[tid]; }
        return;
    }
    {
        const double* __restrict__ row = rp.red_partials + (size_t)a * rp.red_nblocks;
        double x0 = 0.0, x1 = 0.0;                          // virtual threads tid and tid + 128 of the 256-thread fold
        for (int b0 = 0; b0 < rp.red_nblocks; b0 += SOLVE_INFLIGHT * SOLVE_THREADS) {
            double v0[SOLVE_INFLIGHT], v1[SOLVE_INFLIGHT];
#pragma unroll
            for (int j = 0; j < SOLVE_INFLIGHT; j++) {
                const int b = b0 + j * SOLVE_THREADS + tid;
                v0[j] = b < rp.red_nblocks ? row[b] : 0.0;
                v1[j] = b + RING_THREADS < rp.red_nblocks ? row[b + RING_THREADS] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < SOLVE_INFLIGHT; j++) {
                const int b = b0 + j * SOLVE_THREADS + tid;
                if (b < rp.red_nblocks) x0 += v0[j];
                if (b + RING_THREADS < rp.red_nblocks) x1 += v1[j];
            }
        }
        for (int off = 32; off > 0; off >>= 1) { x0 += __shfl_down(x0, off, WAVE); x1 += __shfl_down(x1, off, WAVE); }
        if (lane == 0) { wsum[w] = x0; wsum[2 + w] = x1; }
    }
    RING_STAMP(a, 1);
    __syncthreads();
    if (tid == 0) {
        double x = wsum[0];
        for (int k = 1; k < 4; k++) x += wsum[k];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
        __hip_atomic_store(rp.totals_row + a, granule_of((unsigned int)bits, (unsigned int)(bits >> 32)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RING_STAMP(a, 2);
    if (a != 0) return;
    if (tid == 0) give_up = 0;
    __syncthreads();
    if (tid < NSUM) {
        double v = 0.0;
        if (tid < NSUM_USED) {
            unsigned long long bits = GRANULE_EMPTY;
            for (int spin = 0; spin < SPIN_LIMIT; spin++) {
                bits = __hip_atomic_load(rp.totals_row + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (bits != GRANULE_EMPTY) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (bits == GRANULE_EMPTY) give_up = 1;         // never written within the bound
            v = __longlong_as_double((long long)bits);
        }
        tot[tid] = v;
    }
    __syncthreads();
    RING_STAMP(NSUM_USED, 0);
    const PoseState* pin = rp.ps_in;
    const double n = tot[SUM_N];
    int fault = 0, status = ICP_OK;
    const float* npose = nullptr;
    if (give_up) { fault = 1; if (tid == 0) atomicOr(rp.run_fault, 1); }
    else if (n > 0) { npose = p2plane_lanes_core(tot + SUM_M, pin->pose); if (!npose) fault = 2; }      // (uniform: every thread of the block takes the same branch)
    else status = ICP_ERR_NO_CORRESPONDENCES;              // the pose stays (ICPOptimizer.h:668,680: the reference would hang in ASSERT)
    RING_STAMP(NSUM_USED, 1);
    // the slot: pose, normal matrix, means, fault -- assembled in LDS, published as 16 granules
    if (tid < 16) slot_words[tid] = __float_as_uint(npose ? npose[tid] : pin->pose[tid]);
    if (tid >= 32 && tid < 41) slot_words[16 + tid - 32] = __float_as_uint(npose ? normal_matrix_entry(npose, tid - 32) : pin->nmat[tid - 32]);      // nine lanes, one quotient each
    if (tid >= 64 && tid < 70) {                           // the means of the valid pairs (symmetric ICP reads them; kept defined here): six lanes, one quotient each
        const int k = tid - 64;
        slot_words[25 + k] = __float_as_uint(n > 0 ? (float)(tot[(k < 3 ? SUM_S : SUM_D - 3) + k] / n) : 0.f);
    }
    if (tid == 70) slot_words[31] = (unsigned int)fault;
    __syncthreads();
    for (int q = tid; q < 16 * POSE_REPLICAS; q += RING_THREADS)      // every replica of the slot, 16 granules each
        __hip_atomic_store((unsigned long long*)loop_slot(rp.ps_out, 0, q >> 4) + (q & 15), granule_of(slot_words[2 * (q & 15)], slot_words[2 * (q & 15) + 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    RING_STAMP(NSUM_USED, 2);
    if (rp.final_out && tid < 32) ((unsigned int*)rp.final_out)[tid] = slot_words[tid];
    if (rp.stats && !fault) {                             // the record of the reduced iteration (read by the host after the run)
        if (tid < 16) rp.stats->pose[tid] = __uint_as_float(slot_words[tid]);
        if (tid == 16) { rp.stats->n_src = rp.n_src; rp.stats->n_valid = (int)n; rp.stats->rmse = -1.f; rp.stats->benchmark_error = -1.f; rp.stats->status = status; }
    }
}
// The reducer alone: closes a merged run (the last iteration has no matcher launch behind it to ride in).
__global__ __launch_bounds__(RING_THREADS) void k_ring_reduce_solve(const RingParams rp) { ring_reduce_solve(rp); }

// The matcher side of the pose hand-over: lanes 0..15 of the wave poll the 16 granules of the slot until none is empty, then the
// pose and the normal matrix move to scalar registers.  Returns false when the wait ran out (*run_fault raised) or the slot carries
// a fault: the caller's block leaves without writing anything.
__device__ __forceinline__ bool ring_wait_pose(const PoseState* slot, int lane, int* run_fault, float (&Pm)[16], float (&Nm)[9]) {
    const unsigned long long* g = (const unsigned long long*)slot;
    unsigned long long v = 0ull;
    bool ok = false;
    for (int spin = 0; spin < SPIN_LIMIT; spin++) {
        if (lane < 16) v = __hip_atomic_load(g + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!__any(lane < 16 && v == GRANULE_EMPTY)) { ok = true; break; }
        __builtin_amdgcn_s_sleep(ICP_RING_SLEEP);
    }
    if (!ok) { if (lane == 0) atomicOr(run_fault, 1); return false; }
    const int lo = (int)(unsigned int)v, hi = (int)(unsigned int)(v >> 32);
#pragma unroll
    for (int q = 0; q < 16; q++) Pm[q] = __int_as_float(__builtin_amdgcn_readlane((q & 1) ? hi : lo, q >> 1));
#pragma unroll
    for (int q = 0; q < 9; q++) Nm[q] = __int_as_float(__builtin_amdgcn_readlane(((16 + q) & 1) ? hi : lo, (16 + q) >> 1));
    return __builtin_amdgcn_readlane(hi, 15) == 0;
}

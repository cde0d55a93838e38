// dev_post.hpp -- deterministic reductions, system rows, k_post, the fused matcher k_knn_bvh_post, k_sym_accumulate.
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.
// ------------------------------------------------------------------------------------------------
// Deterministic block reduction of NV doubles per thread: wave shuffle tree, then the 4 wave
// results are added in wave order by wave 0.  Result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double* lds /* [4][NV] */) {
#pragma unroll
    for (int a = 0; a < NV; a++) {
        double x = v[a];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
        v[a] = x;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < NV; a++) lds[w * NV + a] = v[a];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < NV; a++) v[a] = ((lds[a] + lds[NV + a]) + lds[2 * NV + a]) + lds[3 * NV + a];
    }
}

// Same contract for MANY accumulators (the 34 sums of k_post): a full shuffle tree would be 6 x 2 x NV LDS-crossbar permutes
// per wave.  Here two shuffle steps fold 64 lanes to 16, those 16 partials go through LDS transposed ([wave][value][16+1]),
// and thread a < NV adds the NW x 16 partials of value a in a fixed order.  lds: NW * NV * 17 doubles.  Result: thread a holds
// the block total of accumulator a (a < NV); returned through `out`.
template <int NV, int NW>
__device__ __forceinline__ double block_reduce_wide(double (&v)[NV], double* lds) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < NV; a++) {
        double x = v[a];
        x += __shfl_down(x, 32, WAVE);
        x += __shfl_down(x, 16, WAVE);
        if (lane < 16) lds[(w * NV + a) * 17 + lane] = x;
    }
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < NV) {
#pragma unroll
        for (int ww = 0; ww < NW; ww++) {
            const double* row = lds + (ww * NV + threadIdx.x) * 17;
            double part = 0.0;
#pragma unroll
            for (int l = 0; l < 16; l++) part += row[l];
            tot += part;
        }
    }
    return tot;
}

// Rows of the reference's 4n x 6 system in fp32 (kind 0: point-to-plane, ICPOptimizer.h:698-750; kind 1: symmetric,
// ICPOptimizer.h:806-852, s/d already centred, n = n_t + n_s).  Row 0 is dense and scaled by LAMBDA_PLANE/SYMMETRIC = 1 times
// the weight; rows 1-3 are the point rows [0, s2, -s1, 1,0,0 | d0-s0], [-s2, 0, s0, 0,1,0 | d1-s1], [s1, -s0, 0, 0,0,1 | d2-s2]
// scaled by LAMBDA_POINT = 0.1 times the weight (:737-750 / :839-852) -- kept as their non-zero entries only.
struct RowTerms {
    float r0[7];                 // row 0: 6 coefficients + right-hand side
    float p1, p2, rr1;           // row 1: columns 1, 2 (+ g at column 3), rhs
    float q0, q2, rr2;           // row 2: columns 0, 2 (+ g at column 4), rhs
    float t0, t1, rr3;           // row 3: columns 0, 1 (+ g at column 5), rhs
    float g;                     // 1 * f1
};
__device__ __forceinline__ void build_rows(int kind, float s0, float s1, float s2, float d0, float d1, float d2, float n0, float n1, float n2, float w, RowTerms& R) {
    float A0[6], b0;
    if (kind == 0) {
        A0[0] = n2 * s1 - n1 * s2; A0[1] = n0 * s2 - n2 * s0; A0[2] = n1 * s0 - n0 * s1;
        b0 = ((n0 * d0 + n1 * d1) + n2 * d2) - ((n0 * s0 + n1 * s1) + n2 * s2);
    } else {
        const float e0 = s0 + d0, e1 = s1 + d1, e2 = s2 + d2;
        const float g0 = d0 - s0, g1 = d1 - s1, g2 = d2 - s2;
        A0[0] = e1 * n2 - e2 * n1; A0[1] = e2 * n0 - e0 * n2; A0[2] = e0 * n1 - e1 * n0;
        b0 = g0 * n0 + (g1 * n1 + g2 * n2);
    }
    A0[3] = n0; A0[4] = n1; A0[5] = n2;
    const float f0 = 1.0f * w, f1 = 0.1f * w;
#pragma unroll
    for (int c = 0; c < 6; c++) R.r0[c] = A0[c] * f0;
    R.r0[6] = b0 * f0;
    R.g = 1.f * f1;
    R.p1 = s2 * f1; R.p2 = (-s1) * f1; R.rr1 = (d0 - s0) * f1;
    R.q0 = (-s2) * f1; R.q2 = s0 * f1; R.rr2 = (d1 - s1) * f1;
    R.t0 = s1 * f1; R.t1 = (-s0) * f1; R.rr3 = (d2 - s2) * f1;
}

// Contribution of one point's rows to slot A of [J^T J upper triangle (21) | J^T r (6)], in fp64, rows added in order 0..3.
// For finite weights the zero entries of rows 1-3 contribute +0.0 in the dense form, so these sums are exactly those of the
// dense 4n x 6 system.  Upper-triangle slot of (a, c), a <= c: a * 6 - a (a - 1) / 2 + (c - a).
template <int A>
__device__ __forceinline__ double row_slot(const RowTerms& R) {
    constexpr int ta = A < 6 ? 0 : A < 11 ? 1 : A < 15 ? 2 : A < 18 ? 3 : A < 20 ? 4 : A < 21 ? 5 : A - 21;      // row index a (or a of J^T r)
    constexpr int tc = A < 21 ? ta + (A - (ta * 6 - ta * (ta - 1) / 2)) : 6;                                      // column c (6 = rhs)
    double v = (double)R.r0[ta] * (double)R.r0[tc];
    // row 1: entries at columns 1 (p1), 2 (p2), 3 (g), rhs rr1
    {
        constexpr bool ha = ta == 1 || ta == 2 || ta == 3, hc = tc == 1 || tc == 2 || tc == 3 || tc == 6;
        if (ha && hc) v += (double)(ta == 1 ? R.p1 : ta == 2 ? R.p2 : R.g) * (double)(tc == 1 ? R.p1 : tc == 2 ? R.p2 : tc == 3 ? R.g : R.rr1);
    }
    // row 2: columns 0 (q0), 2 (q2), 4 (g), rhs rr2
    {
        constexpr bool ha = ta == 0 || ta == 2 || ta == 4, hc = tc == 0 || tc == 2 || tc == 4 || tc == 6;
        if (ha && hc) v += (double)(ta == 0 ? R.q0 : ta == 2 ? R.q2 : R.g) * (double)(tc == 0 ? R.q0 : tc == 2 ? R.q2 : tc == 4 ? R.g : R.rr2);
    }
    // row 3: columns 0 (t0), 1 (t1), 5 (g), rhs rr3
    {
        constexpr bool ha = ta == 0 || ta == 1 || ta == 5, hc = tc == 0 || tc == 1 || tc == 5 || tc == 6;
        if (ha && hc) v += (double)(ta == 0 ? R.t0 : ta == 1 ? R.t1 : R.g) * (double)(tc == 0 ? R.t0 : tc == 1 ? R.t1 : tc == 5 ? R.g : R.rr3);
    }
    return v;
}
template <int A>
__device__ __forceinline__ void add_row_slots(const RowTerms& R, double* acc) {
    if constexpr (A < 27) {
        // same sequence of additions per slot as the row-by-row accumulation: acc += row0 term, += row1 term, ...
        constexpr int ta = A < 6 ? 0 : A < 11 ? 1 : A < 15 ? 2 : A < 18 ? 3 : A < 20 ? 4 : A < 21 ? 5 : A - 21;
        constexpr int tc = A < 21 ? ta + (A - (ta * 6 - ta * (ta - 1) / 2)) : 6;
        acc[A] += (double)R.r0[ta] * (double)R.r0[tc];
        { constexpr bool ha = ta == 1 || ta == 2 || ta == 3, hc = tc == 1 || tc == 2 || tc == 3 || tc == 6;
          if (ha && hc) acc[A] += (double)(ta == 1 ? R.p1 : ta == 2 ? R.p2 : R.g) * (double)(tc == 1 ? R.p1 : tc == 2 ? R.p2 : tc == 3 ? R.g : R.rr1); }
        { constexpr bool ha = ta == 0 || ta == 2 || ta == 4, hc = tc == 0 || tc == 2 || tc == 4 || tc == 6;
          if (ha && hc) acc[A] += (double)(ta == 0 ? R.q0 : ta == 2 ? R.q2 : R.g) * (double)(tc == 0 ? R.q0 : tc == 2 ? R.q2 : tc == 4 ? R.g : R.rr2); }
        { constexpr bool ha = ta == 0 || ta == 1 || ta == 5, hc = tc == 0 || tc == 1 || tc == 5 || tc == 6;
          if (ha && hc) acc[A] += (double)(ta == 0 ? R.t0 : ta == 1 ? R.t1 : R.g) * (double)(tc == 0 ? R.t0 : tc == 1 ? R.t1 : tc == 5 ? R.g : R.rr3); }
        add_row_slots<A + 1>(R, acc);
    }
}
__device__ __forceinline__ void accumulate_rows(int kind, float s0, float s1, float s2, float d0, float d1, float d2,
                                                float n0, float n1, float n2, float w, double* acc /* 27 */) {
    RowTerms R;
    build_rows(kind, s0, s1, s2, d0, d1, d2, n0, n1, n2, w, R);
    add_row_slots<0>(R, acc);
}

struct PostParams {
    const float* sx; const float* sy; const float* sz;
    const float* snx; const float* sny; const float* snz;
    const uint32_t* srgba; const int* sel; int n;
    const float* tx; const float* ty; const float* tz;
    const float* tnx; const float* tny; const float* tnz; const uint32_t* trgba;
    const PoseState* ps;
    icp_match_t* matches;          // in: after matching; out: after weighting + pruning
    int metric, weighting, rejection;
    float max_dist, cos_reject;  // cos_reject: largest float c with acosf(c) > 60 deg on this host's libm
    double* partials;            // [NSUM][gridDim.x]: sum a of block b at a * gridDim.x + b (the reducer reads rows contiguously)
};

// Weight, reject and filter ONE correspondence (source position k, match m after matching, matched target point d / normal nt /
// colour tcol): the body of applyWeights / pruneCorrespondences / the validity filter.  Writes the final Match back; returns
// whether the pair enters the system, with the transformed source point and the weight.  post_core adds the system build.
template <bool HAVE_S = false>
__device__ __forceinline__ bool post_eval(const PostParams& pp, int k, icp_match_t m, float d0, float d1, float d2, float nt0, float nt1, float nt2, uint32_t tcol,
                                          float& s0, float& s1, float& s2, float& w, float rn0 = 0.f, float rn1 = 0.f, float rn2 = 0.f) {
    const float* __restrict__ P = pp.ps->pose;
    const float* __restrict__ N = pp.ps->nmat;
    const int i = pp.sel ? pp.sel[k] : k;
    float ns0, ns1, ns2;
    if (!HAVE_S) {                                         // HAVE_S: the matcher passes the transformed point and the raw source normal
        xform_point(P, pp.sx[i], pp.sy[i], pp.sz[i], s0, s1, s2);
        rn0 = pp.snx[i]; rn1 = pp.sny[i]; rn2 = pp.snz[i];
    }
    xform_normal(N, rn0, rn1, rn2, ns0, ns1, ns2);
    const bool fin_sd = finite3(s0, s1, s2) && finite3(d0, d1, d2);
    // ---- applyWeights, weighting.h:44-90 ----
    if (pp.weighting != ICP_WEIGHT_CONSTANT) {
        float wnew = 0.0f;
        if (pp.weighting == ICP_WEIGHT_DISTANCES || pp.weighting == ICP_WEIGHT_COLORS) {
            if (fin_sd) {
                const float e0 = s0 - d0, e1 = s1 - d1, e2 = s2 - d2;
                const float q = ((e0 * e0 + e1 * e1) + e2 * e2) / pp.max_dist;
                wnew += (float)(1.0 - (double)q);          // weighting.h:19
            }
        }
        if (pp.weighting == ICP_WEIGHT_NORMALS) {
            if (finite3(ns0, ns1, ns2) && finite3(nt0, nt1, nt2))
                wnew += ns0 * nt0 + (ns1 * nt1 + ns2 * nt2);   // weighting.h:24 (Eigen dot tree)
        }
        if (pp.weighting == ICP_WEIGHT_COLORS) {
            const uint32_t a = pp.srgba[i], b = tcol;
            const int e0 = (int)(uint8_t)((a & 0xFF) - (b & 0xFF));           // weighting.h:28 uint8 wrap-around
            const int e1 = (int)(uint8_t)(((a >> 8) & 0xFF) - ((b >> 8) & 0xFF));
            const int e2 = (int)(uint8_t)(((a >> 16) & 0xFF) - ((b >> 16) & 0xFF));
            const float cq = (float)(e0 * e0 + e1 * e1 + e2 * e2) / (float)195075;
            wnew *= (float)(1.0 - (double)cq);             // weighting.h:29,86
        }
        m.weight = wnew;
    }
    // ---- pruneCorrespondences, ICPOptimizer.h:157-174 ----
    if (pp.rejection == 1) {
        const float dt = ns0 * nt0 + (ns1 * nt1 + ns2 * nt2);
        const float na = sqrtf(ns0 * ns0 + (ns1 * ns1 + ns2 * ns2));
        const float nb = sqrtf(nt0 * nt0 + (nt1 * nt1 + nt2 * nt2));
        const float c = dt / (na * nb);
        // acos(c) > 60deg  <=>  -1 <= c <= cos_reject ; NaN / |c| > 1 => acos is NaN => kept
        if (c >= -1.0f && c <= pp.cos_reject) m.idx = -1;
    }
    pp.matches[k] = m;
    w = m.weight;
    return m.idx >= 0 && fin_sd;                           // ICPOptimizer.h:596-598
}
__device__ __forceinline__ void post_core(const PostParams& pp, int k, icp_match_t m, float d0, float d1, float d2, float nt0, float nt1, float nt2, uint32_t tcol,
                                          double* acc /* 34 */) {
    float s0, s1, s2, w;
    if (!post_eval(pp, k, m, d0, d1, d2, nt0, nt1, nt2, tcol, s0, s1, s2, w)) return;
    acc[SUM_N] += 1.0;
    acc[SUM_S] += (double)s0; acc[SUM_S + 1] += (double)s1; acc[SUM_S + 2] += (double)s2;
    acc[SUM_D] += (double)d0; acc[SUM_D + 1] += (double)d1; acc[SUM_D + 2] += (double)d2;
    if (pp.metric == ICP_METRIC_POINT_TO_PLANE) {
        accumulate_rows(0, s0, s1, s2, d0, d1, d2, nt0, nt1, nt2, w, acc + SUM_M);
    } else if (pp.metric == ICP_METRIC_POINT_TO_POINT) {
        const double wd = (double)w;
        acc[SUM_M] += wd;
        const double ws0 = wd * s0, ws1 = wd * s1, ws2 = wd * s2;
        acc[SUM_M + 1] += ws0; acc[SUM_M + 2] += ws1; acc[SUM_M + 3] += ws2;
        acc[SUM_M + 4] += wd * d0; acc[SUM_M + 5] += wd * d1; acc[SUM_M + 6] += wd * d2;
        acc[SUM_M + 7] += (double)d0 * ws0;  acc[SUM_M + 8] += (double)d0 * ws1;  acc[SUM_M + 9] += (double)d0 * ws2;
        acc[SUM_M + 10] += (double)d1 * ws0; acc[SUM_M + 11] += (double)d1 * ws1; acc[SUM_M + 12] += (double)d1 * ws2;
        acc[SUM_M + 13] += (double)d2 * ws0; acc[SUM_M + 14] += (double)d2 * ws1; acc[SUM_M + 15] += (double)d2 * ws2;
    }
}

// The same, with the matched target gathered from the target planes by original index (scan / projective matchers).
__device__ __forceinline__ void post_point(const PostParams& pp, int k, icp_match_t m, double* acc /* 34 */) {
    if (m.idx < 0) return;
    const int j = m.idx;
    post_core(pp, k, m, pp.tx[j], pp.ty[j], pp.tz[j], pp.tnx[j], pp.tny[j], pp.tnz[j], pp.weighting == ICP_WEIGHT_COLORS ? pp.trgba[j] : 0u, acc);
}

// One fused pass over the correspondences (weight, reject, filter, accumulate) -- used after the scan / projective matchers.
__global__ __launch_bounds__(POST_THREADS) void k_post(const PostParams pp) {
    __shared__ double lds[4 * 34 * 17];
    double acc[34];
#pragma unroll
    for (int a = 0; a < 34; a++) acc[a] = 0.0;
    for (int k = blockIdx.x * POST_THREADS + threadIdx.x; k < pp.n; k += gridDim.x * POST_THREADS) post_point(pp, k, pp.matches[k], acc);
    const double tot = block_reduce_wide<34, 4>(acc, lds);
    if (threadIdx.x < 34) pp.partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = tot;
}

// BVH k-NN with the post stage as its epilogue: the lane that found the neighbour of query k immediately weighs / rejects /
// accumulates it, so matches never make a round trip through memory.  One kernel instead of two per iteration.  Each lane
// has exactly one pair, so the 34 sums are not accumulated in registers first: every value is produced, folded 64 -> 16 lanes
// with two shuffles and parked in LDS right away (groups separated by scheduling barriers), which keeps the kernel at the
// register budget of the walk.  Block partials keep the fixed-order reduction contract.
// N values at a time, stage by stage (all shuffles of a stage in flight together: the cross-lane permutes have ~100 cycles of
// latency each, a value-by-value chain would expose 2 x 34 of them per wave).
template <int N>
__device__ __forceinline__ void fold_store_n(double (&x)[N], double* lds, int a0, int lane, int w) {
    double y[N];
#pragma unroll
    for (int i = 0; i < N; i++) y[i] = __shfl_down(x[i], 32, WAVE);
#pragma unroll
    for (int i = 0; i < N; i++) x[i] += y[i];
#pragma unroll
    for (int i = 0; i < N; i++) y[i] = __shfl_down(x[i], 16, WAVE);
#pragma unroll
    for (int i = 0; i < N; i++) x[i] += y[i];
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < N; i++) lds[(w * 34 + a0 + i) * 17 + lane] = x[i];
    }
}
template <int A, int N, int I = 0>
__device__ __forceinline__ void row_slots_n(const RowTerms& R, bool valid, double (&x)[N]) {
    if constexpr (I < N) { x[I] = valid ? row_slot<A + I>(R) : 0.0; row_slots_n<A, N, I + 1>(R, valid, x); }
}
template <int A, int N>
__device__ __forceinline__ void fold_row_slots(const RowTerms& R, bool valid, double* lds, int lane, int w) {
    double x[N];
    row_slots_n<A, N>(R, valid, x);
    fold_store_n<N>(x, lds, SUM_M + A, lane, w);
}
template <int DIM>
__global__ __launch_bounds__(BVH_THREADS) void k_knn_bvh_post(const KnnParams kp, const BvhViewT<DIM> bv, const int* __restrict__ qorder, const PostParams pp) {
    extern __shared__ uint2 bvh_lbq[];                    // [Lq][BVH_THREADS] pending-sibling bounds; reused by the reduction
    constexpr int NW = BVH_THREADS / WAVE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t = xcd_contiguous_block(blockIdx.x, gridDim.x) * BVH_THREADS + tid;
    const int k = (t < kp.n) ? (qorder ? qorder[t] : t) : -1;
    bool valid = false;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, wt = 0.f;
    float p[DIM];
#pragma unroll
    for (int q = 0; q < DIM; q++) p[q] = 0.f;
    float rn0 = 0.f, rn1 = 0.f, rn2 = 0.f;
    float best = FLT_MAX, lb_others = 0.f; int bi = -1, bpos = -1, q0 = -1;
    float4 ra, rb; ra.x = 0.f; ra.y = 0.f; ra.z = 0.f; ra.w = 0.f; rb = ra;
    bool need_walk = false;
    if (k >= 0) {
        // ---- front end.  Everything that depends only on the query index is requested in ONE batch (point, normal, previous
        // neighbour, search state), then the neighbour's record: two memory round trips before the verify test instead of
        // one per array -- in the late iterations, where almost no query walks, those round trips ARE the kernel.
        const int i = kp.sel ? kp.sel[k] : k;
        const float r0 = kp.sx[i], r1 = kp.sy[i], r2 = kp.sz[i];
        if (DIM == 6) { p[3 % DIM] = kp.scr[i]; p[4 % DIM] = kp.scg[i]; p[5 % DIM] = kp.scb[i]; }
        rn0 = pp.snx[i]; rn1 = pp.sny[i]; rn2 = pp.snz[i];
        const bool seeded = kp.use_prev != 0, inc = kp.incremental && seeded;
        q0 = seeded ? kp.nn_raw[k] : -1;
        float4 st; st.x = 0.f; st.y = 0.f; st.z = 0.f; st.w = 0.f;
        if (inc) st = kp.qstate[k];
        xform_point(kp.ps->pose, r0, r1, r2, p[0], p[1], p[2]);
        if (finite3(p[0], p[1], p[2]) && bv.n_valid > 0) {
            need_walk = true;
            if (q0 >= 0) {                                 // seed_from_previous + knn_try_verify, on the batched loads
                float tq[DIM]; int j0;
                if (DIM == 3) { ra = *(const float4*)(bv.recs + q0); rb = *((const float4*)(bv.recs + q0) + 1); tq[0] = ra.x; tq[1] = ra.y; tq[2] = ra.z; j0 = __float_as_int(ra.w); }
                else {
                    const BvhLeafT<DIM>* lf = bv.leaves + (q0 >> 3);
#pragma unroll
                    for (int q = 0; q < DIM; q++) tq[q] = lf->c[q][q0 & 7];
                    j0 = lf->idx[q0 & 7];
                }
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < DIM; q++) { const float e = p[q] - tq[q]; d = (q == 0) ? e * e : d + e * e; }
                if (d < best) { best = d; bi = j0; bpos = q0; }
                if (inc && bi >= 0) {
                    const float ex = p[0] - st.x, ey = p[1] - st.y, ez = p[2] - st.z;
                    const float delta = sqrtf((ex * ex + ey * ey) + ez * ez) * 1.000001f + 1e-30f;
                    const float lbn = (st.w - delta) * 0.999999f;
                    if (sqrtf(best) * 1.000001f < lbn) { lb_others = lbn; need_walk = false; }
                }
            }
        }
    }
    // ---- the walks.  A wave left with only a few seeded queries to search does them cooperatively, one after the other
    // (coop_search); otherwise every lane walks on its own.
    {
        unsigned long long wm = __ballot(need_walk);
        const unsigned long long cm = __ballot(need_walk && bpos >= 0);
        if (ICP_COOP_MAX > 0 && wm != 0ull && wm == cm && __popcll(wm) <= ICP_COOP_MAX && bv.Lq > 0) {
            const int lane = tid & 63;
            while (wm) {
                const int src = __ffsll((long long)wm) - 1; wm &= wm - 1ull;
                float q[DIM];
#pragma unroll
                for (int a = 0; a < DIM; a++) q[a] = __shfl(p[a], src, WAVE);
                float b = __shfl(best, src, WAVE), lbo = 0.f; int ci = __shfl(bi, src, WAVE), cps = __shfl(bpos, src, WAVE);
                const bool done = coop_search<DIM, BVH_THREADS>(bv, q, b, ci, cps, lbo, bvh_lbq, tid);      // wave-uniform
                if (done && lane == src) { best = b; bi = ci; bpos = cps; lb_others = lbo; need_walk = false; }
            }
        }
    }
    if (need_walk) lb_others = knn_walk<DIM, BVH_THREADS>(bv, p, best, bi, bpos, bvh_lbq, tid);
    if (k >= 0) {
        knn_store_state<DIM>(kp, k, p, best, bpos, lb_others);
        icp_match_t m;
        if (best <= kp.max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }
        if (m.idx < 0) pp.matches[k] = m;
        else {
            if (DIM != 3 || bpos != q0) { ra = *(const float4*)(bv.recs + bpos); rb = *((const float4*)(bv.recs + bpos) + 1); }      // one 32-byte record
            d0 = ra.x; d1 = ra.y; d2 = ra.z; n0 = rb.x; n1 = rb.y; n2 = rb.z;
            s0 = p[0]; s1 = p[1]; s2 = p[2];
            valid = post_eval<true>(pp, k, m, d0, d1, d2, n0, n1, n2, __float_as_uint(rb.w), s0, s1, s2, wt, rn0, rn1, rn2);
        }
    }
    __syncthreads();                                      // the traversal stacks are dead: reuse LDS for the reduction
    double* lds = (double*)bvh_lbq;
    {   // the count is an integer: one ballot per wave, stored as lane 0's "partial" (the other 15 are zero)
        const unsigned long long vm = __ballot(valid);
        if (lane < 16) lds[(w * 34 + SUM_N) * 17 + lane] = (lane == 0) ? (double)__popcll(vm) : 0.0;
    }
    if (pp.metric == ICP_METRIC_POINT_TO_PLANE) {
        // the sums of s and d feed only the means (point-to-point / symmetric): not needed here, their slots stay zero
        if (lane < 16) {
#pragma unroll
            for (int q = 0; q < 6; q++) lds[(w * 34 + SUM_S + q) * 17 + lane] = 0.0;
        }
    } else {
        double x[6] = {valid ? (double)s0 : 0.0, valid ? (double)s1 : 0.0, valid ? (double)s2 : 0.0,
                       valid ? (double)d0 : 0.0, valid ? (double)d1 : 0.0, valid ? (double)d2 : 0.0};
        fold_store_n<6>(x, lds, SUM_S, lane, w);          // SUM_S.., SUM_D.. are slots 1..6
    }
    __builtin_amdgcn_sched_barrier(0);
    if (pp.metric == ICP_METRIC_POINT_TO_PLANE) {
        RowTerms R;
        build_rows(0, s0, s1, s2, d0, d1, d2, n0, n1, n2, wt, R);
        fold_row_slots<0, 7>(R, valid, lds, lane, w);   __builtin_amdgcn_sched_barrier(0);
        fold_row_slots<7, 7>(R, valid, lds, lane, w);   __builtin_amdgcn_sched_barrier(0);
        fold_row_slots<14, 7>(R, valid, lds, lane, w);  __builtin_amdgcn_sched_barrier(0);
        fold_row_slots<21, 6>(R, valid, lds, lane, w);
    } else {                                              // point-to-point moments (see post_core)
        const double wd = (double)wt;
        const double ws[3] = {wd * s0, wd * s1, wd * s2};
        const float dd[3] = {d0, d1, d2};
        {
            double x[7] = {valid ? wd : 0.0, valid ? ws[0] : 0.0, valid ? ws[1] : 0.0, valid ? ws[2] : 0.0,
                           valid ? wd * dd[0] : 0.0, valid ? wd * dd[1] : 0.0, valid ? wd * dd[2] : 0.0};
            fold_store_n<7>(x, lds, SUM_M, lane, w);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            double x[9];
#pragma unroll
            for (int j = 0; j < 3; j++) {
#pragma unroll
                for (int q = 0; q < 3; q++) x[j * 3 + q] = valid ? (double)dd[j] * ws[q] : 0.0;
            }
            fold_store_n<9>(x, lds, SUM_M + 7, lane, w);
        }
        if (lane < 16) {
#pragma unroll
            for (int q = 16; q < 27; q++) lds[(w * 34 + SUM_M + q) * 17 + lane] = 0.0;
        }
    }
    __syncthreads();
    if (tid < 34) {
        double tot = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ww++) {
            const double* row = lds + (ww * 34 + tid) * 17;
            double part = 0.0;
#pragma unroll
            for (int l = 0; l < 16; l++) part += row[l];
            tot += part;
        }
        const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);             // partial slot = logical block -> fixed summation order
        pp.partials[(size_t)tid * gridDim.x + lb] = tot;
    }
}

// Second pass of the symmetric objective: rows need the means of the valid pairs first
// (ICPOptimizer.h:797-809).  Reads the final matches written by k_post.
__global__ __launch_bounds__(POST_THREADS) void k_sym_accumulate(const PostParams pp) {
    __shared__ double lds[4 * 27 * 17];
    double acc[27];
#pragma unroll
    for (int a = 0; a < 27; a++) acc[a] = 0.0;
    const float* __restrict__ P = pp.ps->pose;
    const float* __restrict__ N = pp.ps->nmat;
    const float ms0 = pp.ps->mean_s[0], ms1 = pp.ps->mean_s[1], ms2 = pp.ps->mean_s[2];
    const float md0 = pp.ps->mean_d[0], md1 = pp.ps->mean_d[1], md2 = pp.ps->mean_d[2];
    for (int k = blockIdx.x * POST_THREADS + threadIdx.x; k < pp.n; k += gridDim.x * POST_THREADS) {
        const icp_match_t m = pp.matches[k];
        if (m.idx < 0) continue;
        const int i = pp.sel ? pp.sel[k] : k;
        float s0, s1, s2, ns0, ns1, ns2;
        xform_point(P, pp.sx[i], pp.sy[i], pp.sz[i], s0, s1, s2);
        const int j = m.idx;
        const float d0 = pp.tx[j], d1 = pp.ty[j], d2 = pp.tz[j];
        if (!(finite3(s0, s1, s2) && finite3(d0, d1, d2))) continue;
        xform_normal(N, pp.snx[i], pp.sny[i], pp.snz[i], ns0, ns1, ns2);
        const float n0 = pp.tnx[j] + ns0, n1 = pp.tny[j] + ns1, n2 = pp.tnz[j] + ns2;    // :809
        accumulate_rows(1, s0 - ms0, s1 - ms1, s2 - ms2, d0 - md0, d1 - md1, d2 - md2, n0, n1, n2, m.weight, acc);
    }
    const double tot = block_reduce_wide<27, 4>(acc, lds);
    if (threadIdx.x < 27) pp.partials[(size_t)(SUM_M + threadIdx.x) * gridDim.x + blockIdx.x] = tot;
}

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


// ---- wave-level TRANSPOSING reduction of N doubles per lane, in registers --------------------------------------------------
// A shuffle tree folds every value over all 64 lanes: 6 steps x N values, each step a cross-lane permute.  Here every step
// halves the NUMBER of values a lane holds instead: the two halves of the lane set exchange the values the other half keeps
// (lanes with the step's bit clear keep value 2j and receive the partner's copy of it, lanes with the bit set keep value 2j+1),
// so N values cost N/2 + N/4 + ... exchanges, and the steps over lane bits 5 and 4 are v_permlane32_swap / v_permlane16_swap
// (gfx950, plain VALU) instead of trips through the LDS crossbar; bits 3..0 are DPP row operations.  At the end value v sits,
// summed over the whole wave, in the lanes whose bits (5,4,3,2,1,0) spell (v0,v1,v2,v3,v4,v5) -- see wave_value_of_lane.
// The pairing is fixed, so the sums are bitwise reproducible.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int dbl_lo(double v) { return (unsigned int)(unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned int dbl_hi(double v) { return (unsigned int)((unsigned long long)__double_as_longlong(v) >> 32); }
__device__ __forceinline__ double dbl_from(unsigned int lo, unsigned int hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }

// lanes 0-31 / even rows end up with a (own + partner's), lanes 32-63 / odd rows with b
template <int SPAN>
__device__ __forceinline__ double swap_add(double a, double b) {
    u32x2 r0, r1;
    if (SPAN == 32) { r0 = __builtin_amdgcn_permlane32_swap(dbl_lo(a), dbl_lo(b), false, false); r1 = __builtin_amdgcn_permlane32_swap(dbl_hi(a), dbl_hi(b), false, false); }
    else            { r0 = __builtin_amdgcn_permlane16_swap(dbl_lo(a), dbl_lo(b), false, false); r1 = __builtin_amdgcn_permlane16_swap(dbl_hi(a), dbl_hi(b), false, false); }
    return dbl_from(r0[0], r1[0]) + dbl_from(r0[1], r1[1]);
}
// DPP controls that pair lanes across bit 3 (row_mirror), 2 (row_half_mirror), 1 and 0 (quad permutes)
template <int BIT> struct DppCtrl { static constexpr int v = BIT == 3 ? 0x140 : BIT == 2 ? 0x141 : BIT == 1 ? 0x4E : 0xB1; };
template <int BIT>
__device__ __forceinline__ double dpp_add(double a, double b, int lane) {
    const bool up = (lane >> BIT) & 1;
    const double send = up ? a : b, keep = up ? b : a;
    const double recv = dbl_from((unsigned int)__builtin_amdgcn_update_dpp(0, (int)dbl_lo(send), DppCtrl<BIT>::v, 0xF, 0xF, false),
                                 (unsigned int)__builtin_amdgcn_update_dpp(0, (int)dbl_hi(send), DppCtrl<BIT>::v, 0xF, 0xF, false));
    return keep + recv;
}
template <int STEP, int N>                        // STEP 0..5 <-> lane bit 5..0
__device__ __forceinline__ double wave_transpose_reduce_from(double (&x)[N], int lane) {
    if constexpr (STEP == 6) { static_assert(N == 1, "at most 64 values"); return x[0]; }
    else {
        constexpr int H = (N + 1) / 2;
        double y[H];
#pragma unroll
        for (int j = 0; j < H; j++) {
            const double a = x[2 * j], b = (2 * j + 1 < N) ? x[2 * j + 1] : 0.0;
            if constexpr (STEP == 0) y[j] = swap_add<32>(a, b);
            else if constexpr (STEP == 1) y[j] = swap_add<16>(a, b);
            else y[j] = dpp_add<5 - STEP>(a, b, lane);
        }
        return wave_transpose_reduce_from<STEP + 1, H>(y, lane);
    }
}
// The same reduction evaluated depth-first over a value GENERATOR (f.get<A>() produces value A): group (K, BASE) = the 2^K values
// from BASE on, folded through steps 0..K-1.  Only one partial per level is alive at any time -- O(log N) registers instead of
// N / 2 -- which is what lets the fused matcher keep the register budget of the tree walk.  Same pairing, same result.
template <int K, int BASE, class F>
__device__ __forceinline__ double wave_transpose_reduce_gen(const F& f, int lane) {
    if constexpr (K == 0) return f.template get<BASE>();
    else {
        constexpr int HALF = 1 << (K - 1);
        const double a = wave_transpose_reduce_gen<K - 1, BASE>(f, lane);
        double b = 0.0;                                   // a group past the last value is an exact zero, the exchange still runs
        if constexpr (BASE + HALF < F::N) b = wave_transpose_reduce_gen<K - 1, BASE + HALF>(f, lane);
        if constexpr (K == 3) __builtin_amdgcn_sched_barrier(0);       // keeps the generator's products from being hoisted across groups
        if constexpr (K == 1) return swap_add<32>(a, b);
        else if constexpr (K == 2) return swap_add<16>(a, b);
        else return dpp_add<6 - K>(a, b, lane);
    }
}
// index of the value whose wave total this lane holds after the reduction (values >= N: zero)
__device__ __forceinline__ int wave_value_of_lane(int lane) {
    return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4) | ((lane & 1) << 5);
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
    // (the product of two fp32 values is exact in fp64 -- 48 significant bits -- so fma(a, b, v) rounds the same sum as a * b followed by + v:
    //  one instruction instead of two, bit-identical)
    double v = (double)R.r0[ta] * (double)R.r0[tc];
    // row 1: entries at columns 1 (p1), 2 (p2), 3 (g), rhs rr1
    {
        constexpr bool ha = ta == 1 || ta == 2 || ta == 3, hc = tc == 1 || tc == 2 || tc == 3 || tc == 6;
        if (ha && hc) v = fma((double)(ta == 1 ? R.p1 : ta == 2 ? R.p2 : R.g), (double)(tc == 1 ? R.p1 : tc == 2 ? R.p2 : tc == 3 ? R.g : R.rr1), v);
    }
    // row 2: columns 0 (q0), 2 (q2), 4 (g), rhs rr2
    {
        constexpr bool ha = ta == 0 || ta == 2 || ta == 4, hc = tc == 0 || tc == 2 || tc == 4 || tc == 6;
        if (ha && hc) v = fma((double)(ta == 0 ? R.q0 : ta == 2 ? R.q2 : R.g), (double)(tc == 0 ? R.q0 : tc == 2 ? R.q2 : tc == 4 ? R.g : R.rr2), v);
    }
    // row 3: columns 0 (t0), 1 (t1), 5 (g), rhs rr3
    {
        constexpr bool ha = ta == 0 || ta == 1 || ta == 5, hc = tc == 0 || tc == 1 || tc == 5 || tc == 6;
        if (ha && hc) v = fma((double)(ta == 0 ? R.t0 : ta == 1 ? R.t1 : R.g), (double)(tc == 0 ? R.t0 : tc == 1 ? R.t1 : tc == 5 ? R.g : R.rr3), v);
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
    icp_match_t* matches;          // in: after matching; out: after weighting + pruning.  The fused matcher may get nullptr (records not kept)
    int metric, weighting, rejection;
    float max_dist, cos_reject;  // cos_reject: largest float c with acosf(c) > 60 deg on this host's libm
    double* partials;            // [NSUM][gridDim.x]: sum a of block b at a * gridDim.x + b (the reducer reads rows contiguously)
};

// Weight, reject and filter ONE correspondence (source position k, match m after matching, matched target point d / normal nt /
// colour tcol): the body of applyWeights / pruneCorrespondences / the validity filter.  Writes the final Match back; returns
// whether the pair enters the system, with the transformed source point and the weight.  post_core adds the system build.
template <bool HAVE_S = false>
__device__ __forceinline__ bool post_eval(const PostParams& pp, int k, icp_match_t m, float d0, float d1, float d2, float nt0, float nt1, float nt2, uint32_t tcol,
                                          float& s0, float& s1, float& s2, float& w, float rn0 = 0.f, float rn1 = 0.f, float rn2 = 0.f, const float* nmat = nullptr) {
    const float* __restrict__ P = pp.ps->pose;
    const float* __restrict__ N = nmat ? nmat : pp.ps->nmat;      // (the fused matcher hands over the copy it fetched with scalar loads)
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
    if (pp.matches) pp.matches[k] = m;                     // nullptr: nobody reads the records (fused point-to-point / point-to-plane loop)
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

// Hardware self test of wave_transpose_reduce_gen (icp_selftest_wave_reduce, tests/test_gpu_selftest.py).
struct ArrayGen27 { const double* x; static constexpr int N = 27; template <int A> __device__ __forceinline__ double get() const { return x[A]; } };
__global__ void k_selftest_wave_reduce(const double* __restrict__ in /* [64][27] */, double* __restrict__ out /* [27] */, int* __restrict__ lane_of /* [27] */) {
    const int lane = threadIdx.x & 63;
    double x[27];
#pragma unroll
    for (int a = 0; a < 27; a++) x[a] = in[lane * 27 + a];
    const ArrayGen27 g{x};
    const double tot = wave_transpose_reduce_gen<6, 0>(g, lane);
    const int v = wave_value_of_lane(lane);
    if (v < 27) { out[v] = tot; lane_of[v] = lane; }
}

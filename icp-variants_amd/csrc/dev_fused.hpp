// dev_fused.hpp -- the fused matcher k_knn_bvh_post: exact BVH 1-NN + weight / reject / accumulate in ONE launch per ICP iteration
// (the block partials go to k_reduce_solve).
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.
// ------------------------------------------------------------------------------------------------
// BVH k-NN with the post stage as its epilogue: the lane that found the neighbour of query k immediately weighs / rejects /
// accumulates it, so matches never make a round trip through memory.  One kernel instead of two per iteration.  Each lane
// has at most one pair: its contributions are produced two at a time and fed straight into the first step of the transposing
// wave reduction, which keeps the kernel at the register budget of the walk.  Block partials keep the fixed-order contract.
struct RowSlotGen {                                       // value A = slot A of one pair's point-to-plane rows (a lane without a pair: all-zero rows)
    const RowTerms& R;
    static constexpr int N = 27;
    template <int A> __device__ __forceinline__ double get() const { return row_slot<A>(R); }
};
struct P2pGen {                                           // values 0..21 = sum s (3), sum d (3), then SUM_M + 0..15 of post_core's point-to-point block
    float s[3], d[3]; double wd;
    static constexpr int N = 22;
    template <int A> __device__ __forceinline__ double get() const {
        double v;
        if constexpr (A < 3) v = (double)s[A];
        else if constexpr (A < 6) v = (double)d[A - 3];
        else if constexpr (A == 6) v = wd;
        else if constexpr (A < 10) v = wd * s[A - 7];
        else if constexpr (A < 13) v = wd * d[A - 10];
        else v = (double)d[(A - 13) / 3] * (wd * s[(A - 13) % 3]);
        return v;
    }
};
// Measured and NOT adopted (round 2): folding the block partials and solving INSIDE this launch (two levels of "last arriver
// folds", write-through hand-over without fences).  Correct (the whole GPU suite passed with it) but slower: every level is three
// dependent trips to memory (publish, ticket, fold) at ~1.5 us each, the tail of the launch grew by ~20 us against ~12 us for the
// separate k_reduce_solve launch (17.3 k vs 20.0 k iterations/s).  Also measured and dropped: handing the queries of a block to its lanes
// in the order of their last walk's length (waves of like lengths, walker-free waves once many verify; one byte of assignment per
// query, a stable 8-class rank in the epilogue): iterations 1-9 0.084 vs 0.079 ms -- the prediction is not worth the extra dependent
// load in front of everything and the gathers.  What did pay was making k_reduce_solve itself cheaper (one load
// round, fence-free hand-over): see dev_solve.hpp.
#ifndef ICP_FUSED_WAVES
#define ICP_FUSED_WAVES 6        // waves per SIMD the 3-D fused matchers are compiled for (80 registers)
#endif
#ifndef ICP_WAVE_STRIDE
#define ICP_WAVE_STRIDE ((ICP_XW && ICP_BVH_THREADS > 64) ? 128 : 0)      // 1: the waves of a block come from BVH_THREADS / 64 places of the query order (hard and easy regions meet in one block: knn_walk_shared, XW)
#endif
#ifndef ICP_DEBUG_WALK_ENDS
#define ICP_DEBUG_WALK_ENDS 0
#endif
#ifndef ICP_DEBUG_TIMES
#define ICP_DEBUG_TIMES 0        // (default set in dev_solve.hpp) 1 (with ICP_DEBUG_STEPS=1 for the buffer): lane 0 of every wave leaves 100 MHz timestamps of its phases in dbg_steps[8 * wave ..]
#endif
#if ICP_DEBUG_TIMES
#define ICP_STAMP(j) do { if (kp.dbg_steps && lane == 0) kp.dbg_steps[8 * (wave_slot) + (j)] = (int)(unsigned int)wall_clock64(); } while (0)
#else
#define ICP_STAMP(j)
#endif

// What a query brings along that does not depend on the pose: requested in ONE batch (point, normal, previous neighbour, search
// state), then the neighbour's record -- two memory round trips before the verify test instead of one per array; in the late
// iterations, where almost no query walks, those round trips ARE the per-launch kernels.  The persistent loop (dev_persist.hpp) keeps
// all of it in registers from one iteration to the next for as long as the wave's queries verify.
template <int DIM> struct QueryIn {
    float r0, r1, r2;            // source point (untransformed)
    float c3, c4, c5;            // colour features (DIM == 6)
    float rn0, rn1, rn2;         // source normal
    int q0;                      // position (8 * leaf + slot) of the previous neighbour; -1: none
    float4 st;                   // anchor of the last search + lower bound on the distance from it to every other target
    float4 ra, rb;               // the previous neighbour's 32-byte record (DIM == 3: its coordinates too)
    float tq[DIM]; int j0;       // the previous neighbour's coordinates and original index
};
template <int DIM>
__device__ __forceinline__ void fused_front_clear(QueryIn<DIM>& in) {
    in.r0 = 0.f; in.r1 = 0.f; in.r2 = 0.f; in.c3 = 0.f; in.c4 = 0.f; in.c5 = 0.f; in.rn0 = 0.f; in.rn1 = 0.f; in.rn2 = 0.f; in.q0 = -1; in.j0 = -1;
    in.st.x = 0.f; in.st.y = 0.f; in.st.z = 0.f; in.st.w = 0.f; in.ra = in.st; in.rb = in.st;
#pragma unroll
    for (int q = 0; q < DIM; q++) in.tq[q] = 0.f;
}
template <int DIM>
__device__ __forceinline__ void fused_front_loads(const KnnParams& kp, const PostParams& pp, const BvhViewT<DIM>& bv, int k, bool seeded, bool inc, QueryIn<DIM>& in) {
    fused_front_clear<DIM>(in);
    if (k >= 0) {
        const int i = kp.sel ? kp.sel[k] : k;
        in.r0 = kp.sx[i]; in.r1 = kp.sy[i]; in.r2 = kp.sz[i];
        if (DIM == 6) { in.c3 = kp.scr[i]; in.c4 = kp.scg[i]; in.c5 = kp.scb[i]; }
        in.rn0 = pp.snx[i]; in.rn1 = pp.sny[i]; in.rn2 = pp.snz[i];
        in.q0 = seeded ? kp.nn_raw[k] : -1;
        if (inc) in.st = kp.qstate[k];                     // (the second tier's 8 bytes are fetched only by the queries the first tier does not verify)
        if (in.q0 >= 0) {
            if (DIM == 3) { in.ra = *(const float4*)(bv.recs + in.q0); in.rb = *((const float4*)(bv.recs + in.q0) + 1); in.tq[0] = in.ra.x; in.tq[1] = in.ra.y; in.tq[2] = in.ra.z; in.j0 = __float_as_int(in.ra.w); }
            else {
                const BvhLeafT<DIM>* lf = bv.leaves + (in.q0 >> 3);
#pragma unroll
                for (int q = 0; q < DIM; q++) in.tq[q] = lf->c[q][in.q0 & 7];
                in.j0 = lf->idx[in.q0 & 7];
            }
        }
    }
}

// Which 64 queries of the (Morton-sorted) order does wave w of logical block lb take?  ICP_WAVE_STRIDE 0: the block's waves are
// neighbours; 1: they come from NW places of the order, a grid's worth of waves apart (hard and easy regions meet in one block);
// S > 1: from NW places S waves apart (groups of S blocks share a stretch of NW * S waves; a last, partial group keeps neighbours).
__host__ __device__ __forceinline__ int fused_wave_slot(int lb, int w, int mgrid) {
    constexpr int NW = BVH_THREADS / WAVE, S = ICP_WAVE_STRIDE;
    if (S == 0 || NW == 1) return lb * NW + w;
    if (S == 1) return w * mgrid + lb;
    const int g = lb / S, j = lb - g * S;
    return (g + 1) * S <= mgrid ? (g * NW + w) * S + j : lb * NW + w;
}

// One correspondence, as the epilogue consumes it.
struct PairOut { bool valid; float s0, s1, s2, d0, d1, d2, n0, n1, n2, wt; };

// Transform, verify (both tiers), search what does not verify (walks shared over the wave), weigh / reject: everything of one
// iteration between "the pose is known" and "the lane holds its pair".  `searched` = this lane's query walked or took the two-leaf tier
// (its search state was rewritten).  renewed (the persistent loop, DIM == 3): set for a lane whose query took the two-leaf tier in a wave that
// did not walk -- `in` then holds its NEW state (anchor, bound, neighbour's record), ready to be parked again; a lane whose new state
// cannot be given that way (no record fetched: the neighbour is past the distance threshold) leaves `searched` set instead.
template <int DIM, bool WIDE, bool XW = false>
__device__ __forceinline__ void fused_search_post(const KnnParams& kp, const BvhViewT<DIM>& bv, const PostParams& pp, int k, bool seeded, bool inc, const float* Pm, const float* Nm,
                                                  QueryIn<DIM>& in, uint2* __restrict__ bvh_lbq, int tid, int wave_slot, PairOut& o, bool& searched, bool* renewed = nullptr, int gx_block = 0) {
    const int lane = tid & 63; (void)lane; (void)wave_slot; (void)seeded;
    o.valid = false; o.s0 = 0.f; o.s1 = 0.f; o.s2 = 0.f; o.d0 = 0.f; o.d1 = 0.f; o.d2 = 0.f; o.n0 = 0.f; o.n1 = 0.f; o.n2 = 0.f; o.wt = 0.f;
    float p[DIM];
#pragma unroll
    for (int q = 0; q < DIM; q++) p[q] = 0.f;
    if (DIM == 6) { p[3 % DIM] = in.c3; p[4 % DIM] = in.c4; p[5 % DIM] = in.c5; }
    float rn0 = in.rn0, rn1 = in.rn1, rn2 = in.rn2;
    float best = FLT_MAX, lb_others = 0.f; int bi = -1, bpos = -1, q0 = in.q0;
    float4 ra = in.ra, rb = in.rb;
#if ICP_DEBUG_TIMES
    float dbg_lbo = 0.f, dbg_lb3 = 0.f, dbg_delta = 0.f; int dbg_l2old = -1; const int dbg_q0old = in.q0;
#endif
    bool need_walk = false, verified = false, two_leaf = false;
    float lb3 = 0.f; int l2 = -1;                         // second verification tier: bound on every target outside the neighbour's leaf and the runner-up's leaf l2
    if (k >= 0) {
        float2 st2; st2.x = 0.f; st2.y = __int_as_float(-1);
        xform_point(Pm, in.r0, in.r1, in.r2, p[0], p[1], p[2]);
        if (finite3(p[0], p[1], p[2]) && bv.n_valid > 0) {
            need_walk = true;
            if (q0 >= 0) {                                 // seed_from_previous + knn_try_verify, on the batched loads
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < DIM; q++) { const float e = p[q] - in.tq[q]; d = (q == 0) ? e * e : d + e * e; }
                if (d < best) { best = d; bi = in.j0; bpos = q0; }
                if (inc && bi >= 0) {
                    const float ex = p[0] - in.st.x, ey = p[1] - in.st.y, ez = p[2] - in.st.z;
                    const float delta = sqrt_up((ex * ex + ey * ey) + ez * ez), sbest = sqrt_up(best);
                    const float lbn = (in.st.w - delta) * 0.999999f;
#if ICP_DEBUG_TIMES
                    dbg_lbo = in.st.w; dbg_lb3 = st2.x; dbg_delta = delta;
#endif
                    if (sbest < lbn) { lb_others = lbn; need_walk = false; verified = true; }
                    else {
                        // second tier: every target outside TWO leaves -- the neighbour's and the one the runner-up of the last search lives
                        // in -- is still provably farther than the old neighbour itself -> the nearest neighbour is one of their 16 points: two
                        // leaf evaluations instead of a walk.  This is what retires the queries that sit close to the bisector of two targets,
                        // which the first tier can never verify and which would otherwise walk in every iteration.
                        if (kp.qstate2) st2 = kp.qstate2[k];
#if ICP_DEBUG_TIMES
                        dbg_l2old = __float_as_int(st2.y);
#endif
                        const float lb2 = (st2.x - delta) * 0.999999f;
                        // (a query that walks takes the runner-up's leaf along as a hint: the spread start of knn_walk_shared searches the
                        //  path to it side by side with the path to the seed's leaf)
                        l2 = __float_as_int(st2.y);
                        if (sbest < lb2) { two_leaf = true; lb3 = lb2; }
                    }
                }
            }
        }
    }
    ICP_STAMP(1);
    searched = need_walk || two_leaf;
    if (renewed) *renewed = false;
#if ICP_DEBUG_TIMES
    if (kp.dbg_steps && k >= 0 && (need_walk || two_leaf) && seeded) {      // who is it that still searches?  (slot by query index; the clock tells the launch)
        int* r = kp.dbg_steps + 8 * kp.dbg_waves + 8 * (k & 4095);
        r[0] = k; r[1] = __float_as_int(best); r[2] = __float_as_int(dbg_lbo); r[3] = __float_as_int(dbg_lb3); r[4] = __float_as_int(dbg_delta); r[5] = two_leaf ? l2 : -2; r[6] = q0; r[7] = (int)(unsigned int)wall_clock64();
    }
    if (kp.dbg_steps) { const int nw_ = __popcll(__ballot(need_walk && !two_leaf)), nl_ = __popcll(__ballot(two_leaf)); if (lane == 0) { kp.dbg_steps[8 * wave_slot + 6] = nw_ | (nl_ << 8);
            // where the wave runs: HW_ID (id 4: simd [5:4], cu [11:8], sh [12], se [15:13]) and XCC_ID (id 20) -> bits 16.. of the second word
            kp.dbg_steps[8 * wave_slot + 7] = (int)((__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu) | (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16)); } }
#endif
    const bool took_two_leaf = two_leaf;
    if (two_leaf) {
        f2 p2[DIM];
#pragma unroll
        for (int q = 0; q < DIM; q++) { p2[q].x = p[q]; p2[q].y = p[q]; }
        float b2 = FLT_MAX, b3 = FLT_MAX; int nl2 = -1;
        const int lf = q0 >> 3;
        leaf_eval<DIM>(bv.leaves + lf, lf, p2, best, bi, bpos, b2, nl2, b3);      // exact argmin over the two leaves, seeded with the old neighbour
        if (l2 >= 0 && l2 != lf) leaf_eval<DIM>(bv.leaves + l2, l2, p2, best, bi, bpos, b2, nl2, b3);
        lb_others = fminf(sqrt_dn(b2), lb3);                             // re-anchored here: the runner-up among the 16, or anything outside the two leaves
        lb3 = fminf(sqrt_dn(b3), lb3);
        l2 = nl2;
        need_walk = false;
    }
    // ---- the walks: shared over the wave (knn_walk_shared) -- the lanes whose queries verified, and those whose walks end early, take
    // parked subtrees off the lanes still searching.  (Until round 2 a wave with <= 16 walkers searched them level-synchronously in lane
    // groups instead; the shared walk does that case as well, 0.0281 vs 0.0306 ms in iterations 10-16, and the kernel without the second
    // code path needs 68 instead of 80 VGPRs.)
#if ICP_DEBUG_STEPS && !ICP_DEBUG_TIMES
    if (k >= 0 && kp.dbg_steps) kp.dbg_steps[k] = need_walk ? -1 : two_leaf ? -2 : 0;      // -1: walk (overwritten with its length); -2: second tier, two leaves
#endif
    bool walked = false;                                  // wave-uniform: this wave's lanes searched (or helped): their neighbours' records are read again
#if ICP_SHARE_WALKS
    if (__any(need_walk)) {
        walked = true;
        float rn[3] = {rn0, rn1, rn2};
        knn_walk_shared<DIM, BVH_THREADS, typename std::conditional<WIDE, unsigned long long, unsigned int>::type, XW ? 1 : 0>(bv, p, rn, need_walk, best, bi, bpos, lb_others, lb3, l2, bvh_lbq, tid, kp.fault, &kp.gx, gx_block);
        rn0 = rn[0]; rn1 = rn[1]; rn2 = rn[2];
        q0 = -2;                                          // the neighbour's record is read again below: it need not stay in registers while this lane helps
        // (the persistent loop: a wave that walked reloads its queries' data in the next iteration -- said here in a way the register
        //  allocator can see, so that none of it stays alive across the walk)
        fused_front_clear<DIM>(in);
    }
#else
    if (need_walk) lb_others = knn_walk<DIM, BVH_THREADS>(bv, p, best, bi, bpos, lb3, l2, bvh_lbq, tid, (ICP_DEBUG_STEPS && kp.dbg_steps) ? kp.dbg_steps + k : nullptr);
#endif
    ICP_STAMP(2);
#if ICP_DEBUG_TIMES && ICP_DEBUG_WALK_ENDS                  // (one atomic per walking query on ONE word: it distorts every time stamp -- a build of its own, tools/dev_walk_ends.py)
    if (k >= 0 && need_walk && seeded) {                       // where did the seeded walk end: in the old neighbour's leaf, in the old runner-up's leaf, elsewhere?
        GX_COUNT(8, 1);
        if (dbg_q0old >= 0 && (bpos >> 3) == (dbg_q0old >> 3)) GX_COUNT(9, 1);
        if (dbg_l2old >= 0 && (bpos >> 3) == dbg_l2old) GX_COUNT(10, 1);
    }
#endif
    asm volatile("" : "+v"(k));                           // (the addresses of this query's state and records are formed HERE, not held in registers across the walk)
    if (k >= 0) {
        // A verified query keeps its stored anchor (position of the last full search) and bound: the triangle test stays valid
        // against the OLD anchor -- and is tighter than re-anchoring, (L - d1) - d2 <= L - |d1 + d2| -- and its neighbour is
        // unchanged, so nothing of its search state needs rewriting.  Once ICP has converged that is > 99.9 % of the queries:
        // the 32 B per query of state stores (and the Match record, when nobody reads it) disappear from those launches.
        if (!verified) knn_store_state<DIM>(kp, k, p, best, bpos, lb_others, lb3, l2);
        else if (kp.d2_out) kp.d2_out[k] = best;
        icp_match_t m;
        if (best <= kp.max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }
        if (m.idx < 0) { if (pp.matches) pp.matches[k] = m; }
        else {
            if (walked || DIM != 3 || bpos != q0) { ra = *(const float4*)(bv.recs + bpos); rb = *((const float4*)(bv.recs + bpos) + 1); }      // one 32-byte record
            o.d0 = ra.x; o.d1 = ra.y; o.d2 = ra.z; o.n0 = rb.x; o.n1 = rb.y; o.n2 = rb.z;
            o.s0 = p[0]; o.s1 = p[1]; o.s2 = p[2];
            o.valid = post_eval<true>(pp, k, m, o.d0, o.d1, o.d2, o.n0, o.n1, o.n2, __float_as_uint(rb.w), o.s0, o.s1, o.s2, o.wt, rn0, rn1, rn2, Nm);
        }
        if (DIM == 3 && renewed && took_two_leaf && !walked && (m.idx >= 0 || bpos == q0)) {
            // the two-leaf tier re-anchored this query (knn_store_state above wrote the same to memory): hand the new state back
            in.st.x = p[0]; in.st.y = p[1]; in.st.z = p[2]; in.st.w = lb_others; in.q0 = bpos; in.ra = ra; in.rb = rb;
            in.tq[0] = ra.x; in.tq[1] = ra.y; in.tq[2] = ra.z; in.j0 = __float_as_int(ra.w);
            *renewed = true; searched = false;
        }
    }
    ICP_STAMP(3);
}

// ---- epilogue: the block's sums.  Every lane has at most one pair; its <= 27 contributions are folded over the wave with the
// transposing reduction (wave_transpose_reduce_gen: permlane swaps + DPP, no LDS traffic), the two wave totals meet in LDS, and
// threads 0..33 hand the block partial `tid` to store(tid, value) in the fixed slot order the reducer expects.  Both barriers of the
// block are in here: the first one retires the traversal's use of the LDS rows.
template <class StoreFn>
__device__ __forceinline__ void fused_block_epilogue(const KnnParams& kp, const PostParams& pp, PairOut& o, uint2* __restrict__ bvh_lbq, int tid, int wave_slot, const StoreFn& store) {
    constexpr int NW = BVH_THREADS / WAVE;
    const int lane = tid & 63, w = tid >> 6; (void)wave_slot; (void)kp;
    __syncthreads();                                      // the traversal stacks are dead: reuse LDS for the reduction
    double* lds = (double*)bvh_lbq;                       // [NW][32] wave totals, then [NW] counts
    double tot;
    // a lane without a valid pair contributes zeros: its INPUTS are zeroed (10 selects) rather than each of its 27 contributions (54)
    if (!o.valid) { o.s0 = 0.f; o.s1 = 0.f; o.s2 = 0.f; o.d0 = 0.f; o.d1 = 0.f; o.d2 = 0.f; o.n0 = 0.f; o.n1 = 0.f; o.n2 = 0.f; o.wt = 0.f; }
    if (pp.metric == ICP_METRIC_POINT_TO_PLANE) {
        RowTerms R;
        build_rows(0, o.s0, o.s1, o.s2, o.d0, o.d1, o.d2, o.n0, o.n1, o.n2, o.wt, R);
        const RowSlotGen g{R};                            // 27 slots of [J^T J | J^T r]
        tot = wave_transpose_reduce_gen<6, 0>(g, lane);
    } else {                                              // point-to-point moments (see post_core): sum s, sum d, then the 16 weighted ones
        const P2pGen g{{o.s0, o.s1, o.s2}, {o.d0, o.d1, o.d2}, (double)o.wt};
        tot = wave_transpose_reduce_gen<6, 0>(g, lane);
    }
    ICP_STAMP(4);
    {
        const unsigned long long vm = __ballot(o.valid);  // the count is an integer: one ballot per wave
        const int v = wave_value_of_lane(lane);
        if (v < 32) lds[w * 32 + v] = tot;                // values past the last one are exact zeros
        if (lane == 0) lds[NW * 32 + w] = (double)__popcll(vm);
    }
    __syncthreads();
    if (tid < NSUM_USED) {                                // threads 0..33: this block's sum `tid`
        double out = 0.0;
        if (tid == SUM_N) { for (int ww = 0; ww < NW; ww++) out += lds[NW * 32 + ww]; }
        else {
            // point-to-plane: slots SUM_M.. <- values 0..26 (the sums of s and d feed only the means: not needed, zero);
            // point-to-point: slots 1..22 <- values 0..21
            const int v = pp.metric == ICP_METRIC_POINT_TO_PLANE ? tid - SUM_M : tid - 1;
            const int nv = pp.metric == ICP_METRIC_POINT_TO_PLANE ? 27 : 22;
            if (v >= 0 && v < nv) { for (int ww = 0; ww < NW; ww++) out += lds[ww * 32 + v]; }
        }
        store(tid, out);
    }
    ICP_STAMP(5);
}

// One launch per iteration.  MERGED: the launch of iteration i carries the reducer of iteration i - 1 in its first rp.n_red blocks and
// every matcher block waits for the pose they publish (dev_solve.hpp, "the ring form") -- AFTER it has issued the loads that do not
// need the pose.  (Measured and dropped: holding the matcher's burst back by 3 / 6 / 12 us so that the reducer's loads go first -- 33.5 k /
// 31.6 k / 26.0 k iterations/s against 33.5 k without -- and loading only after the pose has arrived: 32.8 k.)
template <int DIM, bool WIDE, bool MERGED>      // WIDE: trees deeper than 8 levels of 4-wide nodes (> 524 288 targets) keep 64 pending bits per lane
__device__ __forceinline__ void fused_matcher_body(const KnnParams& kp, const BvhViewT<DIM>& bv, const int* __restrict__ qorder, const PostParams& pp, const RingParams& rp) {
    extern __shared__ uint2 bvh_lbq[];                    // [ICP_SHARE_ROWS][BVH_THREADS]: the shared walk's records; reused by the reduction
    constexpr int NW = BVH_THREADS / WAVE;
    if constexpr (MERGED) { if ((int)blockIdx.x < rp.n_red) { if (BVH_THREADS == RING_THREADS || threadIdx.x < RING_THREADS) ring_reduce_solve(rp); return; } }
    const int n_red = MERGED ? rp.n_red : 0;
    const int mblock = (int)blockIdx.x - n_red, mgrid = (int)gridDim.x - n_red;      // this block / the grid among the matcher blocks
    // Pose and normal matrix through the constant address space: wave-uniform and unchanged for the length of the launch (k_reduce_solve
    // wrote them before it), so they can be SCALAR loads, issued here with the arguments, instead of per-lane vector loads that the
    // compiler places behind the wait for the query's own loads (one more dependent trip in front of every launch) and keeps in 21 VGPRs.
    // (MERGED: they arrive through ring_wait_pose below, into scalar registers as well.)
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    float Pm[16], Nm[9];
    if constexpr (!MERGED) {
        const cfloat_p pc = (cfloat_p)(const void*)kp.ps->pose; const cfloat_p nc = (cfloat_p)(const void*)kp.ps->nmat;
#pragma unroll
        for (int q = 0; q < 16; q++) Pm[q] = pc[q];
#pragma unroll
        for (int q = 0; q < 9; q++) Nm[q] = nc[q];
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lb = xcd_contiguous_block(mblock, mgrid);                              // partial slot = logical block -> fixed summation order
    const int wave_slot = fused_wave_slot(lb, w, mgrid);
    const int t = wave_slot * BVH_QPW + lane;
    ICP_STAMP(0);
    const int k = ((BVH_QPW == WAVE || lane < BVH_QPW) && t < kp.n) ? (qorder ? qorder[t] : t) : -1;
    const bool seeded = kp.use_prev != 0, inc = kp.incremental && seeded;
    QueryIn<DIM> in;
    fused_front_loads<DIM>(kp, pp, bv, k, seeded, inc, in);
    constexpr bool XW = xw_enabled<DIM, BVH_THREADS>();
    if constexpr (XW) {                                   // the board of the cross-wave hand-over starts empty (LDS only: no wait for the loads above)
        xw_init<BVH_THREADS>(bvh_lbq, tid);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if constexpr (MERGED) { if (!ring_wait_pose(loop_slot((PoseState*)kp.ps, 0, (int)((blockIdx.x * 2u + (unsigned int)w) % (unsigned int)POSE_REPLICAS)), lane, rp.run_fault, Pm, Nm)) return; }
    PairOut o; bool searched;
    fused_search_post<DIM, WIDE, XW>(kp, bv, pp, k, seeded, inc, Pm, Nm, in, bvh_lbq, tid, wave_slot, o, searched, nullptr, lb);
    if constexpr (XW) {
        // this wave holds its pairs; before the block's sums, it helps the waves of the block that still search (the pair waits in the
        // wave's own LDS rows, which nobody reads once its queries are complete)
        if (xw_block_is_searching<BVH_THREADS>(bvh_lbq)) {
            constexpr int NT = BVH_THREADS;
            bvh_lbq[3 * NT + tid] = make_uint2(__float_as_uint(o.s0), __float_as_uint(o.s1)); bvh_lbq[4 * NT + tid] = make_uint2(__float_as_uint(o.s2), __float_as_uint(o.d0));
            bvh_lbq[5 * NT + tid] = make_uint2(__float_as_uint(o.d1), __float_as_uint(o.d2)); bvh_lbq[7 * NT + tid] = make_uint2(__float_as_uint(o.n0), __float_as_uint(o.n1));
            bvh_lbq[8 * NT + tid] = make_uint2(__float_as_uint(o.n2), __float_as_uint(o.wt)); bvh_lbq[9 * NT + tid] = make_uint2(o.valid ? 1u : 0u, 0u);
            xw_help<DIM, NT, typename std::conditional<WIDE, unsigned long long, unsigned int>::type>(bv, bvh_lbq, tid, kp.fault, &kp.gx, lb);
            const uint2 a = bvh_lbq[3 * NT + tid], b = bvh_lbq[4 * NT + tid], c = bvh_lbq[5 * NT + tid], d = bvh_lbq[7 * NT + tid], e = bvh_lbq[8 * NT + tid], f = bvh_lbq[9 * NT + tid];
            o.s0 = __uint_as_float(a.x); o.s1 = __uint_as_float(a.y); o.s2 = __uint_as_float(b.x); o.d0 = __uint_as_float(b.y); o.d1 = __uint_as_float(c.x); o.d2 = __uint_as_float(c.y);
            o.n0 = __uint_as_float(d.x); o.n1 = __uint_as_float(d.y); o.n2 = __uint_as_float(e.x); o.wt = __uint_as_float(e.y); o.valid = f.x != 0u;
        }
    }
    double* partials = pp.partials;
    fused_block_epilogue(kp, pp, o, bvh_lbq, tid, wave_slot, [=](int a, double v) { partials[(size_t)a * mgrid + lb] = v; });
    if constexpr (XW && ICP_GX) {
        if (kp.gx.slots) {
            // GX: the block's partial is on its way.  Its outbox is as it was found (every posted slot was folded or taken back before the
            // block's sums) -- the header still says otherwise; and if this block walked for long, so do others: wave 0 goes and helps.
            const int* xc = (const int*)(bvh_lbq + ICP_SHARE_ROWS * BVH_THREADS);
            const int posted = xc[20], walked_long = xc[21];
            if (tid == 0 && posted > 0) gx_store((unsigned long long*)kp.gx.hdr + lb, 0ull);
            if (tid < WAVE && walked_long) gx_help<DIM, BVH_THREADS, typename std::conditional<WIDE, unsigned long long, unsigned int>::type>(bv, kp.gx, lb, mgrid, bvh_lbq, tid);
        }
    }
}
template <int DIM, bool WIDE>
__global__ __launch_bounds__(BVH_THREADS, DIM == 3 ? ICP_FUSED_WAVES : 4) void k_knn_bvh_post(const KnnParams kp, const BvhViewT<DIM> bv, const int* __restrict__ qorder, const PostParams pp) {
    const RingParams none{};
    fused_matcher_body<DIM, WIDE, false>(kp, bv, qorder, pp, none);
}
// The merged loop's launch: blocks [0, rp.n_red) = reducer of the previous iteration, the rest = this iteration's matcher (kp.ps = the
// pose slot they wait for).
template <int DIM, bool WIDE>
__global__ __launch_bounds__(BVH_THREADS, DIM == 3 ? ICP_FUSED_WAVES : 4) void k_knn_bvh_post_ring(const KnnParams kp, const BvhViewT<DIM> bv, const int* __restrict__ qorder, const PostParams pp, const RingParams rp) {
    static_assert(BVH_THREADS >= RING_THREADS, "the reducer blocks are the first RING_THREADS threads of a matcher-sized block");
    fused_matcher_body<DIM, WIDE, true>(kp, bv, qorder, pp, rp);
}

// dev_persist.hpp -- the whole ICP loop of one resolution level as ONE launch: k_icp_loop + k_icp_loop_reducer (point-to-plane through the
// fused BVH matcher).  EXPERIMENTAL: on with ICP_HIP_PERSIST=1, measured slower than the merged per-launch loop so far.
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.
// ------------------------------------------------------------------------------------------------
// Why.  With one launch per iteration a converged iteration (nobody walks the tree) pays a launch floor or two, a burst of 76-84 bytes per
// query that re-reads, unchanged, what the previous launch had in registers (source point, normal, search state, the neighbour's
// record), 4.7 us of arithmetic and three dependent trips to memory.  Riding the reducer in front of the next matcher launch (the merged
// loop, dev_solve.hpp) removes one floor, but its hand-overs then queue behind the matcher's load burst (3-5 us per hop on a streaming CU
// instead of 1).  Here the burst goes away: the matcher waves stay resident from one iteration to the next and keep their queries' data
// (parked in LDS during the epilogue); nothing but hand-over granules moves while the run has converged.
//
// Structure.  TWO kernels side by side, on two streams: k_icp_loop -- nb matcher blocks of two waves, all resident at once (the host checks
// nb + LOOP_RED + 1 against the kernel's occupancy and runs one launch per iteration otherwise) -- and k_icp_loop_reducer -- 2 x 34 fold
// blocks and a one-wave solver, small enough (two waves, <= 112 VGPRs) for the holes the matcher grid leaves.  Per iteration g:
//   matcher block   : [loads, unless its waves still hold what they parked] -> waits for ITS replica of pose slot g -> transform / verify /
//                     search / weigh / reject -> block partial (34 doubles) as self-validating 8-byte granules into ring slot g % PRING_DEPTH
//   fold block (a, h): folds half h of row a of that ring slot as the granules arrive (the fold's loads ARE the polls; k_reduce_solve's
//                     summation order, bit for bit), publishes its part of total a into row g of the totals ring, then puts the "empty"
//                     pattern back into the granules it consumed
//   solver          : polls the 3 x 34 partial totals, solves (lane-parallel LDL^T), publishes pose slot g + 1 (every replica), the record
//                     of iteration g and the device clock
// Every hand-over is of the "granule" kind (one naturally aligned 8-byte sc1 store, sc1-load polls, no fences, no flags, no atomics; see
// ring_reduce_solve in dev_solve.hpp for the conventions); the partial ring needs no counter either, because a fold thread only ever
// re-arms granules it has itself consumed, and waits for those stores before it publishes anything a later write to the same granule
// could depend on (causality: re-arm(g) < total(g + 1) < pose(g + 2) < partial(g + 2)).
// Every wait is bounded and watches the abort word: a waiter that gives up raises it (the very first wait of a launch has a short bound:
// it also tells whether the two kernels run side by side at all), the solver raises it together with a fault it publishes (a pivot of
// the 6 x 6 system failed the rank test: the eigen fallback lives in k_reduce_solve only) -- the host then repeats the run with one
// launch per iteration.
#ifndef ICP_PRING_DEPTH
#define ICP_PRING_DEPTH 2
#endif
constexpr int PRING_DEPTH = ICP_PRING_DEPTH;
constexpr int LOOP_LDS_ROWS = ICP_SHARE_ROWS;           // uint2 rows of BVH_THREADS the matcher blocks of k_icp_loop need: the shared walk's
#ifndef ICP_LOOP_WAVES
#define ICP_LOOP_WAVES 6           // waves per SIMD the register allocator must leave room for: the whole grid has to be resident at once
#endif

struct LoopParams {
    int iters;                         // iterations this launch runs
    int first;                         // index of its first iteration in the run: pose slot `first` is what that iteration searches at
    int seed_first;                    // 1: the first iteration is seeded (the launch before matched the same queries)
    PoseState* slots;                  // pose ring of the run: slot g + 1 is published by the reducer of iteration g; POSE_REPLICAS copies of every slot,
                                       // POSE_REPLICA_STRIDE bytes apart (loop_slot): a wave polls ONE of them
    unsigned long long* totals;        // [run iterations][TOTALS_ROW] totals ring: three granules per sum (see loop_reducer)
    unsigned long long* pring;         // [PRING_DEPTH][NSUM_USED][nb] block partials of iteration g in slot g % PRING_DEPTH
    int nb;                            // matcher blocks
    int wave_presleep_eighths;         // a matcher wave sleeps through this many eighths of the last iteration's length (pose to pose) before it polls for the next pose
    int presleep_eighths;              // the reducer sleeps through this many eighths of the last iteration's length before its first poll
    int dictated;                      // 1: every pose slot is filled up front and nobody reduces (icp_match_seeded)
    icp_iter_stats* stats;             // [run iterations] records
    int n_src;
    int* abort_word;                   // != 0: leave
    int record_last;                   // 1: the LAST iteration writes its Match records / distances (pp.matches / kp.d2_out)
    int* dbg; int dbg_iter, dbg_waves; // development builds (ICP_DEBUG_TIMES): phase stamps of iteration dbg_iter: matcher waves at dbg[8 * wave], reducer blocks behind them
    PoseState* final_out; int final_g; // when the solver publishes slot final_g it also leaves that pose state here (beside the records: ONE copy back)
    long long* clocks;                 // [run iterations + 1] the 100 MHz clock when pose slot g became available (block 0; nullptr: not kept)
};

__device__ __forceinline__ int loop_aborted(const LoopParams& L) { return __hip_atomic_load(L.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void loop_abort(const LoopParams& L, int why) { __hip_atomic_store(L.abort_word, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The pose hand-over, matcher side: as ring_wait_pose, with a back-off for the long waits of the early iterations (a wave that has
// finished polls for as long as the slowest walk of the launch lasts) and an eye on the abort word.
__device__ __forceinline__ bool loop_wait_pose(const LoopParams& L, const PoseState* slot, int lane, float (&Pm)[16], float (&Nm)[9]) {
    const unsigned long long* g = (const unsigned long long*)slot;
    unsigned long long v = 0ull;
    bool ok = false;
    for (int spin = 0; spin < SPIN_LIMIT; spin++) {
        if (lane < 16) v = __hip_atomic_load(g + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!__any(lane < 16 && v == GRANULE_EMPTY)) { ok = true; break; }
        if (spin < 8) __builtin_amdgcn_s_sleep(2); else if (spin < 32) __builtin_amdgcn_s_sleep(8); else { __builtin_amdgcn_s_sleep(32); if ((spin & 15) == 0 && loop_aborted(L)) return false; }
    }
    if (!ok) { if (lane == 0) loop_abort(L, 1); return false; }
    const int lo = (int)(unsigned int)v, hi = (int)(unsigned int)(v >> 32);
#pragma unroll
    for (int q = 0; q < 16; q++) Pm[q] = __int_as_float(__builtin_amdgcn_readlane((q & 1) ? hi : lo, q >> 1));
#pragma unroll
    for (int q = 0; q < 9; q++) Nm[q] = __int_as_float(__builtin_amdgcn_readlane(((16 + q) & 1) ? hi : lo, (16 + q) >> 1));
    return __builtin_amdgcn_readlane(hi, 15) == 0;
}

// The reducer: a launch of its own (k_icp_loop_reducer, on a second stream so that it is resident beside the matcher grid).  Inside the
// matcher kernel its granules in flight, their addresses and the solver would have to share the matcher's register budget -- the
// allocator then spills inside the solver's LDL^T loop, on the critical path of every iteration (measured: 7 us for a 2 us solve).
// Its shape is dictated by the holes the matcher grid leaves: 2 895 two-wave blocks on 256 CUs are 11 per CU and a twelfth on 79 of
// them -- 177 CUs keep two wave slots (one on each of two SIMDs that hold five 80-register waves: 112 registers) and ~40 KB of LDS.
// So: blocks of TWO waves, at most 112 VGPRs (tests/test_kernel_budget.py), and two of them per sum -- LOOP_RED = 2 * NSUM_USED blocks;
// block (a, h) is threads 128 h .. 128 h + 127 of k_reduce_solve's 256-thread block a, i.e. its waves 2 h and 2 h + 1: half 0 publishes
// (wave sum 0 + wave sum 1), half 1 publishes wave sums 2 and 3, and block 0 folds the three granules of every sum in k_reduce_solve's
// order, ((w0 + w1) + w2) + w3 -- bit-identical totals.  The fold's loads ARE the polls; what a thread consumed it re-arms.  Block 0
// then solves (lane-parallel LDL^T, point-to-plane) and publishes the pose into every replica of slot g + 1.  A pivot that fails the
// rank test needs the eigen fallback, which only k_reduce_solve carries (131 VGPRs): PoseState::fault = 2 in the published slot, abort
// word raised, the host repeats the run with one launch per iteration.
constexpr int LOOP_RED = 2 * NSUM_USED;
constexpr int TOTALS_ROW = 128;                           // granules per iteration in the totals ring: 3 per sum, padded
// The solver: block LOOP_RED of the reducer grid, ONE wave (its second wave leaves at once: every barrier below is then a barrier of one
// wave, and the lane-parallel solve's dozen of them cost nothing).  Polls the 3 x 34 partial totals, folds them in k_reduce_solve's order,
// solves, publishes the pose into every replica of slot g + 1 and the record of iteration g.
__device__ __forceinline__ void loop_solver(const LoopParams& L) {
    __shared__ double tot[NSUM];
    __shared__ double parts[NSUM_USED][3];
    __shared__ unsigned int slot_words[32];
    const int tid = threadIdx.x;
    if (tid >= WAVE) return;
    if (tid == 0 && L.clocks) L.clocks[L.first] = (long long)wall_clock64();
    for (int it = 0; it < L.iters; it++) {
        const int g = L.first + it;
        const unsigned long long* trow = L.totals + (size_t)g * TOTALS_ROW;
        bool lost = false;
        for (int q = tid; q < 3 * NSUM_USED; q += WAVE) {
            unsigned long long bits = GRANULE_EMPTY;
            for (int spin = 0; spin < SPIN_LIMIT; spin++) {
                bits = __hip_atomic_load(trow + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (bits != GRANULE_EMPTY) break;
                __builtin_amdgcn_s_sleep(2);
                if ((spin & 63) == 63 && loop_aborted(L)) break;
            }
            if (bits == GRANULE_EMPTY) lost = true;
            parts[q / 3][q % 3] = __longlong_as_double((long long)bits);
        }
        LOOP_STAMP(L.dbg_waves + 2 * LOOP_RED, 3);
        const bool give_up = __any(lost);
        __syncthreads();
        if (tid < NSUM) tot[tid] = tid < NSUM_USED ? (parts[tid][0] + parts[tid][1]) + parts[tid][2] : 0.0;
        __syncthreads();
        const PoseState* pin = loop_slot(L.slots, g, 0);    // complete: this wave published it (or the host did, g = first)
        const double n = tot[SUM_N];
        int fault = 0, status = ICP_OK;
        const float* npose = nullptr;
        if (give_up) fault = 1;
        else if (n > 0) { npose = p2plane_lanes_core(tot + SUM_M, pin->pose); if (!npose) fault = 2; }      // (uniform)
        else status = ICP_ERR_NO_CORRESPONDENCES;          // the pose stays (ICPOptimizer.h:668,680: the reference would hang in ASSERT)
        if (tid < 16) slot_words[tid] = __float_as_uint(npose ? npose[tid] : pin->pose[tid]);
        if (tid >= 32 && tid < 41) slot_words[16 + (tid - 32)] = __float_as_uint(npose ? normal_matrix_entry(npose, tid - 32) : pin->nmat[tid - 32]);      // one entry per lane
        if (tid >= 41 && tid < 47) {
            const int k = tid - 41;
            slot_words[25 + k] = __float_as_uint(n > 0 ? (float)(tot[(k < 3 ? SUM_S : SUM_D - 3) + k] / n) : 0.f);
        }
        if (tid == 47) slot_words[31] = (unsigned int)fault;
        __syncthreads();
        if (fault && tid == 0) loop_abort(L, fault);
        if (tid == 0 && L.clocks) L.clocks[g + 1] = (long long)wall_clock64();
        for (int q = tid; q < 16 * POSE_REPLICAS; q += WAVE)          // every replica, 16 granules each
            __hip_atomic_store((unsigned long long*)loop_slot(L.slots, g + 1, q >> 4) + (q & 15), granule_of(slot_words[2 * (q & 15)], slot_words[2 * (q & 15) + 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (L.final_out && g + 1 == L.final_g && tid < 32) ((unsigned int*)L.final_out)[tid] = slot_words[tid];
        LOOP_STAMP(L.dbg_waves + 2 * LOOP_RED, 4);
#if ICP_DEBUG_TIMES
        if (L.dbg && g == L.dbg_iter - 1 && tid == 0) L.dbg[8 * (L.dbg_waves + 2 * LOOP_RED + 2)] = (int)(unsigned int)wall_clock64();      // when the pose of the stamped iteration went out
#endif
        if (L.stats && !fault) {
            icp_iter_stats* st = L.stats + g;
            if (tid < 16) st->pose[tid] = __uint_as_float(slot_words[tid]);
            if (tid == 16) { st->n_src = L.n_src; st->n_valid = (int)n; st->rmse = -1.f; st->benchmark_error = -1.f; st->status = status; }
        }
        if (fault) return;
        __syncthreads();                                   // slot_words / tot are rewritten by the next iteration
    }
}

__global__ __launch_bounds__(RING_THREADS) void k_icp_loop_reducer(const LoopParams L) {
    __shared__ double wsum[2];
    __shared__ int give_up;
    if ((int)blockIdx.x == LOOP_RED) { loop_solver(L); return; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, a = (int)blockIdx.x >> 1, h = (int)blockIdx.x & 1;
    long long t_prev = 0, period = 0;                      // clock (100 MHz) at the end of the last fold, and the time between the last two
    for (int it = 0; it < L.iters; it++) {
        const int g = L.first + it;
        unsigned long long* row = L.pring + ((size_t)(it % PRING_DEPTH) * NSUM_USED + a) * L.nb;
        const int tb = h * RING_THREADS + tid;              // this thread's granules: row[j * 256 + tb]
        // Most of an iteration nothing can have arrived: sleep through a part of the last period (iteration 1 lasts half as long as the
        // unseeded iteration 0; from then on the periods shrink slowly) before the first poll.
        if (period > 0) { const long long until = t_prev + ((period * L.presleep_eighths) >> 3); while ((long long)wall_clock64() < until) __builtin_amdgcn_s_sleep(32); }
        LOOP_STAMP(L.dbg_waves + 2 * (int)blockIdx.x + w, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the re-arm stores of the iteration before (long done): see the header
        if (tid == 0) give_up = 0;
        __syncthreads();
        double x = 0.0;                                     // thread 128 h + tid of k_reduce_solve's fold
        bool lost = false;
        // the very first hand-over of a launch also tells whether the matcher grid runs BESIDE this one at all (two streams): a modest bound
        const int limit = it == 0 ? (1 << 17) : SPIN_LIMIT;
        for (int b0 = 0; b0 < L.nb && !lost; b0 += SOLVE_INFLIGHT * SOLVE_THREADS) {
            // Ask for all of this thread's granules in one round trip; while any is missing, sleep and ask again for those.
            unsigned long long v[SOLVE_INFLIGHT];
#pragma unroll
            for (int j = 0; j < SOLVE_INFLIGHT; j++) { const int b = b0 + j * SOLVE_THREADS + tb; v[j] = b < L.nb ? __hip_atomic_load(row + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull; }
            for (int spin = 0; spin < limit; spin++) {
                bool miss = false;
#pragma unroll
                for (int j = 0; j < SOLVE_INFLIGHT; j++) {
                    if (v[j] == GRANULE_EMPTY) { v[j] = __hip_atomic_load(row + b0 + j * SOLVE_THREADS + tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); miss = true; }
                }
                if (!miss) break;
                if (spin < 64) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(32);
                if ((spin & 31) == 31 && loop_aborted(L)) { lost = true; break; }
                if (spin == limit - 1) { lost = true; loop_abort(L, 1); }
            }
#pragma unroll
            for (int j = 0; j < SOLVE_INFLIGHT; j++) { const int b = b0 + j * SOLVE_THREADS + tb; if (b < L.nb) x += __longlong_as_double((long long)v[j]); }
        }
        LOOP_STAMP(L.dbg_waves + 2 * (int)blockIdx.x + w, 1);
        if (lost) give_up = 1;
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
        if (lane == 0) wsum[w] = x;
        __syncthreads();
        if (give_up) return;                               // (uniform: raised before the barrier)
        { const long long now = (long long)wall_clock64(); if (t_prev) period = now - t_prev; t_prev = now; }
        unsigned long long* trow = L.totals + (size_t)g * TOTALS_ROW;
        if (tid == 0) {
            if (h == 0) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(wsum[0] + wsum[1]);
                __hip_atomic_store(trow + 3 * a, granule_of((unsigned int)bits, (unsigned int)(bits >> 32)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const unsigned long long b2 = (unsigned long long)__double_as_longlong(wsum[0]), b3 = (unsigned long long)__double_as_longlong(wsum[1]);
                __hip_atomic_store(trow + 3 * a + 1, granule_of((unsigned int)b2, (unsigned int)(b2 >> 32)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(trow + 3 * a + 2, granule_of((unsigned int)b3, (unsigned int)(b3 >> 32)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        LOOP_STAMP(L.dbg_waves + 2 * (int)blockIdx.x + w, 2);
        // the consumed granules back to "empty" -- behind the total in the memory pipeline, not in front of it
        for (int b = tb; b < L.nb; b += SOLVE_THREADS) __hip_atomic_store(row + b, GRANULE_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// What a wave carries into the next iteration sits out the epilogue (whose transposing reduction is the other register peak of the
// kernel) in LDS: rows 1..9 of the shared walk's records -- thread t owns element r * BVH_THREADS + t of every row r; row 0 is left
// alone, the block reduction uses its first 66 elements -- 18 dwords: source point and normal, anchor + bound, the neighbour's 32-byte
// record (the neighbour's position stays in a register).  Written and read back by the same thread; a walk (this wave's or, through the same rows, only this wave's helpers) starts
// after the read.  DIM == 6 carries nothing (ten dwords more than the rows hold): it reloads.
template <int DIM>
__device__ __forceinline__ void loop_park(uint2* __restrict__ lbq, int tid, const QueryIn<DIM>& in) {
    constexpr int NT = BVH_THREADS;
    lbq[1 * NT + tid] = make_uint2(__float_as_uint(in.r0), __float_as_uint(in.r1));
    lbq[2 * NT + tid] = make_uint2(__float_as_uint(in.r2), __float_as_uint(in.rn0));
    lbq[3 * NT + tid] = make_uint2(__float_as_uint(in.rn1), __float_as_uint(in.rn2));
    lbq[4 * NT + tid] = make_uint2(__float_as_uint(in.st.x), __float_as_uint(in.st.y));
    lbq[5 * NT + tid] = make_uint2(__float_as_uint(in.st.z), __float_as_uint(in.st.w));
    lbq[6 * NT + tid] = make_uint2(__float_as_uint(in.ra.x), __float_as_uint(in.ra.y));
    lbq[7 * NT + tid] = make_uint2(__float_as_uint(in.ra.z), __float_as_uint(in.ra.w));
    lbq[8 * NT + tid] = make_uint2(__float_as_uint(in.rb.x), __float_as_uint(in.rb.y));
    lbq[9 * NT + tid] = make_uint2(__float_as_uint(in.rb.z), __float_as_uint(in.rb.w));
}
template <int DIM>
__device__ __forceinline__ void loop_unpark(const uint2* __restrict__ lbq, int tid, QueryIn<DIM>& in) {
    constexpr int NT = BVH_THREADS;
    const uint2 a = lbq[1 * NT + tid], b = lbq[2 * NT + tid], c = lbq[3 * NT + tid], d = lbq[4 * NT + tid], e = lbq[5 * NT + tid];
    const uint2 f = lbq[6 * NT + tid], g = lbq[7 * NT + tid], h = lbq[8 * NT + tid], i = lbq[9 * NT + tid];
    in.r0 = __uint_as_float(a.x); in.r1 = __uint_as_float(a.y); in.r2 = __uint_as_float(b.x); in.rn0 = __uint_as_float(b.y); in.rn1 = __uint_as_float(c.x); in.rn2 = __uint_as_float(c.y);
    in.st.x = __uint_as_float(d.x); in.st.y = __uint_as_float(d.y); in.st.z = __uint_as_float(e.x); in.st.w = __uint_as_float(e.y);
    in.ra.x = __uint_as_float(f.x); in.ra.y = __uint_as_float(f.y); in.ra.z = __uint_as_float(g.x); in.ra.w = __uint_as_float(g.y);
    in.rb.x = __uint_as_float(h.x); in.rb.y = __uint_as_float(h.y); in.rb.z = __uint_as_float(i.x); in.rb.w = __uint_as_float(i.y);
    in.c3 = 0.f; in.c4 = 0.f; in.c5 = 0.f; in.j0 = -1;
#pragma unroll
    for (int q = 0; q < DIM; q++) in.tq[q] = 0.f;
    if (DIM == 3) { in.tq[0] = in.ra.x; in.tq[1] = in.ra.y; in.tq[2] = in.ra.z; in.j0 = __float_as_int(in.ra.w); }
}

// The launch's argument: FEW pointers.  Inlined into a loop, the one-launch-per-iteration code has a problem the straight kernel does not
// have: nothing the loop uses changes from iteration to iteration, so the optimiser moves it all in front of the loop -- ~50 arguments
// fetched once and kept in scalar registers (there are ~100; the rest spills, or ends up in vector registers), the address of every
// per-query array and LDS row computed once and kept alive across the walk: 105-119 VGPRs against the 76 of the same code as a launch
// of its own -- a wave per SIMD less, and the grid no longer fits the device.  What is measured to work:
//   * the per-query arrays of a level live in TWO allocations, planes a fixed number of elements apart (source: x y z nx ny nz cr cg cb
//     rgba; search state: neighbour position | anchor + bound | second-tier bound), so that the kernel is handed two base pointers and
//     two strides instead of seventeen pointers;
//   * once per iteration the strides and the query index pass through an empty asm statement: integers the optimiser cannot see
//     through, so every address is computed where it is used -- while the POINTERS stay plain kernel arguments, known to point to
//     global memory (a pointer that went through such a statement, or was fetched from an argument block in memory, is a generic one:
//     flat loads, no scalar base + 32-bit offset addressing, 64-bit addresses per lane);
//   * -mllvm -disable-machine-licm for the translation unit (constants and address arithmetic hoisted after instruction selection).
template <int DIM> struct LoopK {
    const float* src; int src_stride;              // source level: plane p of point i at src[p * src_stride + i]; planes 6..9 only with colours
    char* qpack; int q_cap;                        // search state: int nn_raw[q_cap] | float4 qstate[q_cap] | float2 qstate2[q_cap]  (q_cap a multiple of 64)
    const BvhLeafT<DIM>* leaves; const BvhQuadT<DIM>* qnodes; const TgtRec* recs; int Lq, n_valid;
    icp_match_t* matches; float* d2_out;           // records of the LAST iteration (LoopParams::record_last), nullptr otherwise
    int* dbg_steps;                                // development builds
    int n; float max_dist; int incremental, tier2;
    int metric, weighting, rejection; float cos_reject;
    LoopParams L;
};

// Matcher block: all iterations of the launch.  `have`: what the wave parked in LDS is what memory holds for its queries (nothing of
// their search state was rewritten in the iteration before): no loads at all.
template <int DIM, bool WIDE>
__device__ __forceinline__ void loop_matcher(const LoopK<DIM>& K) {
    extern __shared__ uint2 bvh_lbq[];                    // [ICP_SHARE_ROWS][BVH_THREADS]: the shared walk's records; reused by the reduction
    constexpr int NW = BVH_THREADS / WAVE;
    const LoopParams& L = K.L;
    const int tid_fixed = threadIdx.x;
    const int lb = xcd_contiguous_block((int)blockIdx.x, L.nb);            // partial slot = logical block -> fixed summation order
    const int wave_slot = fused_wave_slot(lb, tid_fixed >> 6, L.nb);      // (the mapping of fused_matcher_body: same block partials, same sums)
    const int t = wave_slot * BVH_QPW + (tid_fixed & 63);
    const int k_fixed = t < K.n ? t : -1;                // (sorted levels: the query index is the position)
    int q0_kept = -1;                                      // (the one carried value that stays in a register)
    bool have = false;
    unsigned long long renew_mask = 0ull;                  // lanes whose query took the two-leaf tier last time: it will again (it sits between two targets)
    int t_pose = 0, period = 0;                            // 100 MHz clock (low word) when the last pose arrived, and the time between the last two arrivals
    for (int it = 0; it < L.iters; it++) {
        int k = k_fixed, tid = tid_fixed, S = K.src_stride, Q = K.q_cap;
        asm volatile("" : "+v"(k), "+v"(tid), "+s"(S), "+s"(Q));     // see LoopK
        tid &= BVH_THREADS - 1;                            // (what the optimiser knew about the thread index before)
        const int lane = tid & 63;
        // the per-iteration view of the arguments, in the shape the shared code of the fused matcher takes them
        KnnParams kp; PostParams pp; BvhViewT<DIM> bv;
        kp.sx = K.src; kp.sy = K.src + S; kp.sz = K.src + 2 * (size_t)S;
        pp.snx = K.src + 3 * (size_t)S; pp.sny = K.src + 4 * (size_t)S; pp.snz = K.src + 5 * (size_t)S;
        kp.scr = K.src + 6 * (size_t)S; kp.scg = K.src + 7 * (size_t)S; kp.scb = K.src + 8 * (size_t)S; pp.srgba = (const uint32_t*)(K.src + 9 * (size_t)S);
        kp.sel = nullptr; pp.sel = nullptr; kp.n = K.n; kp.max_dist = K.max_dist; kp.incremental = K.incremental; kp.dbg_steps = nullptr; kp.dbg_waves = 0;
        kp.nn_raw = (int*)K.qpack; kp.qstate = (float4*)(K.qpack + 4 * (size_t)Q); kp.qstate2 = K.tier2 ? (float2*)(K.qpack + 20 * (size_t)Q) : nullptr;
        pp.metric = K.metric; pp.weighting = K.weighting; pp.rejection = K.rejection; pp.max_dist = K.max_dist; pp.cos_reject = K.cos_reject;
        bv.leaves = K.leaves; bv.qnodes = K.qnodes; bv.recs = K.recs; bv.Lq = K.Lq; bv.n_valid = K.n_valid;
        const int g = L.first + it;
        const bool seeded = it > 0 || L.seed_first != 0, inc = kp.incremental && seeded;
        const bool rec = L.record_last && it == L.iters - 1;
        pp.matches = rec ? K.matches : nullptr; kp.d2_out = rec ? K.d2_out : nullptr;
        LOOP_STAMP(wave_slot, 0);
        QueryIn<DIM> in;
        if (!have) {
            fused_front_loads<DIM>(kp, pp, bv, k, seeded, inc, in);
            if (DIM == 3 && inc) { loop_park<DIM>(bvh_lbq, tid, in); q0_kept = in.q0; }      // (a walk of this wave overwrites the rows: then nobody reads them back)
        } else { loop_unpark<DIM>(bvh_lbq, tid, in); in.q0 = q0_kept; }
        // A query of the two-leaf tier pays three dependent trips to memory behind the pose (second-tier bound -> two leaves), and with it
        // its wave and the wave it shares the block barrier with: the last ones of every converged iteration by 5 us (tools/dev_loop_times.py).
        // The same lines can be asked for NOW, while the wave waits for the pose anyway: afterwards they are cache hits.
        if (have && ((renew_mask >> lane) & 1ull) && kp.qstate2) {
            const float2 st2 = kp.qstate2[k];
            const int l2 = __float_as_int(st2.y);
            int touch = *(const int*)(bv.leaves + (in.q0 >> 3));
            if (l2 >= 0) touch |= *(const int*)(bv.leaves + l2);
            asm volatile("" :: "v"(touch));
        }
        // A wave that is done long before the slowest walk of the launch would poll all the while (the early iterations: 5 000 waves, 100 us);
        // it sleeps through most of what the last iteration took instead -- iterations get shorter slowly, and a wave that oversleeps costs
        // only itself -- and polls from there.
        if (period > 0 && L.wave_presleep_eighths > 0) {
            const int until = t_pose + ((period >> 3) * L.wave_presleep_eighths);
            while ((int)((unsigned int)wall_clock64() - (unsigned int)until) < 0) __builtin_amdgcn_s_sleep(16);
        }
        float Pm[16], Nm[9];
        if (!loop_wait_pose(L, loop_slot(L.slots, g, L.dictated ? 0 : (int)((blockIdx.x * 2u + (unsigned int)(tid >> 6)) % (unsigned int)POSE_REPLICAS)), lane, Pm, Nm)) return;
        { const int now = (int)(unsigned int)wall_clock64(); if (it > 0) period = now - t_pose; t_pose = now; }
        LOOP_STAMP(wave_slot, 1);
        PairOut o; bool searched, renewed;
        fused_search_post<DIM, WIDE>(kp, bv, pp, k, seeded, inc, Pm, Nm, in, bvh_lbq, tid, wave_slot, o, searched, &renewed);
        LOOP_STAMP(wave_slot, 2);
#if ICP_DEBUG_TIMES
        { const int ns_ = __popcll(__ballot(searched)), nr_ = __popcll(__ballot(renewed)); if (L.dbg && g == L.dbg_iter && lane == 0) { L.dbg[8 * wave_slot + 5] = ns_ | (nr_ << 8) | ((have ? 1 : 0) << 16); } }
#endif
        // (a query of the two-leaf tier -- a handful sit between two targets and take it in EVERY iteration -- gives its new state back: parked
        //  again, the wave keeps everything; reloading it all from memory made those few waves the last of every converged iteration by 6 us)
        have = DIM == 3 && inc && !__any(searched);
        renew_mask = __ballot(renewed);
        if (have && renewed) { loop_park<DIM>(bvh_lbq, tid, in); q0_kept = in.q0; }
        unsigned long long* prow = L.pring + (size_t)(it % PRING_DEPTH) * NSUM_USED * L.nb;
        const int nb = L.nb;
        fused_block_epilogue(kp, pp, o, bvh_lbq, tid, wave_slot, [=](int sum, double v) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            __hip_atomic_store(prow + (size_t)sum * nb + lb, granule_of((unsigned int)bits, (unsigned int)(bits >> 32)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
        LOOP_STAMP(wave_slot, 3);
        __syncthreads();                                   // the reduction's LDS rows become the next iteration's traversal records
        LOOP_STAMP(wave_slot, 4);
    }
}

template <int DIM, bool WIDE>
__global__ __launch_bounds__(BVH_THREADS, ICP_LOOP_WAVES) void k_icp_loop(const LoopK<DIM> K) {
    loop_matcher<DIM, WIDE>(K);
}

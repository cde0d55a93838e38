// dev_knn_brute.hpp -- exact brute-force 1-NN (k_knn_brute, k_knn_finalize).
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.
// ------------------------------------------------------------------------------------------------
// Exact brute-force 1-NN.  One lane = one query; the target planes are wave-uniform and reach the
// VALU as SGPR operands (s_load_dwordx16 per coordinate per chunk -- no LDS, no VGPR staging).
// Hot loop per PAIR of targets: 5 v_pk_add_f32 + 3 v_pk_mul_f32 + v_min3_f32 (no index tracking);
// a chunk whose minimum beats the lane's running best is rescanned with the reference's
// sequential strict-< loop, which alone defines the result (the packed pass is only a filter:
// v_pk_* and scalar ops round identically, so it has no false negatives).
// Block = 4 waves sharing the same 64 queries; wave w scans quarter w of the block's target
// segment; blockIdx.y splits the target range further for small query counts (merged with a
// packed (d2 bits, index) 64-bit atomicMin = lexicographic first-minimum).
struct GxParams {               // hand-over of parked subtrees between the blocks of a fused BVH matcher launch (dev_bvh.hpp, GX)
    unsigned long long* slots;   // [blocks][GX_SLOTS][GX_GRANULES]; nullptr: off
    unsigned int* hdr;           // [blocks][2]: slots posted so far, lowest slot that may still be unclaimed
    int start_trips;             // a wave posts once its walk has lasted this many passes of the hand-over loop
    int empty_rounds;            // a helper leaves after this many probe rounds without a claim
};
struct KnnParams {
    const float* sx; const float* sy; const float* sz;       // source planes (untransformed unless pretransformed)
    const float* scr; const float* scg; const float* scb;    // source colour features (DIM=6)
    const int* sel;                                          // optional selection (multires); nullptr = identity
    int n;                                                   // queries
    const float* tx; const float* ty; const float* tz;       // target planes, padded with +inf to mpad
    const float* tcr; const float* tcg; const float* tcb;
    int mpad;                                                // multiple of KNN_CH
    const PoseState* ps; int pretransformed;
    float max_dist;
    icp_match_t* out; float* d2_out;                           // direct outputs (nseg == 1)
    unsigned long long* best64;                              // packed partial results (nseg > 1)
    int nseg;
    int* nn_raw;                                             // [n] position (8 * leaf + slot) of the nearest target of this launch (BVH backend), seed of the next one
    int use_prev;                                            // 1: nn_raw holds the previous iteration's result for the same queries
    float4* qstate;                                          // [n] (query xyz when last searched or verified, lower bound on the distance to every OTHER target)
    int incremental;                                         // 1: verify-and-skip with qstate (needs use_prev)
    float2* qstate2;                                         // [n] (lower bound, at the same anchor, on every target outside the neighbour's leaf and the runner-up's leaf; that second leaf as int bits, -1: none)
    int* dbg_steps;                                          // development builds (ICP_DEBUG_STEPS): [n] nodes | leaves << 16 visited by the walk of query k; nullptr otherwise
    int* fault;                                              // fused BVH matcher: raised when a bounded wait of the cross-wave hand-over runs out (cannot happen; knn_walk_shared)
    GxParams gx;                                             // fused BVH matcher: outboxes of the hand-over between blocks (dev_bvh.hpp, GX); slots == nullptr: off
    int dbg_waves;                                           // development builds (ICP_DEBUG_TIMES): waves of the launch = where the per-query records start in dbg_steps
};

template <int DIM>
__global__ __launch_bounds__(256) void k_knn_brute(const KnnParams kp) {
    __shared__ float sd[4][WAVE];
    __shared__ int si[4][WAVE];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.x * WAVE + lane;
    const int kk = k < kp.n ? k : kp.n - 1;
    const int i = kp.sel ? kp.sel[kk] : kk;
    float px = kp.sx[i], py = kp.sy[i], pz = kp.sz[i];
    if (!kp.pretransformed) { float a, b, c; xform_point(kp.ps->pose, px, py, pz, a, b, c); px = a; py = b; pz = c; }
    float pr = 0.f, pg = 0.f, pb = 0.f;
    if (DIM == 6) { pr = kp.scr[i]; pg = kp.scg[i]; pb = kp.scb[i]; }
    const f2 px2 = {px, px}, py2 = {py, py}, pz2 = {pz, pz};
    const f2 pr2 = {pr, pr}, pg2 = {pg, pg}, pb2 = {pb, pb};

    const int nch = kp.mpad / KNN_CH;
    const int s0 = (int)(((long long)nch * blockIdx.y) / kp.nseg), s1 = (int)(((long long)nch * (blockIdx.y + 1)) / kp.nseg);
    const int c0 = s0 + ((s1 - s0) * w) / 4, c1 = s0 + ((s1 - s0) * (w + 1)) / 4;

    float best = FLT_MAX; int bi = -1;
    for (int c = c0; c < c1; c++) {
        const int j0 = c * KNN_CH;
        float mm = FLT_MAX;
#pragma unroll
        for (int t = 0; t < KNN_CH; t += 2) {
            f2 qx = *(const f2*)(kp.tx + j0 + t), qy = *(const f2*)(kp.ty + j0 + t), qz = *(const f2*)(kp.tz + j0 + t);
            f2 dx = px2 - qx, dy = py2 - qy, dz = pz2 - qz;
            f2 s = (dx * dx + dy * dy) + dz * dz;
            if (DIM == 6) {
                f2 qr = *(const f2*)(kp.tcr + j0 + t), qg = *(const f2*)(kp.tcg + j0 + t), qb = *(const f2*)(kp.tcb + j0 + t);
                f2 dr = pr2 - qr, dg = pg2 - qg, db = pb2 - qb;
                s = ((s + dr * dr) + dg * dg) + db * db;
            }
            mm = fminf(fminf(mm, s.x), s.y);
        }
        if (mm < best) {
            for (int t = 0; t < KNN_CH; t++) {
                float dx = px - kp.tx[j0 + t], dy = py - kp.ty[j0 + t], dz = pz - kp.tz[j0 + t];
                float d = (dx * dx + dy * dy) + dz * dz;
                if (DIM == 6) {
                    float dr = pr - kp.tcr[j0 + t], dg = pg - kp.tcg[j0 + t], db = pb - kp.tcb[j0 + t];
                    d = ((d + dr * dr) + dg * dg) + db * db;
                }
                if (d < best) { best = d; bi = j0 + t; }       // strict: first minimum (NearestNeighbor.h:87)
            }
        }
    }
    sd[w][lane] = best; si[w][lane] = bi;
    __syncthreads();
    if (w == 0 && k < kp.n) {
#pragma unroll
        for (int v = 1; v < 4; v++) { float d = sd[v][lane]; int j = si[v][lane]; if (d < best) { best = d; bi = j; } }
        if (kp.nseg == 1) {
            icp_match_t m;
            if (best <= kp.max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }   // NearestNeighbor.h:93-96,182-185
            kp.out[k] = m;
            if (kp.d2_out) kp.d2_out[k] = best;
        } else if (bi >= 0) {
            unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned int)bi;
            atomicMin(kp.best64 + k, key);
        }
    }
}

__global__ void k_knn_finalize(const unsigned long long* __restrict__ best64, int n, float max_dist,
                               icp_match_t* __restrict__ out, float* __restrict__ d2_out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    unsigned long long key = best64[k];
    float best = __uint_as_float((unsigned int)(key >> 32));
    int bi = (int)(unsigned int)(key & 0xFFFFFFFFu);
    if (bi == -1) best = FLT_MAX;
    icp_match_t m;
    if (best <= max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }
    out[k] = m;
    if (d2_out) d2_out[k] = best;
}

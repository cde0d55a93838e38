// =====================================================================================
// icp_device.hpp -- hand-written HIP kernels of the ICP hot path for gfx950 (MI355X, wave64).
//
// Build contract: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math): every fp32
// result is one IEEE rounding per operation, in the operation order of the CPU restatement
// (oracle/icp_oracle.cpp), so match indices / distances / weights are bit-identical to it.
// fp32 sqrt and divide are correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
//
// Kernel map (reference file:line relative to icp-variants/ of the reference):
//   k_deinterleave      AoS -> SoA upload conversion (+ colour features NearestNeighbor.h:212-221)
//   k_knn_brute<DIM>    transformPoints (utils.h:106-118) fused with exact 1-NN, first-minimum argmin
//                       (NearestNeighbor.h:81-97 semantics, squared L2 + squared threshold :181-185);
//                       DIM=6 adds rgb/255 (NearestNeighbor.h:209-303)
//   k_knn_finalize      merges target-split partial results (packed u64 atomicMin) into Match records
//   k_bvh_* / k_knn_bvh exact LBVH index (build once per pair = buildIndex, NearestNeighbor.h:122-141) and its query:
//                       bit-identical argmin to k_knn_brute at O(log M) per query
//   k_projective        NearestNeighborSearchProjective::queryMatches (NearestNeighbor.h:333-421)
//   k_post              transformNormals (utils.h:122-133) + applyWeights (weighting.h:39-99) +
//                       pruneCorrespondences (ICPOptimizer.h:157-174) + validity filter (:594-610) +
//                       normal-equation / moment accumulation (ICPOptimizer.h:676-751, ProcrustesAligner.h:43-55)
//   k_sym_accumulate    second pass of the symmetric objective with the means (ICPOptimizer.h:797-853)
//   k_reduce_solve      fixed-order reduction of block partials + fp64 solve + pose composition
//                       (ICPOptimizer.h:614-620,753-781,855-897; ProcrustesAligner.h:56-66)
//   k_rmse_partial      ConvergenceMeasure::rmseAlignmentError (ConvergenceMeasure.h:50-66)
// =====================================================================================
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "../../include/icp_hip.h"

namespace icpdev {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;
constexpr int KNN_CH = 16;          // targets per filter chunk (one s_load_dwordx16 per coordinate)
constexpr int NSUM = 40;            // doubles per block partial (34 used)
constexpr int SUM_N = 0, SUM_S = 1, SUM_D = 4, SUM_M = 7;   // count, sum s, sum d, metric-specific block
constexpr int POST_THREADS = 256;

// Device-resident pose: column-major 4x4 (Eigen layout) + row-major (R^-1)^T for the normals.
struct PoseState {
    float pose[16];
    float nmat[9];
    float mean_s[3];       // unweighted means of the current valid correspondences (symmetric ICP)
    float mean_d[3];
    float pad;
};

struct SoA3 { const float* x; const float* y; const float* z; };

__device__ __forceinline__ bool finite3(float a, float b, float c) {
    return isfinite(a) && isfinite(b) && isfinite(c);
}

// utils.h:113-115 : ((R_i0*x + R_i1*y) + R_i2*z) + t_i  (sequential, fp32, no contraction)
__device__ __forceinline__ void xform_point(const float* __restrict__ P, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = ((P[0] * x + P[4] * y) + P[8] * z) + P[12];
    oy = ((P[1] * x + P[5] * y) + P[9] * z) + P[13];
    oz = ((P[2] * x + P[6] * y) + P[10] * z) + P[14];
}
// utils.h:128-130 with the hoisted (R^-1)^T
__device__ __forceinline__ void xform_normal(const float* __restrict__ N, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = (N[0] * x + N[1] * y) + N[2] * z;
    oy = (N[3] * x + N[4] * y) + N[5] * z;
    oz = (N[6] * x + N[7] * y) + N[8] * z;
}

// ------------------------------------------------------------------------------------------------
// AoS (N x 3 fp32) -> SoA planes.  pad_to > n fills [n, pad_to) with pad_value (+inf for targets so a
// padded slot can never win the argmin).
__global__ void k_deinterleave3(const float* __restrict__ aos, int n, int pad_to, float pad_value,
                                float* __restrict__ x, float* __restrict__ y, float* __restrict__ z) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { x[i] = aos[(size_t)i * 3]; y[i] = aos[(size_t)i * 3 + 1]; z[i] = aos[(size_t)i * 3 + 2]; }
    else if (i < pad_to) { x[i] = pad_value; y[i] = pad_value; z[i] = pad_value; }
}
// RGBA bytes -> packed u32 + colour features (color_scale*color_normalize)*float(c), NearestNeighbor.h:212-221
__global__ void k_colors(const uint8_t* __restrict__ rgba, int n, int pad_to, uint32_t* __restrict__ packed,
                         float* __restrict__ cr, float* __restrict__ cg, float* __restrict__ cb) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float color_normalize = 1 / float(255);
    const float color_scale = 1;
    if (i < n) {
        uint32_t v = ((const uint32_t*)rgba)[i];
        packed[i] = v;
        cr[i] = color_scale * color_normalize * (float)(int)(v & 0xFF);
        cg[i] = color_scale * color_normalize * (float)(int)((v >> 8) & 0xFF);
        cb[i] = color_scale * color_normalize * (float)(int)((v >> 16) & 0xFF);
    } else if (i < pad_to) { cr[i] = 0.f; cg[i] = 0.f; cb[i] = 0.f; }
}

__global__ void k_fill_u64(unsigned long long* p, int n, unsigned long long v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

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
    int* work_items; int* work_n;                            // two-pass incremental search: queries that failed verification (list, count)
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

// ------------------------------------------------------------------------------------------------
// Exact kd-ordered BVH 1-NN: the index the reference builds once per pair (NearestNeighbor.h:122-141 / :209-232, a FLANN
// kd-tree over xyz or over the 6-D xyz+rgb/255 features) rebuilt on the device as a balanced kd-tree in implicit heap
// layout, queried with the SAME fp32 distance and the same lexicographic (d2, lowest index) argmin as k_knn_brute<DIM> --
// bit-identical results, O(log M) nodes per query instead of M distance evaluations.  DIM = 3 or 6.
//   build : level by level, every node's points are sorted along the widest axis of the node's bounding box
//           (one rocPRIM sort per level over keys (node id << 32 | ordered coordinate bits)); the implicit node k
//           covers a fixed, leaf-aligned slice of the array, so the count-balanced median split is simply
//           "first half / second half".  Leaves hold BVH_LEAF points SoA + original indices; node records hold BOTH
//           child boxes, pair-interleaved for packed-f32 math, and are filled bottom-up.
//   query : one lane = one query, depth-first "near child first".  A node is skipped only if its box lower bound
//           exceeds the running best; the bound uses the same operation sequence as the point distance, so by
//           monotonicity of IEEE rounding fl(d2(point)) >= fl(lb) for every point in the box (a 1e-5 relative margin is
//           kept on top).  Equal distances resolve to the lowest original index, exactly like the strict-< scan
//           (NearestNeighbor.h:87).
#ifndef ICP_PREFETCH_PATH
#define ICP_PREFETCH_PATH 1
#endif
constexpr int BVH_LEAF = 8;
constexpr int BVH_THREADS = 128;

template <int DIM> struct BvhNodeT { float lo[DIM][2]; float hi[DIM][2]; float pad[DIM == 3 ? 4 : 8]; };   // 64 B / 128 B
template <int DIM> struct BvhLeafT { float c[DIM][BVH_LEAF]; int idx[BVH_LEAF]; float pad[DIM == 3 ? 0 : 8]; };   // 128 B / 256 B
template <> struct BvhLeafT<3> { float c[3][BVH_LEAF]; int idx[BVH_LEAF]; };
typedef BvhNodeT<3> BvhNode;
typedef BvhLeafT<3> BvhLeaf;

// 4-wide node of the same tree with two binary levels collapsed: the boxes of the four grandchildren, SoA per axis (two
// packed-f32 pairs each).  96 B / 192 B.  Half the dependent loads per query of the binary walk -- the search is bound by
// the latency of that chain, not by bytes or flops.  128 B / 256 B.
template <int DIM> struct BvhQuadT { float lo[DIM][4]; float hi[DIM][4]; float pad[DIM == 3 ? 8 : 16]; };   // padded to one / two 128-byte lines

// Everything the loop needs about a matched target point in ONE 32-byte record, stored in kd (leaf) order -- position
// pos = 8 * leaf + slot.  Neighbouring (Morton-sorted) queries match neighbouring positions, so the gather of the
// correspondence (point, normal, colour) is one sector per query instead of seven scattered planes.
struct TgtRec { float x, y, z; int idx; float nx, ny, nz; uint32_t rgba; };

template <int DIM> struct CoordPtrs { const float* c[DIM]; };

template <int DIM> struct BvhViewT {
    const BvhLeafT<DIM>* leaves;  // [max(n_leaves,1)] kd-ordered points, 8 per leaf; pads are +inf with index -1
    const BvhNodeT<DIM>* nodes;   // [Lp - 1] internal nodes in heap order (node k: children 2k+1, 2k+2; leaves start at Lp-1)
    const TgtRec* recs;           // [8 * max(n_leaves,1)] point + normal + colour + original index by position
    const BvhQuadT<DIM>* qnodes;  // [(4^Lq - 1) / 3] 4-wide nodes, level l at offset (4^l - 1) / 3; the children of level Lq - 1 are the leaves
    int Lq;                       // 4-wide levels = ceil(log2(Lp) / 2)  (an odd binary depth gets a virtual root with one empty half)
    int n_valid;                  // finite target points in the tree
    int Lp;                       // leaves rounded up to a power of two
    CoordPtrs<DIM> tgt;           // target planes by original index (seeding)
};

__device__ __forceinline__ unsigned long long spread21(unsigned int v) {   // 21 bits -> every third bit
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__device__ __forceinline__ unsigned int ordered_bits(float f) {          // monotone float -> uint map
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// Per-level bounding boxes of the nodes, without contended atomics:
//   k_bvh_wave_boxes : every wave (or aligned sub-wave segment of 32 / 16 positions, for the last levels) reduces the box
//                      of its consecutive positions with a shuffle tree -> segbox[segment][2*DIM]
//   k_bvh_node_boxes : one wave per node folds the node's wave boxes (segments of >= 64 positions are wave-aligned)
template <int DIM>
__global__ void k_bvh_wave_boxes(const CoordPtrs<DIM> cp, const int* __restrict__ perm, int n_valid, int seg_shift /* <= 6 */, unsigned int* __restrict__ segbox) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < n_valid;
    const int j = act ? perm[i] : 0;
    unsigned int v[2 * DIM];
#pragma unroll
    for (int k = 0; k < DIM; k++) { const unsigned int a = act ? ordered_bits(cp.c[k][j]) : 0u; v[k] = act ? a : 0xFFFFFFFFu; v[DIM + k] = a; }
    const int seg = 1 << seg_shift;                       // 64 (whole wave) or a sub-wave segment of 32 / 16 positions
    for (int off = seg >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < DIM; k++) { v[k] = min(v[k], (unsigned int)__shfl_down((int)v[k], off, 64)); v[DIM + k] = max(v[DIM + k], (unsigned int)__shfl_down((int)v[DIM + k], off, 64)); }
    }
    if ((threadIdx.x & (seg - 1)) == 0 && (i < n_valid || seg == 64)) {
        unsigned int* o = segbox + (size_t)(i >> seg_shift) * 2 * DIM;
#pragma unroll
        for (int k = 0; k < 2 * DIM; k++) o[k] = v[k];
    }
}
template <int DIM>
__global__ void k_bvh_node_boxes(const unsigned int* __restrict__ wavebox, int n_waves, int waves_per_node_shift, int n_nodes, unsigned int* __restrict__ boxes) {
    const int node = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (node >= n_nodes) return;
    const int w0 = node << waves_per_node_shift, w1 = min(w0 + (1 << waves_per_node_shift), n_waves);
    unsigned int v[2 * DIM];
#pragma unroll
    for (int k = 0; k < DIM; k++) { v[k] = 0xFFFFFFFFu; v[DIM + k] = 0u; }
    for (int w = w0 + lane; w < w1; w += 64) {
        const unsigned int* b = wavebox + (size_t)w * 2 * DIM;
#pragma unroll
        for (int k = 0; k < DIM; k++) { v[k] = min(v[k], b[k]); v[DIM + k] = max(v[DIM + k], b[DIM + k]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < DIM; k++) { v[k] = min(v[k], (unsigned int)__shfl_down((int)v[k], off, 64)); v[DIM + k] = max(v[DIM + k], (unsigned int)__shfl_down((int)v[DIM + k], off, 64)); }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 2 * DIM; k++) boxes[(size_t)node * 2 * DIM + k] = v[k];
    }
}
// sort key of every point at this level: (node id, coordinate along the node's widest axis)
template <int DIM>
__global__ void k_bvh_level_keys(const CoordPtrs<DIM> cp, const int* __restrict__ perm, int n_valid, int seg_shift, const unsigned int* __restrict__ boxes,
                                 unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_valid) return;
    const int node = i >> seg_shift;
    const unsigned int* b = boxes + (size_t)node * 2 * DIM;
    int axis = 0; float ext = -1.f;
#pragma unroll
    for (int k = 0; k < DIM; k++) { const float e = from_ordered_bits(b[DIM + k]) - from_ordered_bits(b[k]); if (e > ext) { ext = e; axis = k; } }
    const int j = perm[i];
    float c = cp.c[0][j];
#pragma unroll
    for (int k = 1; k < DIM; k++) c = (axis == k) ? cp.c[k][j] : c;
    keys[i] = ((unsigned long long)(unsigned int)node << 32) | ordered_bits(c);
}

template <int DIM>
__global__ void k_bvh_gather(const CoordPtrs<DIM> cp, const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz, const uint32_t* __restrict__ rgba,
                             const int* __restrict__ sorted_idx, int n_valid, int n_slots, BvhLeafT<DIM>* __restrict__ leaves, TgtRec* __restrict__ recs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    BvhLeafT<DIM>* lf = leaves + (i / BVH_LEAF); const int t = i % BVH_LEAF;
    TgtRec r; r.x = INFINITY; r.y = INFINITY; r.z = INFINITY; r.idx = -1; r.nx = 0.f; r.ny = 0.f; r.nz = 0.f; r.rgba = 0u;
    if (i < n_valid) {
        const int j = sorted_idx[i];
#pragma unroll
        for (int k = 0; k < DIM; k++) lf->c[k][t] = cp.c[k][j];
        lf->idx[t] = j;
        r.x = cp.c[0][j]; r.y = cp.c[1][j]; r.z = cp.c[2][j]; r.idx = j;
        if (nx) { r.nx = nx[j]; r.ny = ny[j]; r.nz = nz[j]; }
        if (rgba) r.rgba = rgba[j];
    } else {
#pragma unroll
        for (int k = 0; k < DIM; k++) lf->c[k][t] = (k < 3) ? INFINITY : 0.f;
        lf->idx[t] = -1;
    }
    recs[i] = r;
}

// Boxes of the children of the internal nodes [first, first + count), bottom-up.  child_is_leaf: children are leaves.
template <int DIM>
__device__ __forceinline__ void child_box(const BvhLeafT<DIM>* __restrict__ leaves, const BvhNodeT<DIM>* __restrict__ nodes, int child, int Lp, int n_leaves,
                                          bool child_is_leaf, float* lo, float* hi) {
#pragma unroll
    for (int k = 0; k < DIM; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }          // empty box: lower bound = +inf
    if (child_is_leaf) {
        const int leaf = child - (Lp - 1);
        if (leaf < n_leaves) {
            const BvhLeafT<DIM>* lf = leaves + leaf;
            for (int t = 0; t < BVH_LEAF; t++) {
                if (lf->c[0][t] < INFINITY) {
#pragma unroll
                    for (int k = 0; k < DIM; k++) { lo[k] = fminf(lo[k], lf->c[k][t]); hi[k] = fmaxf(hi[k], lf->c[k][t]); }
                }
            }
        }
    } else {
        const BvhNodeT<DIM>* nd = nodes + child;
#pragma unroll
        for (int k = 0; k < DIM; k++) { lo[k] = fminf(nd->lo[k][0], nd->lo[k][1]); hi[k] = fmaxf(nd->hi[k][0], nd->hi[k][1]); }
    }
}
template <int DIM>
__global__ void k_bvh_nodes(const BvhLeafT<DIM>* __restrict__ leaves, int n_leaves, int Lp, int first, int count, int children_are_leaves, BvhNodeT<DIM>* __restrict__ nodes) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int node = first + t;
    float lo0[DIM], hi0[DIM], lo1[DIM], hi1[DIM];
    child_box<DIM>(leaves, nodes, 2 * node + 1, Lp, n_leaves, children_are_leaves != 0, lo0, hi0);
    child_box<DIM>(leaves, nodes, 2 * node + 2, Lp, n_leaves, children_are_leaves != 0, lo1, hi1);
    BvhNodeT<DIM>* out = nodes + node;
#pragma unroll
    for (int k = 0; k < DIM; k++) { out->lo[k][0] = lo0[k]; out->lo[k][1] = lo1[k]; out->hi[k][0] = hi0[k]; out->hi[k][1] = hi1[k]; }
}

// 4-wide nodes from the finished binary records.  Virtual binary depth v = real depth + pad (pad = 1 when the real depth of
// the leaves is odd: a virtual root whose second half is empty); 4-wide node (l, idx) is virtual node (2l, idx) and stores the
// boxes of the virtual nodes (2l + 2, 4 idx + c), each of which is a child box of a real binary record one level up.
template <int DIM>
__global__ void k_bvh_quad_nodes(const BvhNodeT<DIM>* __restrict__ nodes, int pad, int Lq, BvhQuadT<DIM>* __restrict__ qnodes) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = (int)((((1ll << (2 * Lq)) - 1) / 3) * 4);
    if (t >= total) return;
    const int q = t >> 2, c = t & 3;
    int l = 0; while ((int)(((1ll << (2 * (l + 1))) - 1) / 3) <= q) l++;         // level of 4-wide node q
    const int idx = q - (int)(((1ll << (2 * l)) - 1) / 3);
    const int rd = 2 * l + 2 - pad, ri = 4 * idx + c;                          // real depth / index of child c
    float lo[DIM], hi[DIM];
#pragma unroll
    for (int k = 0; k < DIM; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }      // empty box: lower bound = +inf
    if (ri < (1 << rd)) {
        const BvhNodeT<DIM>* nd = nodes + ((1 << (rd - 1)) - 1 + (ri >> 1));
#pragma unroll
        for (int k = 0; k < DIM; k++) { lo[k] = nd->lo[k][ri & 1]; hi[k] = nd->hi[k][ri & 1]; }
    }
#pragma unroll
    for (int k = 0; k < DIM; k++) { qnodes[q].lo[k][c] = lo[k]; qnodes[q].hi[k][c] = hi[k]; }
}

// Lower bounds of the fp32 squared distance from the query to any point of the two child boxes, both at once (packed
// f32), accumulated in the SAME order as the point distance: ((e0^2 + e1^2) + e2^2) [+ e3^2 + e4^2 + e5^2].
template <int DIM>
__device__ __forceinline__ f2 pair_lb(const BvhNodeT<DIM>* __restrict__ nd, const f2* p2) {
    f2 acc;
#pragma unroll
    for (int k = 0; k < DIM; k++) {
        const f2 lo = *(const f2*)nd->lo[k], hi = *(const f2*)nd->hi[k];
        const f2 a = lo - p2[k], b = p2[k] - hi;
        const f2 e = {fmaxf(fmaxf(a.x, b.x), 0.f), fmaxf(fmaxf(a.y, b.y), 0.f)};
        const f2 sq = e * e;
        acc = (k == 0) ? sq : acc + sq;
    }
    return acc;
}

// Evaluate the 8 points of a leaf against the lane's query; exact lexicographic (d2, lowest index) update.
// best2 follows the smallest distance among all evaluated points OTHER than the current winner (see k_knn_bvh).
template <int DIM>
__device__ __forceinline__ void leaf_eval(const BvhLeafT<DIM>* __restrict__ lf, int leaf, const f2* p2, float& best, int& bi, int& bpos, float& best2) {
    float dd[BVH_LEAF];
    float m = FLT_MAX;
#pragma unroll
    for (int t = 0; t < BVH_LEAF; t += 2) {
        f2 d;
#pragma unroll
        for (int k = 0; k < DIM; k++) {
            const f2 q = *(const f2*)(&lf->c[k][t]);
            const f2 e = p2[k] - q;
            const f2 sq = e * e;
            d = (k == 0) ? sq : d + sq;
        }
        dd[t] = d.x; dd[t + 1] = d.y;
        m = fminf(fminf(m, d.x), d.y);
    }
    if (m <= best) {                     // something in this leaf ties or beats the running best (or IS the running best)
#pragma unroll
        for (int t = 0; t < BVH_LEAF; t++) {
            const int j = lf->idx[t];
            const bool take = (dd[t] < best) | ((dd[t] == best) & (j < bi));     // first minimum = lowest original index
            const float other = take ? best : ((j != bi) ? dd[t] : FLT_MAX);      // the dethroned winner, or a non-winning point
            best2 = fminf(best2, other);
            best = take ? dd[t] : best; bi = take ? j : bi; bpos = take ? leaf * BVH_LEAF + t : bpos;
        }
    } else best2 = fminf(best2, m);      // nobody here can win: all 8 are "others"
}

// Temporal seeding: ICP moves the queries a little per iteration, so the previous iteration's neighbour j0 is a
// good first candidate.  The traversal starts with (best, bi) = (d2(p, target[j0]), j0) -- a real candidate evaluated
// with the same fp32 formula -- and the final (d2, index) is still the exact lexicographic minimum over ALL targets
// (a box is skipped only if its lower bound exceeds the running best).
template <int DIM>
__device__ __forceinline__ void seed_from_previous(const int* __restrict__ nn_pos, int use_prev, const BvhViewT<DIM>& bv, int k, const float* p, float& best, int& bi, int& bpos) {
    if (!use_prev) return;
    const int q0 = nn_pos[k];                             // POSITION (8 * leaf + slot) of the previous neighbour
    if (q0 < 0) return;
    float t[DIM]; int j0;
    if (DIM == 3) { const float4 r = *(const float4*)(bv.recs + q0); t[0] = r.x; t[1] = r.y; t[2] = r.z; j0 = __float_as_int(r.w); }
    else {
        const BvhLeafT<DIM>* lf = bv.leaves + (q0 >> 3);
#pragma unroll
        for (int q = 0; q < DIM; q++) t[q] = lf->c[q][q0 & 7];
        j0 = lf->idx[q0 & 7];
    }
    float d = 0.f;
#pragma unroll
    for (int q = 0; q < DIM; q++) { const float e = p[q] - t[q]; d = (q == 0) ? e * e : d + e * e; }
    if (d < best) { best = d; bi = j0; bpos = q0; }
}

// Per-lane traversal state of the complete binary tree in heap order: three registers -- depth, index within the level
// and a bit mask of the levels whose far sibling is still pending.  The only per-level storage is the far sibling's
// lower bound, kept as a 16-bit truncated (never larger, hence conservative) value in LDS: 2 B x depth per lane, which
// leaves room for the full 32 waves per CU.
struct TravState { int depth; int idx; unsigned int pending; bool alive; };

__device__ __forceinline__ void trav_pop(TravState& st, const unsigned short* __restrict__ lb16, int tid, int nthreads, float best, float& minlb) {
    while (!st.alive && st.pending) {                     // deepest pending sibling that survives the (possibly improved) bound
        const int d = 31 - __clz((int)st.pending);
        st.pending &= ~(1u << d);
        const float lb = __uint_as_float((unsigned int)lb16[d * nthreads + tid] << 16);      // <= true bound
        if (!(lb * 0.99999f > best)) { st.idx = (st.idx >> (st.depth - d - 1)) ^ 1; st.depth = d + 1; st.alive = true; }
        else minlb = fminf(minlb, lb);                    // skipped subtree: everything in it is at least this far
    }
}

// ---- 4-wide walk -------------------------------------------------------------------------------------------------------
// Same exactness argument as the binary walk (a box is skipped only if its lower bound, computed with the operation order
// of the point distance, exceeds the running best), half the depth.  Per-lane state: level, index within the level and
// 4 pending-child bits per level in one 64-bit mask; the pending children's bounds live in LDS as 4 x 16-bit truncated
// floats per level and lane (one 8-byte access).  Siblings are visited in ascending order of their bound.
struct QuadState { int L; int idx; unsigned long long pending; bool alive; };

template <int DIM>
__device__ __forceinline__ void quad_lb(const BvhQuadT<DIM>* __restrict__ nd, const f2* p2, f2& l01, f2& l23) {
#pragma unroll
    for (int k = 0; k < DIM; k++) {
        const f2 lo0 = *(const f2*)&nd->lo[k][0], lo1 = *(const f2*)&nd->lo[k][2], hi0 = *(const f2*)&nd->hi[k][0], hi1 = *(const f2*)&nd->hi[k][2];
        const f2 a0 = lo0 - p2[k], b0 = p2[k] - hi0, a1 = lo1 - p2[k], b1 = p2[k] - hi1;
        const f2 e0 = {fmaxf(fmaxf(a0.x, b0.x), 0.f), fmaxf(fmaxf(a0.y, b0.y), 0.f)};
        const f2 e1 = {fmaxf(fmaxf(a1.x, b1.x), 0.f), fmaxf(fmaxf(a1.y, b1.y), 0.f)};
        const f2 s0 = e0 * e0, s1 = e1 * e1;
        l01 = (k == 0) ? s0 : l01 + s0;
        l23 = (k == 0) ? s1 : l23 + s1;
    }
}

__device__ __forceinline__ void quad_pop(QuadState& st, const uint2* __restrict__ lbq, int tid, int nthreads, float best, float& minlb) {
    while (!st.alive && st.pending) {
        const int lv = (63 - __clzll((long long)st.pending)) >> 2;                // deepest level with pending children
        const unsigned int bits = (unsigned int)(st.pending >> (4 * lv)) & 0xFu;
        const uint2 w = lbq[lv * nthreads + tid];
        const float l0 = (bits & 1u) ? __uint_as_float(w.x << 16) : FLT_MAX, l1 = (bits & 2u) ? __uint_as_float(w.x & 0xFFFF0000u) : FLT_MAX;
        const float l2 = (bits & 4u) ? __uint_as_float(w.y << 16) : FLT_MAX, l3 = (bits & 8u) ? __uint_as_float(w.y & 0xFFFF0000u) : FLT_MAX;
        const float m = fminf(fminf(l0, l1), fminf(l2, l3));                      // truncated bounds: <= the true ones
        if (m * 0.99999f > best) {                                               // the nearest pending sibling is out: so are the others
            minlb = fminf(minlb, m);
            st.pending &= ~(0xFull << (4 * lv));
        } else {
            const int c = (l0 == m) ? 0 : (l1 == m) ? 1 : (l2 == m) ? 2 : 3;
            st.pending &= ~(1ull << (4 * lv + c));
            st.idx = ((st.idx >> (2 * (st.L - lv))) << 2) | c; st.L = lv + 1; st.alive = true;
        }
    }
}

// The walk is a chain of dependent loads, each a trip to L2 or HBM.  A seeded query already knows where it will most
// likely end up: in or next to the leaf of its previous neighbour, whose ancestors are known arithmetically in the implicit
// layout.  Touching that whole root-to-leaf path up front turns the chain of misses into ONE round of parallel misses followed
// by cache hits.  (The lowest 8 levels; anything above is shared by everybody and hot.)
template <int DIM>
__device__ __forceinline__ unsigned int quad_prefetch_path(const BvhViewT<DIM>& bv, int leaf) {
    // plain loads whose values are only consumed (by an empty asm) AFTER the walk: nothing waits for them specially, they
    // simply travel together with the walk's first node load
    unsigned int sink = *(const unsigned int*)(bv.leaves + leaf);
    if (DIM == 6) sink |= *((const unsigned int*)(bv.leaves + leaf) + 32);
    unsigned int t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {                         // branch-free (levels above the root clamp to the root): the loads issue back to back
        const int L = max(bv.Lq - 1 - u, 0), sh = min(2 * (u + 1), 2 * bv.Lq);
        const unsigned int* nd = (const unsigned int*)(bv.qnodes + ((0x5555555555555555ull & ((1ull << (2 * L)) - 1ull)) + (unsigned long long)(leaf >> sh)));
        t[u] = nd[31];
        if (DIM == 6) t[u] |= nd[63];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) sink |= t[u];
    return sink;
}

template <int DIM>
__device__ __forceinline__ void quad_run(const BvhViewT<DIM>& bv, const f2* p2, QuadState& st,
                                         float& best, int& bi, int& bpos, float& best2, float& minlb, uint2* __restrict__ lbq, int tid, int nthreads) {
    const int Lq = bv.Lq;
    while (st.alive) {
        while (st.alive && st.L < Lq) {
            f2 l01, l23;
            quad_lb<DIM>(bv.qnodes + ((0x5555555555555555ull & ((1ull << (2 * st.L)) - 1ull)) + (unsigned long long)st.idx), p2, l01, l23);
            const float m = fminf(fminf(l01.x, l01.y), fminf(l23.x, l23.y));
            const bool s0 = !(l01.x * 0.99999f > best), s1 = !(l01.y * 0.99999f > best), s2 = !(l23.x * 0.99999f > best), s3 = !(l23.y * 0.99999f > best);
            minlb = fminf(minlb, fminf(fminf(s0 ? FLT_MAX : l01.x, s1 ? FLT_MAX : l01.y), fminf(s2 ? FLT_MAX : l23.x, s3 ? FLT_MAX : l23.y)));   // skipped right here
            if (!(m * 0.99999f > best)) {
                const int c = (l01.x == m) ? 0 : (l01.y == m) ? 1 : (l23.x == m) ? 2 : 3;
                const unsigned int pend = ((s0 ? 1u : 0u) | (s1 ? 2u : 0u) | (s2 ? 4u : 0u) | (s3 ? 8u : 0u)) & ~(1u << c);
                if (pend) {
                    uint2 w;
                    w.x = (__float_as_uint(l01.x) >> 16) | (__float_as_uint(l01.y) & 0xFFFF0000u);
                    w.y = (__float_as_uint(l23.x) >> 16) | (__float_as_uint(l23.y) & 0xFFFF0000u);
                    lbq[st.L * nthreads + tid] = w;
                    st.pending |= (unsigned long long)pend << (4 * st.L);
                }
                st.idx = (st.idx << 2) | c; st.L++;
            } else st.alive = false;                      // all four children pruned
            quad_pop(st, lbq, tid, nthreads, best, minlb);
        }
        if (st.alive) {
            leaf_eval<DIM>(bv.leaves + st.idx, st.idx, p2, best, bi, bpos, best2);
            st.alive = false;
            quad_pop(st, lbq, tid, nthreads, best, minlb);
        }
    }
}

// XCD-aware block mapping: workgroups are dealt round-robin over the 8 XCDs (block b runs on the XCD group b % 8), each
// with a private 4 MiB L2.  With Morton-sorted queries, giving every XCD group ONE contiguous slice of the sorted list
// means its L2 only has to hold the part of the tree under that slice (plus the shared top levels) instead of all of it.
__device__ __forceinline__ int xcd_contiguous_block(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, j = b >> 3;        // XCD group x owns q (+1 if x < r) consecutive logical blocks
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// Every lane walks the tree on its own for its own query.  Measured on MI355X (370k x 370k, DIM 3): ~26 node records and
// ~3 leaves per query, ~50 % of the wave time waiting on dependent loads, ~33 % of the lanes active on average (traversal
// lengths differ per lane).  What moved it: Morton-sorted queries + XCD-contiguous slices (L2 hit 59 % -> 89 %), the
// 2-byte-per-level stack (occupancy), temporal seeding, and -- once ICP has converged -- the verify-and-skip test below,
// which retires whole waves without a traversal.  Tried and rejected (slower, see git history): wave-packet
// traversal with scalar node loads (the union of 64 lanes' subtrees is 3x larger), persistent lanes with wave-level
// refill (fewer waves in flight), a second cooperative pass for queries over a step budget.
template <int DIM>
__device__ __forceinline__ void knn_load_query(const KnnParams& kp, int k, float* p) {
    const int i = kp.sel ? kp.sel[k] : k;
    p[0] = kp.sx[i]; p[1] = kp.sy[i]; p[2] = kp.sz[i];
    if (!kp.pretransformed) { float a, b, c; xform_point(kp.ps->pose, p[0], p[1], p[2], a, b, c); p[0] = a; p[1] = b; p[2] = c; }
    if (DIM == 6) { p[3 % DIM] = kp.scr[i]; p[4 % DIM] = kp.scg[i]; p[5 % DIM] = kp.scb[i]; }
}

// Incremental search.  The last full search left, for this query, a lower bound L on the distance to every target
// other than its neighbour j0.  The query has since moved by delta, so every other target is still at least
// L - delta away (triangle inequality); if the re-evaluated distance to j0 is strictly below that, j0 is still THE
// unique fp32 argmin and the traversal is skipped.  All margins (1e-6 relative) dominate the fp32 rounding of the
// distance formula (< 4e-7), so the result is bit-identical to a full search; otherwise a full search runs.
// Seeds (best, bi) with the previous neighbour either way.
template <int DIM>
__device__ __forceinline__ bool knn_try_verify(const KnnParams& kp, const BvhViewT<DIM>& bv, int k, const float* p, float& best, int& bi, int& bpos, float& lb_others) {
    seed_from_previous<DIM>(kp.nn_raw, kp.use_prev, bv, k, p, best, bi, bpos);
    if (kp.incremental && kp.use_prev && bi >= 0) {
        const float4 s = kp.qstate[k];
        const float ex = p[0] - s.x, ey = p[1] - s.y, ez = p[2] - s.z;
        const float delta = sqrtf((ex * ex + ey * ey) + ez * ez) * 1.000001f + 1e-30f;
        const float lbn = (s.w - delta) * 0.999999f;
        if (sqrtf(best) * 1.000001f < lbn) { lb_others = lbn; return true; }
    }
    return false;
}

template <int DIM>
__device__ __forceinline__ void knn_store_state(const KnnParams& kp, int k, const float* p, float best, int bpos, float lb_others) {
    if (kp.qstate) { float4 s; s.x = p[0]; s.y = p[1]; s.z = p[2]; s.w = lb_others; kp.qstate[k] = s; }
    if (kp.nn_raw) kp.nn_raw[k] = bpos;
    if (kp.d2_out) kp.d2_out[k] = best;
}

template <int DIM>
__device__ __forceinline__ void knn_bvh_query(const KnnParams& kp, const BvhViewT<DIM>& bv, int k, uint2* __restrict__ lbq, int tid,
                                              float& best, int& bi, int& bpos) {
    float p[DIM];
    knn_load_query<DIM>(kp, k, p);
    best = FLT_MAX; bi = -1; bpos = -1;
    float lb_others = 0.f;               // lower bound on the (real) distance from p to every target except bi
    if (finite3(p[0], p[1], p[2]) && bv.n_valid > 0) {
        if (!knn_try_verify<DIM>(kp, bv, k, p, best, bi, bpos, lb_others)) {
            f2 p2[DIM];
#pragma unroll
            for (int q = 0; q < DIM; q++) { p2[q].x = p[q]; p2[q].y = p[q]; }
            float best2 = FLT_MAX, minlb = FLT_MAX;
            unsigned int touched = 0u;
            if (ICP_PREFETCH_PATH && bpos >= 0) touched = quad_prefetch_path<DIM>(bv, bpos >> 3);
            QuadState st; st.L = 0; st.idx = 0; st.pending = 0ull; st.alive = true;
            quad_run<DIM>(bv, p2, st, best, bi, bpos, best2, minlb, lbq, tid, BVH_THREADS);
            asm volatile("" ::"v"(touched));
            lb_others = sqrtf(fminf(best2, minlb)) * 0.999999f;
        }
    }
    knn_store_state<DIM>(kp, k, p, best, bpos, lb_others);
}

// Which query does this lane serve?  Either position t of the (Morton-sorted) query order, or -- second pass of the
// incremental search -- entry t of the work list, whose length is only known on the device: the launch covers the worst case
// and blocks past the end retire at once; the XCD-contiguous slices are cut over the blocks actually in use.
__device__ __forceinline__ int knn_bvh_lane_query(const KnnParams& kp, const int* __restrict__ qorder, int tid) {
    if (kp.work_items) {
        const int n = *kp.work_n, nb = (n + BVH_THREADS - 1) / BVH_THREADS;
        if ((int)blockIdx.x >= nb) return -1;
        const int t = xcd_contiguous_block(blockIdx.x, nb) * BVH_THREADS + tid;
        return t < n ? kp.work_items[t] : -1;
    }
    const int t = xcd_contiguous_block(blockIdx.x, gridDim.x) * BVH_THREADS + tid;
    if (t >= kp.n) return -1;
    return qorder ? qorder[t] : t;                        // spatially sorted queries: neighbouring lanes walk similar paths
}

template <int DIM>
__global__ __launch_bounds__(BVH_THREADS) void k_knn_bvh(const KnnParams kp, const BvhViewT<DIM> bv, const int* __restrict__ qorder) {
    extern __shared__ uint2 bvh_lbq[];                    // [Lq][BVH_THREADS] pending-sibling bounds
    const int tid = threadIdx.x;
    const int k = knn_bvh_lane_query(kp, qorder, tid);
    if (k < 0) return;
    float best; int bi, bpos;
    knn_bvh_query<DIM>(kp, bv, k, bvh_lbq, tid, best, bi, bpos);
    icp_match_t m;
    if (best <= kp.max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }
    kp.out[k] = m;
}

// First pass of the incremental search: a streaming kernel that re-evaluates every query against its previous neighbour.
// Verified queries are finished here; the others are appended to the work list for the tree walk (one wave-aggregated
// atomic per wave; the list order varies from run to run, the per-query results do not depend on it).  Packing the
// survivors densely matters: left in place they would keep almost every wave walking the tree at a few lanes' utilisation.
constexpr int VERIFY_THREADS = 256;
template <int DIM>
__global__ __launch_bounds__(VERIFY_THREADS) void k_knn_verify(const KnnParams kp, const BvhViewT<DIM> bv, const int* __restrict__ qorder) {
    const int t = xcd_contiguous_block(blockIdx.x, gridDim.x) * VERIFY_THREADS + threadIdx.x;
    bool push = false; int k = -1;
    if (t < kp.n) {
        k = qorder ? qorder[t] : t;
        float p[DIM];
        knn_load_query<DIM>(kp, k, p);
        float best = FLT_MAX, lb_others = 0.f; int bi = -1, bpos = -1;
        if (finite3(p[0], p[1], p[2]) && bv.n_valid > 0) {
            push = !knn_try_verify<DIM>(kp, bv, k, p, best, bi, bpos, lb_others);
        }
        if (!push) {
            knn_store_state<DIM>(kp, k, p, best, bpos, lb_others);
            icp_match_t m;
            if (best <= kp.max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }
            kp.out[k] = m;
        }
    }
    const unsigned long long mask = __ballot(push);
    if (mask) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(kp.work_n, __popcll(mask));
        base = __shfl(base, 0, WAVE);
        if (push) kp.work_items[base + __popcll(mask & ((1ull << lane) - 1ull))] = k;
    }
}

// ------------------------------------------------------------------------------------------------
// Surface normals from the K nearest neighbours (the PCL NormalEstimation the reference runs on the ETH scans,
// PointCloud.h:41-76: setKSearch(5), viewpoint (0,0,0)): K-NN over the cloud's own kd-ordered BVH (the point itself is its
// first neighbour, as with pcl::search::KdTree), fp64 covariance of the K points, eigenvector of the smallest eigenvalue
// (fp64 Jacobi), flipped towards the viewpoint (pcl::flipNormalTowardsViewpoint), curvature = l0 / (l0 + l1 + l2).
// Neighbour sets are the exact K smallest (d2, index) pairs.  PCL itself is absent here: parity unpinned, checked against numpy.
template <int n> __device__ inline void jacobi_eig_sym(double* A, double* V, double* ev);     // defined with the solvers below

template <int K>
__device__ __forceinline__ void knn_insert(float (&bd)[K], int (&bj)[K], float d, int j) {
    // keep (bd, bj) sorted ascending by (d, j); called only when (d, j) beats the current worst
    bd[K - 1] = d; bj[K - 1] = j;
#pragma unroll
    for (int q = K - 1; q > 0; q--) {
        const bool sw = (bd[q] < bd[q - 1]) | ((bd[q] == bd[q - 1]) & (bj[q] < bj[q - 1]));
        const float td = bd[q]; const int tj = bj[q];
        bd[q] = sw ? bd[q - 1] : bd[q]; bj[q] = sw ? bj[q - 1] : bj[q];
        bd[q - 1] = sw ? td : bd[q - 1]; bj[q - 1] = sw ? tj : bj[q - 1];
    }
}

template <int K>
__global__ __launch_bounds__(BVH_THREADS) void k_normals_knn(const BvhViewT<3> bv, int n, int tree_depth, float vpx, float vpy, float vpz,
                                                             float* __restrict__ nrm_out /* AoS n x 3 */, float* __restrict__ curv_out) {
    extern __shared__ unsigned short bvh_lb16[];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * BVH_THREADS + tid;
    if (i >= n) return;
    const float px = bv.tgt.c[0][i], py = bv.tgt.c[1][i], pz = bv.tgt.c[2][i];
    float nx = NAN, ny = NAN, nz = NAN, curv = NAN;
    if (finite3(px, py, pz) && bv.n_valid >= 3) {
        float bd[K]; int bj[K];
#pragma unroll
        for (int q = 0; q < K; q++) { bd[q] = FLT_MAX; bj[q] = 0x7fffffff; }
        f2 p2[3] = {{px, px}, {py, py}, {pz, pz}};
        TravState st; st.depth = 0; st.idx = 0; st.pending = 0u; st.alive = true;
        float unused_minlb = FLT_MAX;
        while (st.alive) {
            while (st.alive && st.depth < tree_depth) {
                const f2 l = pair_lb<3>(bv.nodes + ((1 << st.depth) - 1 + st.idx), p2);
                const bool swap = l.y < l.x;
                const float ln = swap ? l.y : l.x, lf = swap ? l.x : l.y;
                const float worst = bd[K - 1];
                const bool take_near = !(ln * 0.99999f > worst), take_far = !(lf * 0.99999f > worst);
                if (take_near) {
                    if (take_far) { bvh_lb16[st.depth * BVH_THREADS + tid] = (unsigned short)(__float_as_uint(lf) >> 16); st.pending |= 1u << st.depth; }
                    st.idx = 2 * st.idx + (swap ? 1 : 0); st.depth++;
                } else st.alive = false;
                trav_pop(st, bvh_lb16, tid, BVH_THREADS, bd[K - 1], unused_minlb);
            }
            if (st.alive) {
                const BvhLeafT<3>* __restrict__ lf = bv.leaves + st.idx;
#pragma unroll
                for (int t = 0; t < BVH_LEAF; t++) {
                    const float dx = px - lf->c[0][t], dy = py - lf->c[1][t], dz = pz - lf->c[2][t];
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    const int j = lf->idx[t];
                    if (j >= 0 && ((d < bd[K - 1]) | ((d == bd[K - 1]) & (j < bj[K - 1])))) knn_insert<K>(bd, bj, d, j);
                }
                st.alive = false;
                trav_pop(st, bvh_lb16, tid, BVH_THREADS, bd[K - 1], unused_minlb);
            }
        }
        int cnt = 0;
        double m[3] = {0, 0, 0}, cxx = 0, cxy = 0, cxz = 0, cyy = 0, cyz = 0, czz = 0;
#pragma unroll
        for (int q = 0; q < K; q++) if (bd[q] < FLT_MAX) { const int j = bj[q]; m[0] += bv.tgt.c[0][j]; m[1] += bv.tgt.c[1][j]; m[2] += bv.tgt.c[2][j]; cnt++; }
        if (cnt >= 3) {
            m[0] /= cnt; m[1] /= cnt; m[2] /= cnt;
#pragma unroll
            for (int q = 0; q < K; q++) if (bd[q] < FLT_MAX) {
                const int j = bj[q];
                const double a = bv.tgt.c[0][j] - m[0], b = bv.tgt.c[1][j] - m[1], c = bv.tgt.c[2][j] - m[2];
                cxx += a * a; cxy += a * b; cxz += a * c; cyy += b * b; cyz += b * c; czz += c * c;
            }
            double A[9] = {cxx / cnt, cxy / cnt, cxz / cnt, cxy / cnt, cyy / cnt, cyz / cnt, cxz / cnt, cyz / cnt, czz / cnt}, V[9], ev[3];
            jacobi_eig_sym<3>(A, V, ev);
            int s0 = 0; if (ev[1] < ev[s0]) s0 = 1; if (ev[2] < ev[s0]) s0 = 2;
            double vx = V[0 * 3 + s0], vy = V[1 * 3 + s0], vz = V[2 * 3 + s0];
            const double len = sqrt(vx * vx + vy * vy + vz * vz);
            vx /= len; vy /= len; vz /= len;
            if ((vpx - px) * vx + (vpy - py) * vy + (vpz - pz) * vz < 0) { vx = -vx; vy = -vy; vz = -vz; }   // flipNormalTowardsViewpoint
            nx = (float)vx; ny = (float)vy; nz = (float)vz;
            const double tr = ev[0] + ev[1] + ev[2];
            curv = tr > 0 ? (float)(fabs(ev[s0]) / tr) : 0.f;
        }
    }
    nrm_out[(size_t)i * 3] = nx; nrm_out[(size_t)i * 3 + 1] = ny; nrm_out[(size_t)i * 3 + 2] = nz;
    if (curv_out) curv_out[i] = curv;
}

// out[t] = in[idx[t]] (one plane of a cloud) / out[t] = sel[order[t]]: the one-off physical permutation of the source into Morton order
__global__ void k_gather_f32(const float* __restrict__ in, const int* __restrict__ idx, int n, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) out[t] = in[idx[t]];
}
__global__ void k_gather_u32(const uint32_t* __restrict__ in, const int* __restrict__ idx, int n, uint32_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) out[t] = in[idx[t]];
}
__global__ void k_compose_idx(const int* __restrict__ sel, const int* __restrict__ order, int n, int* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) out[t] = sel ? sel[order[t]] : order[t];
}
__global__ void k_iota(int* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i; }

// Morton key of the (untransformed) query points -> spatially coherent waves for k_knn_bvh.
__global__ void k_query_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const int* __restrict__ sel, int n,
                             float lox, float loy, float loz, float sx, float sy, float sz, unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i = sel ? sel[t] : t;
    const float a = x[i], b = y[i], c = z[i];
    unsigned long long key = ~0ull;
    if (finite3(a, b, c)) {
        const float fa = fminf(fmaxf((a - lox) * sx, 0.f), 2097151.f), fb = fminf(fmaxf((b - loy) * sy, 0.f), 2097151.f), fc = fminf(fmaxf((c - loz) * sz, 0.f), 2097151.f);
        key = spread21((unsigned int)fa) | (spread21((unsigned int)fb) << 1) | (spread21((unsigned int)fc) << 2);
    }
    keys[t] = key; vals[t] = t;
}

// ------------------------------------------------------------------------------------------------
// Projective matcher, NearestNeighbor.h:333-421.  One lane = one query; the 25x25 window of the
// organised target is read through L1/L2 (neighbouring lanes' windows overlap almost entirely).
struct ProjParams {
    const float* sx; const float* sy; const float* sz; const int* sel; int n;
    const float* tx; const float* ty; const float* tz; int width; int height;
    float fx, fy, mx, my; int window;
    const PoseState* ps; int pretransformed; float max_dist;
    icp_match_t* out; float* d2_out;
};

__global__ __launch_bounds__(256) void k_projective(const ProjParams pp) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= pp.n) return;
    const int i = pp.sel ? pp.sel[k] : k;
    float px = pp.sx[i], py = pp.sy[i], pz = pp.sz[i];
    if (!pp.pretransformed) { float a, b, c; xform_point(pp.ps->pose, px, py, pz, a, b, c); px = a; py = b; pz = c; }
    icp_match_t m; float best = FLT_MAX;
    if (px == -INFINITY) {                                   // :372-373 leaves the value-initialised Match{0, 0.f}
        m.idx = 0; m.weight = 0.f;
    } else {
        const float uf = roundf(((px * pp.fx) / pz) + pp.mx);    // :378
        const float vf = roundf(((py * pp.fy) / pz) + pp.my);    // :379
        const float wf = (float)pp.window;
        int bi = -1;
        // unsigned underflow (:385-386): a window starting below 0 never runs; NaN / negative / huge => no match
        if (uf >= wf && vf >= wf && uf < 2147483648.f && vf < 2147483648.f) {
            const long long u0 = (long long)uf - pp.window, u1 = (long long)uf + pp.window;
            const long long v0 = (long long)vf - pp.window, v1 = (long long)vf + pp.window;
            const int ve = (int)(v1 < (long long)pp.height - 1 ? v1 : (long long)pp.height - 1);
            const int ue = (int)(u1 < (long long)pp.width - 1 ? u1 : (long long)pp.width - 1);
            if (v0 < pp.height && u0 < pp.width) {
                for (int v = (int)v0; v <= ve; v++) {
                    const int row = v * pp.width;
                    for (int u = (int)u0; u <= ue; u++) {
                        const int j = row + u;
                        const float qx = pp.tx[j];
                        if (qx == -INFINITY) continue;           // :392
                        const float dx = px - qx, dy = py - pp.ty[j], dz = pz - pp.tz[j];
                        const float d = dx * dx + (dy * dy + dz * dz);     // :396 Eigen squaredNorm tree
                        if (d < best) { best = d; bi = j; }      // :399
                    }
                }
            }
        }
        if (best <= pp.max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }   // :407-415
    }
    pp.out[k] = m;
    if (pp.d2_out) pp.d2_out[k] = best;
}

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
__device__ __forceinline__ bool post_eval(const PostParams& pp, int k, icp_match_t m, float d0, float d1, float d2, float nt0, float nt1, float nt2, uint32_t tcol,
                                          float& s0, float& s1, float& s2, float& w) {
    const float* __restrict__ P = pp.ps->pose;
    const float* __restrict__ N = pp.ps->nmat;
    const int i = pp.sel ? pp.sel[k] : k;
    float ns0, ns1, ns2;
    xform_point(P, pp.sx[i], pp.sy[i], pp.sz[i], s0, s1, s2);
    xform_normal(N, pp.snx[i], pp.sny[i], pp.snz[i], ns0, ns1, ns2);
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
__device__ __forceinline__ void fold_store(double x, double* lds, int a, int lane, int w) {
    x += __shfl_down(x, 32, WAVE);
    x += __shfl_down(x, 16, WAVE);
    if (lane < 16) lds[(w * 34 + a) * 17 + lane] = x;
}
template <int A, int END>
__device__ __forceinline__ void fold_row_slots(const RowTerms& R, bool valid, double* lds, int lane, int w) {
    if constexpr (A < END) {
        fold_store(valid ? row_slot<A>(R) : 0.0, lds, SUM_M + A, lane, w);
        fold_row_slots<A + 1, END>(R, valid, lds, lane, w);
    }
}
template <int DIM>
__global__ __launch_bounds__(BVH_THREADS) void k_knn_bvh_post(const KnnParams kp, const BvhViewT<DIM> bv, const int* __restrict__ qorder, const PostParams pp) {
    extern __shared__ uint2 bvh_lbq[];                    // [Lq][BVH_THREADS] pending-sibling bounds; reused by the reduction
    constexpr int NW = BVH_THREADS / WAVE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int k = knn_bvh_lane_query(kp, qorder, tid);
    bool valid = false;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, wt = 0.f;
    if (k >= 0) {
        float best; int bi, bpos;
        knn_bvh_query<DIM>(kp, bv, k, bvh_lbq, tid, best, bi, bpos);
        icp_match_t m;
        if (best <= kp.max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }
        if (m.idx < 0) pp.matches[k] = m;
        else {
            const float4 ra = *(const float4*)(bv.recs + bpos), rb = *((const float4*)(bv.recs + bpos) + 1);      // one 32-byte record
            d0 = ra.x; d1 = ra.y; d2 = ra.z; n0 = rb.x; n1 = rb.y; n2 = rb.z;
            valid = post_eval(pp, k, m, d0, d1, d2, n0, n1, n2, __float_as_uint(rb.w), s0, s1, s2, wt);
        }
    }
    __syncthreads();                                      // the traversal stacks are dead: reuse LDS for the reduction
    double* lds = (double*)bvh_lbq;
    fold_store(valid ? 1.0 : 0.0, lds, SUM_N, lane, w);
    fold_store(valid ? (double)s0 : 0.0, lds, SUM_S, lane, w); fold_store(valid ? (double)s1 : 0.0, lds, SUM_S + 1, lane, w); fold_store(valid ? (double)s2 : 0.0, lds, SUM_S + 2, lane, w);
    fold_store(valid ? (double)d0 : 0.0, lds, SUM_D, lane, w); fold_store(valid ? (double)d1 : 0.0, lds, SUM_D + 1, lane, w); fold_store(valid ? (double)d2 : 0.0, lds, SUM_D + 2, lane, w);
    if (pp.metric == ICP_METRIC_POINT_TO_PLANE) {
        RowTerms R;
        build_rows(0, s0, s1, s2, d0, d1, d2, n0, n1, n2, wt, R);
        fold_row_slots<0, 7>(R, valid, lds, lane, w);   __builtin_amdgcn_sched_barrier(0);
        fold_row_slots<7, 14>(R, valid, lds, lane, w);  __builtin_amdgcn_sched_barrier(0);
        fold_row_slots<14, 21>(R, valid, lds, lane, w); __builtin_amdgcn_sched_barrier(0);
        fold_row_slots<21, 27>(R, valid, lds, lane, w);
    } else {                                              // point-to-point moments (see post_core)
        const double wd = (double)wt;
        const double ws[3] = {wd * s0, wd * s1, wd * s2};
        const float dd[3] = {d0, d1, d2};
        fold_store(valid ? wd : 0.0, lds, SUM_M, lane, w);
#pragma unroll
        for (int q = 0; q < 3; q++) fold_store(valid ? ws[q] : 0.0, lds, SUM_M + 1 + q, lane, w);
#pragma unroll
        for (int q = 0; q < 3; q++) fold_store(valid ? wd * dd[q] : 0.0, lds, SUM_M + 4 + q, lane, w);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; j++) {
#pragma unroll
            for (int q = 0; q < 3; q++) fold_store(valid ? (double)dd[j] * ws[q] : 0.0, lds, SUM_M + 7 + j * 3 + q, lane, w);
        }
#pragma unroll
        for (int q = 16; q < 27; q++) fold_store(0.0, lds, SUM_M + q, lane, w);
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
        const int lb = kp.work_items ? (int)blockIdx.x : xcd_contiguous_block(blockIdx.x, gridDim.x);             // partial slot = logical block -> fixed summation order
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
    double A[36], V[36], ev[6];
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
__device__ inline void solve_fullpiv_lu6(double* M, double* rhs, double* x) {
    const int n = 6;
    int colp[6];
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
    double y[6] = {0, 0, 0, 0, 0, 0};
    for (int k = r - 1; k >= 0; k--) { double s = rhs[k]; for (int j = k + 1; j < r; j++) s -= M[k * n + j] * y[j]; y[k] = s / M[k * n + k]; }
    for (int k = 0; k < n; k++) x[colp[k]] = (k < r) ? y[k] : 0.0;
}

// Rotation of the weighted Procrustes problem: R = U diag(1,1,det(UV^T)) V^T of A = U S V^T
// (ProcrustesAligner.h:55-64).  V from the fp64 eigen-decomposition of A^T A (descending), U_c = A v_c/|A v_c|
// for the two leading columns.  With c = U_0 x U_1 the reference's product collapses to
//   R = U_0 V_0^T + U_1 V_1^T + det(V) * c * V_2^T
// (flipping the sign of the third left vector flips det(UV^T) too), which stays well defined when sigma_3 -> 0.
__device__ inline void procrustes_rotation(const double* A /* 3x3 row-major */, double* R) {
    double B[9], V[9], ev[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[k * 3 + i] * A[k * 3 + j]; B[i * 3 + j] = s; }
    jacobi_eig_sym<3>(B, V, ev);
    int o[3] = {0, 1, 2};
    for (int a = 0; a < 2; a++) for (int b = a + 1; b < 3; b++) if (ev[o[b]] > ev[o[a]]) { const int t = o[a]; o[a] = o[b]; o[b] = t; }
    double Vs[9], U[9];
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
};

// Grid of NSUM blocks: block a folds the partials of sum a in a fixed order (lanes stride the producer blocks, shuffle tree,
// then the waves in order) -- identical on every run and independent of block scheduling.  The block that finishes last (ticket
// counter, release/acquire fences at agent scope) gathers the NSUM totals and runs the small fp64 solve + pose composition.
constexpr int SOLVE_THREADS = 256;
__global__ __launch_bounds__(SOLVE_THREADS) void k_reduce_solve(const SolveParams sp) {
    __shared__ double tot[NSUM];
    __shared__ double wsum[SOLVE_THREADS / WAVE];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, a = blockIdx.x;
    {
        const double* __restrict__ row = sp.partials + (size_t)a * sp.nblocks;
        double x = 0.0;
        for (int b = threadIdx.x; b < sp.nblocks; b += SOLVE_THREADS) x += row[b];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
        if (lane == 0) wsum[w] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double x = wsum[0];
        for (int k = 1; k < SOLVE_THREADS / WAVE; k++) x += wsum[k];
        __hip_atomic_store(sp.totals + a, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned t = atomicAdd(sp.ticket, 1u);
        is_last = (t == (unsigned)(NSUM - 1));
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (threadIdx.x < NSUM) tot[threadIdx.x] = __hip_atomic_load(sp.totals + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x != 0) return;
    *sp.ticket = 0u;                                      // ready for the next launch on this stream
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
        double A[9];
        for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++)
            A[j * 3 + k] = m[7 + j * 3 + k] - m[4 + j] * (double)msf[k] - (double)mdf[j] * m[1 + k] + m[0] * (double)mdf[j] * (double)msf[k];
        double Rd[9]; float R[9];
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
    } else {
        // symmetric: M = A^T A + lambda^2 I, FullPivLU (ICPOptimizer.h:858-868)
        double M[36], g[6], x[6];
        int q = 0;
        for (int a = 0; a < 6; a++) for (int c = a; c < 6; c++) { M[a * 6 + c] = tot[SUM_M + q]; M[c * 6 + a] = tot[SUM_M + q]; q++; }
        for (int a = 0; a < 6; a++) g[a] = tot[SUM_M + 21 + a];
        const float lambda = 0.0001f; const float l2 = lambda * lambda;
        for (int a = 0; a < 6; a++) M[a * 6 + a] += (double)l2;
        solve_fullpiv_lu6(M, g, x);
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

// ------------------------------------------------------------------------------------------------
// ConvergenceMeasure::rmseAlignmentError (ConvergenceMeasure.h:50-66): sum of squared distances between
// pose*src[i] and ref[i] over pairs where both are finite.  fp64 block partials {sum, count}.
__global__ __launch_bounds__(256) void k_rmse_partial(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                                      const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz,
                                                      int n, const PoseState* __restrict__ ps, double* __restrict__ partials) {
    __shared__ double lds[4 * 2];
    double acc[2] = {0.0, 0.0};
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        float a, b, c;
        xform_point(ps->pose, sx[k], sy[k], sz[k], a, b, c);
        const float r0 = rx[k], r1 = ry[k], r2 = rz[k];
        if (finite3(a, b, c) && finite3(r0, r1, r2)) {
            const float e0 = a - r0, e1 = b - r1, e2 = c - r2;
            acc[0] += (double)(e0 * e0 + (e1 * e1 + e2 * e2));
            acc[1] += 1.0;
        }
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) { partials[blockIdx.x * 2] = acc[0]; partials[blockIdx.x * 2 + 1] = acc[1]; }
}
__global__ void k_rmse_finish(const double* __restrict__ partials, int nblocks, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0, c = 0.0;
    for (int b = 0; b < nblocks; b++) { s += partials[b * 2]; c += partials[b * 2 + 1]; }
    *out = (float)sqrt(s / c);
}

// ConvergenceMeasure::benchmarkError / calculate_error (ConvergenceMeasure.h:104-151), the Fontana-style metric of the ETH
// runs:  mean_i ( |T s_i - r_i| / |T s_i - centroid(T s)| ).  Pass 1: fp64 sums of the transformed points (PCL's
// compute3DCentroid accumulates in double), pass 2: fp32 distances as pcl::euclideanDistance computes them, fp64 sum.
__global__ __launch_bounds__(256) void k_fontana_centroid(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                                          int n, const PoseState* __restrict__ ps, double* __restrict__ partials /* [blocks][4] */) {
    __shared__ double lds[4 * 3];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        float a, b, c; xform_point(ps->pose, sx[k], sy[k], sz[k], a, b, c);
        acc[0] += (double)a; acc[1] += (double)b; acc[2] += (double)c;
    }
    block_reduce<3>(acc, lds);
    if (threadIdx.x == 0) { partials[blockIdx.x * 4] = acc[0]; partials[blockIdx.x * 4 + 1] = acc[1]; partials[blockIdx.x * 4 + 2] = acc[2]; }
}
__global__ __launch_bounds__(256) void k_fontana_error(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                                       const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz,
                                                       int n, const PoseState* __restrict__ ps, const double* __restrict__ cpart, int cblocks,
                                                       double* __restrict__ partials /* [blocks] */) {
    __shared__ double lds[4];
    __shared__ float cen[3];
    if (threadIdx.x < 3) {        // every block folds the centroid partials in the same fixed order
        double s = 0.0; for (int b = 0; b < cblocks; b++) s += cpart[b * 4 + threadIdx.x];
        cen[threadIdx.x] = (float)(s / (double)n);                    // pcl::PointXYZ centroid(centroid_v[0], ...) :114
    }
    __syncthreads();
    const float c0 = cen[0], c1 = cen[1], c2 = cen[2];
    double acc[1] = {0.0};
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        float a, b, c; xform_point(ps->pose, sx[k], sy[k], sz[k], a, b, c);
        const float e0 = a - rx[k], e1 = b - ry[k], e2 = c - rz[k];
        const float g0 = a - c0, g1 = b - c1, g2 = c - c2;
        const float dist = sqrtf(e0 * e0 + (e1 * e1 + e2 * e2));      // euclideanDistance: (p1 - p2).norm() in fp32
        const float cdist = sqrtf(g0 * g0 + (g1 * g1 + g2 * g2));
        acc[0] += (double)dist / (double)cdist;                        // :117-119 (double division)
    }
    block_reduce<1>(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}
__global__ void k_fontana_finish(const double* __restrict__ partials, int nblocks, int n, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; b++) s += partials[b];
    *out = (float)(s / (double)n);
}

// RANDOM_SAMPLING selection (selection.h:88-106): every point of the current (possibly decimated) cloud is kept with
// probability p, independently per iteration.  The reference draws from std::mt19937 seeded by random_device; here the
// decision is a counter-based hash of (seed, iteration, original point index), identical on host and device, and the
// kept points are compacted in increasing order (stable, deterministic): block counts -> scan -> scatter.
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
__host__ __device__ __forceinline__ uint32_t select_hash(uint32_t seed, uint32_t iteration, uint32_t index) {
    return fmix32(index * 0x9E3779B9u + fmix32(seed + iteration * 0x7F4A7C15u + 0x165667B1u));
}
__global__ __launch_bounds__(256) void k_select_count(const int* __restrict__ base, int n, uint32_t seed, uint32_t iteration, uint32_t threshold, int take_all,
                                                      int* __restrict__ block_counts) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (t < n) { const int i = base ? base[t] : t; keep = take_all || select_hash(seed, iteration, (uint32_t)i) < threshold; }
    const int c = __syncthreads_count(keep ? 1 : 0);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = c;
}
__global__ __launch_bounds__(1024) void k_select_scan(int* __restrict__ block_counts, int nblocks, int* __restrict__ total_out) {
    __shared__ int carry;
    __shared__ int tmp[1024];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const int v = b < nblocks ? block_counts[b] : 0;
        tmp[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {             // Hillis-Steele inclusive scan
            const int a = threadIdx.x >= off ? tmp[threadIdx.x - off] : 0;
            __syncthreads();
            tmp[threadIdx.x] += a;
            __syncthreads();
        }
        if (b < nblocks) block_counts[b] = carry + tmp[threadIdx.x] - v;     // exclusive offset of block b
        __syncthreads();
        if (threadIdx.x == 1023) carry += tmp[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}
__global__ __launch_bounds__(256) void k_select_scatter(const int* __restrict__ base, int n, uint32_t seed, uint32_t iteration, uint32_t threshold, int take_all,
                                                        const int* __restrict__ block_offsets, int* __restrict__ out) {
    __shared__ int wave_off[4];
    const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int i = 0; bool keep = false;
    if (t < n) { i = base ? base[t] : t; keep = take_all || select_hash(seed, iteration, (uint32_t)i) < threshold; }
    const unsigned long long m = __ballot(keep);
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    if (lane == 0) wave_off[w] = __popcll(m);
    __syncthreads();
    int off = block_offsets[blockIdx.x];
    for (int v = 0; v < w; v++) off += wave_off[v];
    if (keep) out[off + rank] = i;
}

// PointCloud(depthMap, colorFrame, K, extrinsics, width, height, ...) (PointCloud.h:78-165): back-projection of a depth
// image and central-difference normals, the step in front of the ICP loop for RGB-D input.  One lane = one pixel; output is
// organised (invalid = MINF), `valid` marks what the keepOriginalSize = false filter keeps (:148-152).
// Quirks kept: normals are NOT rotated by the extrinsics (:128-129); the colour of pixel i is read from bytes i..i+3 of the
// RGBX frame instead of 4i..4i+3 (:156-157) unless fix_color_index is set.
__global__ void k_backproject(const float* __restrict__ depth, const uint8_t* __restrict__ rgbx, int width, int height,
                              float fx, float fy, float cx, float cy, const float* __restrict__ inv /* 3x3 row-major R^-1, then t^-1 */,
                              float max_distance_halved, int fix_color_index,
                              float* __restrict__ xyz, float* __restrict__ nrm, uint8_t* __restrict__ rgba, uint8_t* __restrict__ valid) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = width * height;
    if (idx >= n) return;
    const int v = idx / width, u = idx - v * width;
    const float d = depth[idx];
    float p0 = -INFINITY, p1 = -INFINITY, p2 = -INFINITY;
    if (d != -INFINITY) {                                           // :104-110
        const float a = ((float)u - cx) / fx * d, b = ((float)v - cy) / fy * d, c = d;
        p0 = (inv[0] * a + (inv[1] * b + inv[2] * c)) + inv[9];
        p1 = (inv[3] * a + (inv[4] * b + inv[5] * c)) + inv[10];
        p2 = (inv[6] * a + (inv[7] * b + inv[8] * c)) + inv[11];
    }
    float n0 = -INFINITY, n1 = -INFINITY, n2 = -INFINITY;
    if (v >= 1 && v < height - 1 && u >= 1 && u < width - 1) {      // :117-131, borders stay MINF (:134-141)
        const float du = 0.5f * (depth[idx + 1] - depth[idx - 1]);
        const float dv = 0.5f * (depth[idx + width] - depth[idx - width]);
        if (isfinite(du) && isfinite(dv) && !(fabsf(du) > max_distance_halved) && !(fabsf(dv) > max_distance_halved)) {
            const float x = -du, y = -dv, z = 1.f;
            const float sq = x * x + (y * y + z * z);
            const float len = sqrtf(sq);
            n0 = x / len; n1 = y / len; n2 = z / len;
        }
    }
    xyz[(size_t)idx * 3] = p0; xyz[(size_t)idx * 3 + 1] = p1; xyz[(size_t)idx * 3 + 2] = p2;
    nrm[(size_t)idx * 3] = n0; nrm[(size_t)idx * 3 + 1] = n1; nrm[(size_t)idx * 3 + 2] = n2;
    if (rgba && rgbx) {
        const size_t base = fix_color_index ? (size_t)idx * 4 : (size_t)idx;
        const size_t last = (size_t)n * 4 - 1;
#pragma unroll
        for (int k = 0; k < 4; k++) rgba[(size_t)idx * 4 + k] = rgbx[base + k <= last ? base + k : last];
    }
    if (valid) valid[idx] = (finite3(p0, p1, p2) && finite3(n0, n1, n2)) ? 1 : 0;
}

// utils.h:106-133 as stand-alone kernels for the adaptor's transformPoints / transformNormals
__global__ void k_transform_aos(const float* __restrict__ in, int n, const PoseState* __restrict__ ps, int normals, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float x = in[(size_t)k * 3], y = in[(size_t)k * 3 + 1], z = in[(size_t)k * 3 + 2];
    float a, b, c;
    if (normals) xform_normal(ps->nmat, x, y, z, a, b, c); else xform_point(ps->pose, x, y, z, a, b, c);
    out[(size_t)k * 3] = a; out[(size_t)k * 3 + 1] = b; out[(size_t)k * 3 + 2] = c;
}

}  // namespace icpdev

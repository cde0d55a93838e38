// dev_bvh.hpp -- exact kd-ordered BVH: build kernels, 4-wide nodes, the walk (per lane, and shared over the wave), k_knn_bvh.
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.
// ------------------------------------------------------------------------------------------------
// Exact kd-ordered BVH 1-NN: the index the reference builds once per pair (NearestNeighbor.h:122-141 / :209-232, a FLANN
// kd-tree over xyz or over the 6-D xyz+rgb/255 features) rebuilt on the device as a balanced kd-tree in implicit heap
// layout, queried with the SAME fp32 distance and the same lexicographic (d2, lowest index) argmin as k_knn_brute<DIM> --
// bit-identical results, O(log M) nodes per query instead of M distance evaluations.  DIM = 3 or 6.
//   build : level by level, every node's points are sorted along the widest axis of the node's bounding box; the implicit
//           node k covers a fixed, leaf-aligned slice of the array, so the count-balanced median split is simply "first
//           half / second half".  Slices > 2048 points: one rocPRIM sort per level over keys (node id << 32 | ordered
//           coordinate bits); below that, all remaining levels in one LDS kernel (k_bvh_block_levels).  Leaves hold
//           BVH_LEAF points SoA + original indices; binary node records hold BOTH child boxes, pair-interleaved for
//           packed-f32 math, filled bottom-up; the 1-NN walk uses 4-wide nodes derived from them (two levels collapsed)
//           and 32-byte target records in leaf order.
//   query : one lane = one query, depth-first "nearest child first".  A node is skipped only if its box lower bound
//           exceeds the running best times ICP_PRUNE_SLACK (1.001, see there); the bound uses the same operation sequence as the
//           point distance, so by monotonicity of IEEE rounding fl(d2(point)) >= fl(lb) for every point in the box -- any factor
//           >= 1 keeps the search exact.  Equal distances resolve to the lowest original index, exactly like the strict-< scan
//           (NearestNeighbor.h:87).
#ifndef ICP_ISEL_NATURAL
#define ICP_ISEL_NATURAL 0        // 1: the natural spellings of two expressions in k_bvh_block_levels that crash the ROCm 7.2 gfx950 instruction selector
#endif
#ifndef ICP_SEED_DESCENT
#define ICP_SEED_DESCENT 1
#endif
#ifndef ICP_PRUNE_SLACK
// A box is skipped when its (squared) bound exceeds best * ICP_PRUNE_SLACK.  Anything above 1 + a few ulp is exact.  But a box skipped with a
// bound barely above the neighbour's distance leaves the query a bound on "everything else" with no margin, and it can then never be
// verified without a search: 1.001 (0.05 % in distance, far more than a query moves per iteration once ICP has converged) costs a
// handful of extra box visits and retires those queries.  (Until round 2: 1.00002.)
#define ICP_PRUNE_SLACK 1.001f
#endif
#ifndef ICP_PREFETCH_PATH
// 1: a seeded walk first touches the nodes of its seed's root-to-leaf path (quad_prefetch_path).  It paid while every lane walked alone
// (round 1 / early round 2); with the shared walk it no longer does (iterations 1-9 0.0567 ms without, 0.0578 with): off.
#define ICP_PREFETCH_PATH 0
#endif
constexpr int BVH_LEAF = 8;
#ifndef ICP_BVH_THREADS
#define ICP_BVH_THREADS 256
#endif
constexpr int BVH_THREADS = ICP_BVH_THREADS;    // threads per block of the BVH matchers
#ifndef ICP_QUERIES_PER_WAVE
#define ICP_QUERIES_PER_WAVE 64  // fused matcher: queries per wave; below 64 the remaining lanes carry no query of their own and only help (shared walk)
#endif
constexpr int BVH_QPW = ICP_QUERIES_PER_WAVE;
constexpr int BVH_QPB = BVH_THREADS / 64 * BVH_QPW;     // queries per block of the fused matcher
__host__ __device__ inline int fused_nblocks(int n) { return (n + BVH_QPB - 1) / BVH_QPB; }

template <int DIM> struct BvhNodeT { float lo[DIM][2]; float hi[DIM][2]; float pad[DIM == 3 ? 4 : 8]; };   // 64 B / 128 B
template <int DIM> struct BvhLeafT { float c[DIM][BVH_LEAF]; int idx[BVH_LEAF]; float pad[DIM == 3 ? 0 : 8]; };   // 128 B / 256 B
template <> struct BvhLeafT<3> { float c[3][BVH_LEAF]; int idx[BVH_LEAF]; };
typedef BvhNodeT<3> BvhNode;
typedef BvhLeafT<3> BvhLeaf;

// 4-wide node of the same tree with two binary levels collapsed: the boxes of the four grandchildren, SoA per axis (two
// packed-f32 pairs each).  Half the dependent loads per query of the binary walk -- the search is bound by the latency of
// that chain, not by bytes or flops.
template <int DIM> struct BvhQuadT { float lo[DIM][4]; float hi[DIM][4]; float pad[DIM == 3 ? 8 : 16]; };   // padded to one / two 128-byte lines

// (Round 2 measured a QUANTISED form of these nodes -- child boxes on a uniform 14-bit grid, 48 B per node, integer box test with
// saturating packed-u16 arithmetic, conservative by construction -- and dropped it: half the node loads, 60 % more VALU per node,
// iterations 1-9 unchanged, the unseeded iteration twice as slow.  The walk is not bound by L1 bytes.  See DESIGN.md section 4; the
// code is in the history: commit "quantised 4-wide nodes and XCD chunking as build variants".)
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
    const int* pos_of;            // [M] position (8 * leaf + slot) by original index (the inverse of recs[].idx; -1: not in the tree)
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

// ---- upper levels from PRESORTED axes ---------------------------------------------------------------------------------------
// Round 1 re-sorted the whole array once per upper level (key = node id | coordinate along the node's widest axis: one global sort
// of ~120 us per level, 8 levels).  The classic kd-tree construction needs each axis sorted only ONCE: keep one index list per axis,
// all of them grouped by node; per level every node reads its box from the ends of its segments (first / last element of each
// axis list), picks the widest axis, declares the first half of THAT list its left child, and every list is stably partitioned by
// that left / right flag (exclusive scan of the flags + scatter) so that it stays sorted inside both children.  DIM sorts of
// 32-bit keys once, then per level DIM x (scan + scatter) -- and the result is the same kind of tree (median split by count along
// the widest axis; ties now keep index order).
template <int DIM> struct AxisLists { const int* L[DIM]; };
__global__ void k_axis_keys(const float* __restrict__ plane, const int* __restrict__ ids, int nv, unsigned int* __restrict__ keys) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nv) keys[t] = ordered_bits(plane[ids[t]]);
}
template <int DIM>
__global__ void k_presort_axis(const CoordPtrs<DIM> cp, const AxisLists<DIM> al, int nv, int seg_shift, int n_nodes, unsigned char* __restrict__ axis_of_node) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const long long start = (long long)i << seg_shift;
    int axis = 0;
    if (start < nv) {
        const int end = (int)min((long long)nv, start + (1ll << seg_shift));
        float ext = -1.f;
#pragma unroll
        for (int k = 0; k < DIM; k++) {
            const float e = cp.c[k][al.L[k][end - 1]] - cp.c[k][al.L[k][(int)start]];
            if (e > ext) { ext = e; axis = k; }
        }
    }
    axis_of_node[i] = (unsigned char)axis;
}
template <int DIM>
__global__ void k_presort_side(const AxisLists<DIM> al, int nv, int seg_shift, const unsigned char* __restrict__ axis_of_node, unsigned char* __restrict__ side /* by point id */) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nv) return;
    const int i = t >> seg_shift, a = axis_of_node[i];
    int j = al.L[0][t];
#pragma unroll
    for (int k = 1; k < DIM; k++) { const int jk = al.L[k][t]; j = (a == k) ? jk : j; }      // (values selected, not pointers: see k_bvh_block_levels)
    side[j] = ((t - (i << seg_shift)) >= (1 << (seg_shift - 1))) ? 1 : 0;
}
// The stable partition of all DIM lists of one level in three launches (grid.y = axis): per 256-position block the number of
// right-child entries, an exclusive scan of those block counts (one block per list), and the scatter that recomputes its block's
// flags, ranks them with ballots and adds the block and segment offsets.  (A library scan per list cost two launches and ~12 us
// each, three lists and eight levels: more than the sorts it replaced were worth.)  Segments are multiples of 4096 positions, so a
// segment always starts on a block boundary: the number of right entries before it is a block offset.
constexpr int PRS_THREADS = 256;
template <int DIM>
__global__ __launch_bounds__(PRS_THREADS) void k_presort_count(const AxisLists<DIM> al, const unsigned char* __restrict__ side, int nv, int* __restrict__ blk_cnt /* [DIM][nblk] */) {
    __shared__ int wc[PRS_THREADS / WAVE];
    const int k = blockIdx.y, t = blockIdx.x * PRS_THREADS + threadIdx.x;
    int j = 0;
#pragma unroll
    for (int q = 0; q < DIM; q++) { if (q == k) j = t < nv ? al.L[q][t] : 0; }       // (value selected per axis, not the pointer)
    const bool f = t < nv && side[j] != 0;
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[(size_t)k * gridDim.x + blockIdx.x] = (wc[0] + wc[1]) + (wc[2] + wc[3]);
}
__global__ __launch_bounds__(1024) void k_presort_blockscan(const int* __restrict__ blk_cnt, int nblk, int* __restrict__ blk_off /* exclusive */) {
    __shared__ int part[1024];
    const int* in = blk_cnt + (size_t)blockIdx.x * nblk; int* out = blk_off + (size_t)blockIdx.x * nblk;
    const int per = (nblk + 1023) / 1024, b0 = threadIdx.x * per;
    int sum = 0;
    for (int q = 0; q < per; q++) if (b0 + q < nblk) sum += in[b0 + q];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                  // Hillis-Steele over the 1024 thread sums
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;                   // exclusive prefix of this thread's chunk
    for (int q = 0; q < per; q++) if (b0 + q < nblk) { out[b0 + q] = run; run += in[b0 + q]; }
}
template <int DIM>
__global__ __launch_bounds__(PRS_THREADS) void k_presort_scatter(const AxisLists<DIM> al, const unsigned char* __restrict__ side, const int* __restrict__ blk_off, int nv, int seg_shift,
                                                                 AxisLists<DIM> out_lists) {
    __shared__ int wc[PRS_THREADS / WAVE];
    const int k = blockIdx.y, t = blockIdx.x * PRS_THREADS + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int j = 0;
#pragma unroll
    for (int q = 0; q < DIM; q++) { if (q == k) j = t < nv ? al.L[q][t] : 0; }
    const bool f = t < nv && side[j] != 0;
    const unsigned long long m = __ballot(f);
    if (lane == 0) wc[w] = __popcll(m);
    __syncthreads();
    int before = __popcll(m & ((1ull << lane) - 1ull));
    for (int ww = 0; ww < w; ww++) before += wc[ww];
    if (t >= nv) return;
    const int* off = blk_off + (size_t)k * gridDim.x;
    const int S = (t >> seg_shift) << seg_shift;          // segment start: a multiple of 4096, i.e. of the block size
    const int r_before = off[blockIdx.x] + before - off[S / PRS_THREADS];
    const int cnt = min(1 << seg_shift, nv - S), nleft = min(1 << (seg_shift - 1), cnt);
    const int pos = f ? S + nleft + r_before : S + (t - S - r_before);
#pragma unroll
    for (int q = 0; q < DIM; q++) { if (q == k) ((int*)out_lists.L[q])[pos] = j; }
}

// The lower levels of the build in one kernel.  Once a node's slice is <= 2048 points the remaining levels only permute
// points INSIDE that slice, so a block takes 2048 consecutive positions into LDS and runs all of them there: per level the
// boxes of the slices (shuffle tree + at most 4 wave boxes), the widest axis, and a bitonic sort of every slice on
// (ordered coordinate bits, position) -- the position makes the keys unique and reproduces the stable order of the global
// radix sort, so the tree is exactly the one the level-by-level build produces.  Replaces 8 global sorts (~120 us each).
constexpr int BLV_POINTS = 2048, BLV_THREADS = 256, BLV_PER = BLV_POINTS / BLV_THREADS;
template <int DIM>
__global__ __launch_bounds__(BLV_THREADS) void k_bvh_block_levels(const CoordPtrs<DIM> cp, const int* __restrict__ perm_in, int n_valid, int first_shift /* <= 11 */,
                                                                  int* __restrict__ perm_out) {
    __shared__ unsigned long long keys[BLV_POINTS];
    __shared__ int vals[BLV_POINTS];
    __shared__ unsigned int wbox[BLV_THREADS / WAVE][2 * DIM];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = blockIdx.x * BLV_POINTS;
#pragma unroll
    for (int q = 0; q < BLV_PER; q++) { const int pos = tid * BLV_PER + q, gi = base + pos; vals[pos] = gi < n_valid ? perm_in[gi] : -1; }
    __syncthreads();
    for (int sh = first_shift; sh >= 4; sh--) {            // slices of 2^sh positions; the last level splits 16 -> two leaves of 8
        const int S = 1 << sh, g = S / BLV_PER;             // g threads per slice (2 .. 256), aligned
        unsigned int lo[DIM], hi[DIM];
#pragma unroll
        for (int k = 0; k < DIM; k++) { lo[k] = 0xFFFFFFFFu; hi[k] = 0u; }
        for (int q = 0; q < BLV_PER; q++) {
            const int j = vals[tid * BLV_PER + q];
            if (j >= 0) {
#pragma unroll
                for (int k = 0; k < DIM; k++) { const unsigned int o = ordered_bits(cp.c[k][j]); lo[k] = min(lo[k], o); hi[k] = max(hi[k], o); }
            }
        }
        for (int off = 1; off < g && off < WAVE; off <<= 1) {
#pragma unroll
            for (int k = 0; k < DIM; k++) { lo[k] = min(lo[k], (unsigned int)__shfl_xor((int)lo[k], off, WAVE)); hi[k] = max(hi[k], (unsigned int)__shfl_xor((int)hi[k], off, WAVE)); }
        }
        if (g > WAVE) {                                    // slice spans 2 or 4 waves
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < DIM; k++) { wbox[w][k] = lo[k]; wbox[w][DIM + k] = hi[k]; }
            }
            __syncthreads();
            const int nw = g / WAVE, w0 = w & ~(nw - 1);
            for (int ww = w0; ww < w0 + nw; ww++) {
#pragma unroll
                for (int k = 0; k < DIM; k++) { lo[k] = min(lo[k], wbox[ww][k]); hi[k] = max(hi[k], wbox[ww][DIM + k]); }
            }
        }
        int axis = 0; float ext = -1.f;
#pragma unroll
        for (int k = 0; k < DIM; k++) {
            // from_ordered_bits spelled with ^ instead of & 0x7FFFFFFF: the other spelling crashes the gfx950 instruction selector here
            // (ROCm 7.2).  tests/test_compiler_workarounds.py compiles the natural spellings (ICP_ISEL_NATURAL=1) and reports whether
            // the crash is still there; the GPU parity tests of the BVH cover the results of this spelling.
            const unsigned int uh = hi[k], ul = lo[k];
#if ICP_ISEL_NATURAL
            const float fh = from_ordered_bits(uh), fl = from_ordered_bits(ul);
#else
            const float fh = __uint_as_float((uh & 0x80000000u) ? (uh ^ 0x80000000u) : ~uh), fl = __uint_as_float((ul & 0x80000000u) ? (ul ^ 0x80000000u) : ~ul);
#endif
            const float e = fh - fl;
            if (e > ext) { ext = e; axis = k; }
        }
        for (int q = 0; q < BLV_PER; q++) {
            const int pos = tid * BLV_PER + q, j = vals[pos];
            unsigned long long key = ~0ull;
            if (j >= 0) {
#if ICP_ISEL_NATURAL
                const float c = cp.c[axis][j];
#else
                float c = cp.c[0][j];
#pragma unroll
                for (int k = 1; k < DIM; k++) c = (axis == k) ? cp.c[k][j] : c;      // (selecting the plane POINTER instead crashes the gfx950 instruction selector)
#endif
                key = ((unsigned long long)ordered_bits(c) << 32) | (unsigned int)pos;
            }
            keys[pos] = key;
        }
        __syncthreads();
        for (int k = 2; k <= S; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
                for (int pp = 0; pp < BLV_POINTS / 2 / BLV_THREADS; pp++) {
                    const int idx = tid + BLV_THREADS * pp;
                    const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1)), ip = i + j;
                    const bool asc = ((i & k) == 0) || (k == S);
                    const unsigned long long a = keys[i], b = keys[ip];
                    if ((a > b) == asc) { keys[i] = b; keys[ip] = a; const int va = vals[i]; vals[i] = vals[ip]; vals[ip] = va; }
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int q = 0; q < BLV_PER; q++) { const int pos = tid * BLV_PER + q, gi = base + pos; if (gi < n_valid) perm_out[gi] = vals[pos]; }
}

template <int DIM>
__global__ void k_bvh_gather(const CoordPtrs<DIM> cp, const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz, const uint32_t* __restrict__ rgba,
                             const int* __restrict__ sorted_idx, int n_valid, int n_slots, BvhLeafT<DIM>* __restrict__ leaves, TgtRec* __restrict__ recs, int* __restrict__ pos_of) {
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
        pos_of[j] = i;
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

// Square roots for BOUNDS: one v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence (17 instructions), with the rounding
// absorbed by the safety factor every bound carries anyway (1e-6 = 8 ulp).  sqrt_up never underestimates -- a denormal argument, which
// the instruction may flush to zero, is raised to FLT_MIN first; sqrt_dn never overestimates (a flush to zero is a valid lower bound).
__device__ __forceinline__ float sqrt_up(float x) { return __builtin_amdgcn_sqrtf(fmaxf(x, 1.17549435e-38f)) * 1.000001f; }
__device__ __forceinline__ float sqrt_dn(float x) { return __builtin_amdgcn_sqrtf(x) * 0.999999f; }

// What a search remembers about the points that did NOT win, for next iteration's verify tests: b2 = the smallest distance among them
// and l2 = the leaf that point lives in (the runner-up's leaf; it can be the winner's own leaf), b3 = a lower bound on the non-winners
// outside leaf l2.  An entry is (distance, leaf); entries of leaf l2 other than the runner-up itself may or may not reach b3 -- either
// way b3 stays a valid (if smaller) bound.
__device__ __forceinline__ void others_insert(float x, int xl, float& b2, int& l2, float& b3) {
    const bool better = x < b2, same = xl == l2;
    b3 = same ? b3 : fminf(b3, better ? b2 : x);
    l2 = better ? xl : l2; b2 = better ? x : b2;
}

// Evaluate the 8 points of a leaf against the lane's query; exact lexicographic (d2, lowest index) update.  The leaf's non-winners
// enter (b2, l2, b3) as ONE entry (their minimum); when the win moves here from another leaf, the dethroned winner -- the minimum of
// the leaf it lives in, as far as this search has seen it -- enters under that leaf.
template <int DIM>
__device__ __forceinline__ void leaf_eval(const BvhLeafT<DIM>* __restrict__ lf, int leaf, const f2* p2, float& best, int& bi, int& bpos, float& b2, int& l2, float& b3) {
    const int prev_leaf = bpos >> 3; const float prev_best = best;
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
        float mo = FLT_MAX;              // smallest distance among this leaf's points that do not end up as the winner
        bool here = prev_leaf == leaf;   // the running winner lives in this leaf
#pragma unroll
        for (int t = 0; t < BVH_LEAF; t++) {
            const int j = lf->idx[t];
            const bool take = (dd[t] < best) | ((dd[t] == best) & (j < bi));     // first minimum = lowest original index
            const float other = take ? (here ? best : FLT_MAX) : ((j != bi) ? dd[t] : FLT_MAX);      // a winner of this leaf dethroned by a later one, or a plain non-winner
            mo = fminf(mo, other);
            here = here | take;
            best = take ? dd[t] : best; bi = take ? j : bi; bpos = take ? leaf * BVH_LEAF + t : bpos;
        }
        others_insert(mo, leaf, b2, l2, b3);
        if (prev_leaf != leaf && (bpos >> 3) == leaf) others_insert(prev_best, prev_leaf, b2, l2, b3);      // the win moved here (an unseeded start enters (FLT_MAX, -1): nothing)
    } else others_insert(m, leaf, b2, l2, b3);      // nobody here can win: all 8 are "others"
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
// 4 pending-child bits per level in one mask register (see quad_run for why bits are enough).
template <class MaskT> struct QuadStateT { int L; int idx; MaskT pending; bool alive; };     // MaskT: 32 bits hold 8 levels, 64 bits 16

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

// The query as the node and leaf tests need it: packed fp32 pairs.
template <int DIM> struct QueryPt { f2 p2[DIM]; };
template <int DIM>
__device__ __forceinline__ void make_query(const BvhViewT<DIM>& bv, const float* p, QueryPt<DIM>& q) {
#pragma unroll
    for (int k = 0; k < DIM; k++) { q.p2[k].x = p[k]; q.p2[k].y = p[k]; }
}
// lower bounds of the four children of 4-wide node `node` (index over all levels)
template <int DIM>
__device__ __forceinline__ void quad_lb_at(const BvhViewT<DIM>& bv, unsigned int node, const QueryPt<DIM>& q, f2& l01, f2& l23) {
    quad_lb<DIM>(bv.qnodes + node, q.p2, l01, l23);
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

// The pending siblings are remembered as BITS only (4 per level, one register), not with their bounds.  A seeded walk is in
// effect a range query -- its prune threshold is almost final from the first step on -- so re-testing a parked sibling against the
// improved threshold when it is popped (round 1 kept 16-bit bounds in LDS for that) buys next to nothing, while the packing, the LDS
// round trip and the unpack / min / select of every pop were a third of the walk's instructions (measured: iterations 1-9 0.091 ->
// 0.083 ms, the always-walk loop 9.3 k -> 10.6 k iterations/s).  With bits only a pop is "deepest level with a bit set, lowest bit":
// no loop, no LDS.  A sibling that went stale is noticed one node later (its own children all fail the test), which only costs
// that node; exactness is untouched (a box is still skipped only on lb > thr, and every skipped box enters minlb when tested).
template <class MaskT>
__device__ __forceinline__ void quad_pop_bits(QuadStateT<MaskT>& st) {
    if (!st.alive && st.pending) {
        const int top = sizeof(MaskT) == 8 ? 63 - __clzll((long long)st.pending) : 31 - __clz((int)st.pending);
        const int lv = top >> 2;                                                  // deepest level with pending children
        const unsigned int bits = (unsigned int)(st.pending >> (4 * lv)) & 0xFu;
        const int c = __ffs((int)bits) - 1;
        st.pending &= ~((MaskT)1 << (4 * lv + c));
        st.idx = ((st.idx >> (2 * (st.L - lv))) << 2) | c; st.L = lv + 1; st.alive = true;
    }
}
#ifndef ICP_DEBUG_STEPS
#define ICP_DEBUG_STEPS 0        // 1: development build that records nodes + leaves visited per query (icp_debug_steps)
#endif
#if ICP_DEBUG_STEPS
#define ICP_COUNT_STEP(x) ((x)++)
__device__ int g_dbg_nodes_dummy;
#else
#define ICP_COUNT_STEP(x)
#endif
template <int DIM, class MaskT>
__device__ __forceinline__ void quad_run(const BvhViewT<DIM>& bv, const QueryPt<DIM>& qp, QuadStateT<MaskT>& st,
                                         float& best, int& bi, int& bpos, float& b2, int& l2, float& b3, float& minlb, int& dbg_nodes, int& dbg_leaves) {
    const int Lq = bv.Lq;
    // A box is skipped when its lower bound exceeds thr = best * ICP_PRUNE_SLACK (clamped so that the +inf bound of an empty box is
    // always skipped): that implies bound > best with margin, one multiply per change of `best` instead of one per box test.
    float thr = fminf(best * ICP_PRUNE_SLACK, FLT_MAX);
    // smallest skipped bound, kept as its bit pattern: bounds are >= +0, so unsigned order is value order and the integer minimum
    // needs none of the NaN canonicalisation a float minimum of selected values drags in
    unsigned int mlb = __float_as_uint(minlb);
    constexpr unsigned int NONE = 0x7F800000u;            // +inf
    while (st.alive) {
        while (st.alive && st.L < Lq) {
            f2 l01, l23;
            ICP_COUNT_STEP(dbg_nodes);
            quad_lb_at<DIM>(bv, (0x55555555u & ((1u << (2 * st.L)) - 1u)) + (unsigned int)st.idx, qp, l01, l23);
            const float m = fminf(fminf(l01.x, l01.y), fminf(l23.x, l23.y));
            const bool s0 = !(l01.x > thr), s1 = !(l01.y > thr), s2 = !(l23.x > thr), s3 = !(l23.y > thr);
            mlb = min(min(mlb, min(s0 ? NONE : __float_as_uint(l01.x), s1 ? NONE : __float_as_uint(l01.y))), min(s2 ? NONE : __float_as_uint(l23.x), s3 ? NONE : __float_as_uint(l23.y)));   // skipped right here
            if (!(m > thr)) {
                // nearest child first (selects, not branches).  Measured: taking the survivors in index order instead saves 5 instructions
                // per node and costs 0.080 -> 0.096 ms in iterations 1-9 (0.17 -> 0.80 ms unseeded): the order is worth its price.
                const bool c0 = l01.x == m, c1 = l01.y == m, c2 = l23.x == m;
                int c = 3; c = c2 ? 2 : c; c = c1 ? 1 : c; c = c0 ? 0 : c;
                const unsigned int pend = ((s0 ? 1u : 0u) | (s1 ? 2u : 0u) | (s2 ? 4u : 0u) | (s3 ? 8u : 0u)) & ~(1u << c);
                st.pending |= (MaskT)pend << (4 * st.L);
                st.idx = (st.idx << 2) | c; st.L++;
            } else st.alive = false;                      // all four children pruned
            quad_pop_bits(st);
        }
        if (st.alive) {
            ICP_COUNT_STEP(dbg_leaves);
            leaf_eval<DIM>(bv.leaves + st.idx, st.idx, qp.p2, best, bi, bpos, b2, l2, b3);
            thr = fminf(best * ICP_PRUNE_SLACK, FLT_MAX);
            st.alive = false;
            quad_pop_bits(st);
        }
    }
    minlb = __uint_as_float(mlb);
}

// XCD-aware block mapping: workgroups are dealt round-robin over the 8 XCDs (block b runs on the XCD group b % 8), each
// with a private 4 MiB L2.  With Morton-sorted queries, giving an XCD consecutive blocks of the order means its L2 only has to hold
// the part of the tree under them.  Round 1 gave every XCD ONE contiguous eighth of the order; but the hard queries (far from the
// target, long walks) come in runs, so some eighths hold 1.5 x the work of others (per-slice walk time, tools/dev_wave_times.py)
// and the launch waits for that XCD.  Chunks of 16 blocks (2048 queries) dealt to the XCDs in turn keep the locality and even out
// the work: iterations 1-9 0.0615 -> 0.058 ms (chunks of 1 / 4 / 8 / 32 / 64: 0.061 / 0.0595 / 0.059 / 0.058 / 0.059).
// Measured on top of that and dropped: a wave made of 2 / 4 / 8 / 16 runs of consecutive queries from as many places of its chunk
// instead of 64 consecutive ones (evens out the waves, costs coherence: no gain beyond noise).
#ifndef ICP_XCD_CHUNK
#define ICP_XCD_CHUNK 16         // chunks of this many consecutive blocks dealt to the XCDs in turn; 0: one contiguous slice per XCD
#endif
__device__ __forceinline__ int xcd_contiguous_block(int b, int nb) {
#if ICP_XCD_CHUNK
    {
        constexpr int C = ICP_XCD_CHUNK;
        const int full = nb / (8 * C) * (8 * C);                      // the part of the grid that deals out evenly; the rest keeps its order
        if (b >= full) return b;
        const int x = b & 7, j = b >> 3;
        return ((j / C) * 8 + x) * C + j % C;
    }
#endif
    const int q = nb >> 3, r = nb & 7, x = b & 7, j = b >> 3;        // XCD group x owns q (+1 if x < r) consecutive logical blocks
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// Every lane walks the tree on its own for its own query.  Measured on MI355X (370k x 370k, DIM 3): 12.5 4-wide nodes and
// 3.1 leaves per walked query, ~40 % of the lanes active on average (walk lengths differ per lane: 15.6 steps on average,
// ~70 for the longest), a third of the wave time waiting on dependent loads.  The kernel lasts as long as its longest walks.
// What moved it: kd-ordered tree, Morton-sorted queries + XCD-contiguous slices, compact per-level stack in LDS, temporal
// seeding, the verify-and-skip test below (which retires whole waves without a walk once ICP has converged), 4-wide nodes.
// Measured and not adopted: wave-packet traversal with scalar node loads, persistent lanes with
// wave-level refill, a second cooperative pass for over-budget queries, block-level re-packing of the walking queries, a
// separate verify pass + packed walk pass, raised wave priority for long walkers.
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
        const float delta = sqrt_up((ex * ex + ey * ey) + ez * ez);
        const float lbn = (s.w - delta) * 0.999999f;
        if (sqrt_up(best) < lbn) { lb_others = lbn; return true; }
    }
    return false;
}

template <int DIM>
__device__ __forceinline__ void knn_store_state(const KnnParams& kp, int k, const float* p, float best, int bpos, float lb_others, float lb3, int l2) {
    if (kp.qstate) { float4 s; s.x = p[0]; s.y = p[1]; s.z = p[2]; s.w = lb_others; kp.qstate[k] = s; }
    if (kp.qstate2) { float2 t; t.x = lb3; t.y = __int_as_float(l2); kp.qstate2[k] = t; }
    if (kp.nn_raw) kp.nn_raw[k] = bpos;
    if (kp.d2_out) kp.d2_out[k] = best;
}

#ifndef ICP_SHARE_WALKS
#define ICP_SHARE_WALKS 1        // 1: the lanes of a wave that have nothing (left) to search take pending subtrees off the lanes that still walk
#endif
#ifndef ICP_SHARE_ROUNDS
#define ICP_SHARE_ROUNDS 2       // hand-over rounds per pass (each pairs the idle lanes with as many donors, one subtree per donor)
#endif
#ifndef ICP_SHARE_SPREAD
#define ICP_SHARE_SPREAD 1       // 1: few seeded walkers in a wave -> the levels of their seeds' paths are searched side by side by the idle lanes
#endif
#ifndef ICP_LONE_WALK
#define ICP_LONE_WALK 1          // 1: a wave with ONE (seeded) walker searches level-synchronously, two levels per dependent load (knn_walk_shared)
#endif
#ifndef ICP_SPREAD_TWO
#define ICP_SPREAD_TWO 1         // 1: ... and, while the lanes suffice, the levels of the path to the last search's runner-up leaf as well
#endif
#define ICP_SHARE_ROWS 10        // LDS rows (of NT uint2) the shared walk needs per wave
// The walks of one wave, shared.  A wave lasts as long as its longest walk while the lanes whose queries verified, or whose
// walks ended early, idle.  Here an idle lane adopts a parked subtree -- the SHALLOWEST pending sibling of a lane that still walks
// -- together with that lane's query and running best, searches it with the same code, and folds what it found into the owner's
// record in LDS: (distance, index) by a 64-bit atomic minimum (= the lexicographic minimum the lone walk computes), the bounds
// on all other points by 32-bit atomic minima of their bit patterns.  Every subtree is still searched by exactly one lane against
// a bound that is at least the final distance, so the result is the same exact neighbour; only the bounds used by NEXT iteration's
// verify tests may differ from the lone walk's (never larger than what is true).
//   What the merge keeps per owner, and why both verify tiers may use it (the invariant of the fold below): key = the minimum
//   (distance, index) over everything any searcher evaluated = the neighbour; key2 = the minimum (distance, leaf) over the runner-up
//   entries -- a searcher's own (b2, l2), its winner when that lost against key, a winner it dethroned; `rest` = a lower bound on every
//   evaluated or skipped target that is NEITHER the neighbour NOR covered by key2, where "covered" means: lies in key2's leaf and is at
//   least key2's distance away.  Every time an entry loses against another one it goes into `rest` unless it lives in the winning entry's
//   leaf (then that entry covers it); every skipped box goes into the separate skipped-bound minimum.  Hence, for the finished search:
//     lb_others = min(key2.distance, rest, skipped)  bounds every target but the neighbour              (first tier),
//     lb3       = min(rest, skipped)                 bounds every target outside the neighbour's leaf AND key2's leaf, plus possibly some
//                                                    inside them -- a smaller, still valid bound          (second tier, l2 = key2's leaf).
//   (A winner dethroned during the merge has no leaf on record and goes straight to `rest`: conservative for that one iteration.)
// Rows of the wave's LDS slots: 0 key, 1 key2 (runner-up entry), 2 (rest, position of the winner), 3-5 and 7-9 the lanes' own results,
// query and normal parked meanwhile (so that a helper needs no registers of its own for them), 6 the donors' lane numbers (first half)
// and the smallest skipped box bound per owner (second half).
// ---- the same hand-over between the WAVES of a block (XW) ------------------------------------------------------------------------
// A launch lasts as long as its slowest wave, and all waves of the grid are resident from the start: a wave that ends early frees a
// slot nobody fills (iteration 2 of configs[1]: mean wave end 48 us, launch 80 us).  With XW a block is made of waves from far-apart
// places of the query order (fused_matcher_body, ICP_WAVE_STRIDE), and a wave that has nothing left to do -- its own walks are over, or
// it had none -- adopts parked subtrees of the block's other waves through a small board in LDS:
//   board  : XW_SLOTS items (owner's thread | level, node, the donor's running best / index / position), state word per slot
//            0 free -> 2 (a donor lane claimed it by compare-and-swap, writes the item) -> 1 full -> 3 (a lane of an idle wave claimed it,
//            reads the item) -> 0.  An idle wave takes slot i with lane i; from there on the wave's own hand-over spreads the work
//            over its other lanes.
//   out[w] : parts of wave w's queries that are on the board or held by lanes of OTHER waves; a wave's queries are complete when all its
//            lanes are idle and out[w] = 0 (every fold into an owner's record comes before the decrement, LDS executes a wave's
//            operations in order, the owner reads its record after it has seen the zero).
//   nreg, ndone : waves that walk / whose queries are complete; a wave leaves when its own are complete and ndone >= nreg.  A wave
//            that starts late (nreg counts it only then) may find the others gone: it then simply takes its own items back.
//   idlew  : waves polling the board -- donors put items there only while somebody waits for them.
// The fold is the one of the wave-level hand-over (atomic minima in the owner's record), with one difference: the plain write of the
// winner's POSITION next to the 64-bit (distance, index) key is no longer ordered by lockstep, so an owner whose queries had help from
// another wave checks the record at that position against the key's index and, on a mismatch, takes the position from the inverse
// map (BvhViewT::pos_of).  Results are the exact neighbours as before; the bounds kept for the verify tests may differ from run to run
// (never larger than what is true), i.e. WHICH queries search in a later iteration may differ, not what they find.
#ifndef ICP_XW
#define ICP_XW 1                 // 1: blocks of more than one wave share walks between their waves (DIM 3)
#endif
#ifndef ICP_XW_SLEEP
#define ICP_XW_SLEEP 4
#endif
constexpr int XW_SLOTS = 64, XW_CTRL = 32;
// ---- ... and between the BLOCKS of the launch (GX) -------------------------------------------------------------------------------
// What remains after XW is the imbalance between CUs: a CU holds 5 or 6 blocks for the whole launch and their costs do not average out
// (iteration 2: the mean CU is done at 64 us, the last at 85).  A block whose waves walk for long POSTS parked subtrees in its own
// outbox in global memory (GX_SLOTS slots of 16 eight-byte granules, each granule self-validating like the pose hand-over: written
// once by a write-through store, all-ones = not there yet); wave 0 of a block that is finished with everything (its partial is stored)
// probes the outboxes of other blocks, claims a posted slot (compare-and-swap on the slot's claim word: 1 posted -> 2 claimed),
// searches the subtree with its 64 lanes like a query of its own (knn_walk_shared, MODE 3) and writes the result -- winner, runner-up
// entry, bound on the rest, smallest skipped bound, position -- into the slot's result granules.  The owner block folds that result
// into the query's record like the result of one of its own lanes (the slot counts in out[] of the owner's wave until then), takes
// back what nobody claimed (claim 1 -> 3), and leaves every slot it used as it found it: nothing to re-arm between launches.
//   slot granules: 0 (owner thread | level << 16, node)  1 (best, index)  2 (position, p.x)  3 (p.y, p.z)  4 claim word
//                  5 (index, best) of the winner  6 runner-up entry (leaf, distance)  7 (rest, skipped)  8 (position, 1)
// MEASURED AND NOT ADOPTED (round 3; compiled out, -DICP_GX=1 builds it, ICP_HIP_GX=0 switches it off at run time): exact -- the whole
// GPU suite passes with it -- but slower: 32.6 k against 33.3 k iterations/s with it switched off in the same binary, and that binary is
// itself 2.5 % behind the one without the code (34.1 k): the extra paths raise the walk's natural register demand from 74-78 to 85-89,
// which the 80-register budget turns into 28-44 bytes of scratch.  What the counters said (tools/dev_gx_counts.py): waves pass the
// posting threshold (16 passes of the hand-over loop) almost only in iteration 0; there 2 100 subtrees are posted, 260 claimed by helpers,
// 1 840 taken back by their owners; a helper needs a dozen probe rounds of 2-3 us to come across one of the ~ 60 posting blocks among
// 1 448.  To make it pay it would need: a posting criterion in TIME (waves still walking past the launch's mean), a list of the posting
// blocks instead of probing, and 10 registers.
#ifndef ICP_GX
#define ICP_GX 0
#endif
constexpr int GX_SLOTS = 32, GX_GRANULES = 16;
constexpr unsigned long long GX_EMPTY = ~0ull;
#if defined(ICP_DEBUG_TIMES) && ICP_DEBUG_TIMES
__device__ unsigned int g_gx_dbg[16];    // development builds: 0 posted, 1 claimed by helpers, 2 taken back, 3 results folded, 4 helper waves, 5 helper rounds, 6 helper rounds with a claim
#define GX_COUNT(i, n) atomicAdd(&g_gx_dbg[i], (unsigned int)(n))
#ifndef ICP_DEBUG_WALK_TRACE
#define ICP_DEBUG_WALK_TRACE 0
#endif
__device__ unsigned int g_walk_trace[64];   // ICP_DEBUG_WALK_TRACE: the last sparse walk (<= 3 walkers in the wave: spread start) of the launch: 0 walkers, 1 paths per walker, 2 clock at entry, 3 after the spread, 4 passes, 5 polls, 6 clock at exit, 8.. clock at the top of every pass (tools/dev_walk_trace.py)
#else
#define GX_COUNT(i, n)
#endif
constexpr int XW_INTS = XW_CTRL + XW_SLOTS + GX_SLOTS + XW_SLOTS * 5 + GX_SLOTS * 5;      // control words, slot states (board, outbox), items (2 x uint2 + int per slot), shadow of the posted items
constexpr int XW_SPIN_LIMIT = 1 << 22;                   // polls of an idle wave (~0.3 us each) before it gives up and raises the fault word
template <int DIM, int NT> constexpr bool xw_enabled() { return ICP_XW && DIM == 3 && NT > WAVE && NT / WAVE <= 8; }
template <int DIM, int NT> constexpr size_t xw_lds_bytes() { return xw_enabled<DIM, NT>() ? (size_t)XW_INTS * 4 : 0; }
// (block start, before the first __syncthreads of the kernel: thread t clears word t of the control block and the slot states)
template <int NT> __device__ __forceinline__ void xw_init(uint2* lbq, int tid) { if (tid < XW_CTRL + XW_SLOTS + GX_SLOTS) ((int*)(lbq + ICP_SHARE_ROWS * NT))[tid] = 0; }
// sc1 (write-through / L1-bypassing) accesses to the outboxes
__device__ __forceinline__ void gx_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long gx_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long gx_pack(unsigned int lo, unsigned int hi) { const unsigned long long g = ((unsigned long long)hi << 32) | lo; return g == GX_EMPTY ? g ^ 1ull : g; }
// is any wave of this block searching right now?  (a wave without walkers of its own asks once)
template <int NT> __device__ __forceinline__ bool xw_block_is_searching(uint2* lbq) {
    const int* xc = (const int*)(lbq + ICP_SHARE_ROWS * NT);
    const int nreg = __builtin_amdgcn_readfirstlane(__hip_atomic_load(xc + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)), ndone = __builtin_amdgcn_readfirstlane(__hip_atomic_load(xc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    return nreg > ndone;
}

// MODE 0: the wave on its own; 1: XW, the wave's own walks (it leaves when its queries are complete, and takes items off the board while it
// waits for parts of them that other waves hold); 2: XW, help only (xw_help: no queries of its own, rows 3.. of the wave untouched);
// 3: GX helper (gx_help): the wave on its own, lane by lane a SUBTREE (start_L, start_idx) of somebody else's query with that query's
// running best as the seed; the results come back raw -- best / bi / bpos = the winner, lb_others / l2o = the runner-up entry (squared
// distance, leaf), lb3 = the bound on the rest (squared), keep3[0] = the smallest skipped box bound (squared).
template <int DIM, int NT, class MaskT, int MODE = 0>
__device__ __forceinline__ void knn_walk_shared(const BvhViewT<DIM>& bv, float* p, float* keep3, bool need_walk, float& best, int& bi, int& bpos, float& lb_others, float& lb3, int& l2o,
                                                uint2* __restrict__ lbq, int tid, int* fault = nullptr, const GxParams* gx = nullptr, int gx_block = 0, int start_L = 0, int start_idx = 0) {
    constexpr bool XW = MODE == 1 || MODE == 2, HELP = MODE == 2, PROXY = MODE == 3;
    const bool GXON = ICP_GX && XW && gx && gx->slots;                    // (uniform) this launch posts to / collects from the block's outbox
    static_assert(!XW || xw_enabled<DIM, NT>(), "cross-wave sharing: DIM 3, 2..8 waves per block");
    const int lane = tid & 63, Lq = bv.Lq, wbase = tid & ~63, myw = tid >> 6;
    uint2* R = lbq + wbase;                                               // this wave's columns of the rows
    constexpr unsigned int FMAXB = 0x7F7FFFFFu, NONE = 0x7F800000u;
    unsigned long long* keys = (unsigned long long*)lbq;                  // keys[t]: row 0 of thread t (block-wide: an owner may sit in another wave)
    int* xc = (int*)(lbq + ICP_SHARE_ROWS * NT);                          // XW: nreg, ndone, idlew, -, out[8], helped[8], ...
    int* xstate = xc + XW_CTRL; int* gstate = xstate + XW_SLOTS; uint2* xa = (uint2*)(gstate + GX_SLOTS); uint2* xb = xa + XW_SLOTS; int* xp = (int*)(xb + XW_SLOTS);
    uint2* ga = (uint2*)(xp + XW_SLOTS); uint2* gb = ga + GX_SLOTS; int* gp = (int*)(gb + GX_SLOTS);      // GX: what this block posted (the owner takes it back from here)
    if (!HELP) {
        R[3 * NT + lane] = make_uint2(__float_as_uint(best), (unsigned int)bi);
        R[4 * NT + lane] = make_uint2((unsigned int)bpos, __float_as_uint(lb_others));
        R[5 * NT + lane] = make_uint2(__float_as_uint(lb3), __float_as_uint(p[0]));
        R[7 * NT + lane] = make_uint2(__float_as_uint(p[1]), __float_as_uint(p[2]));
        R[8 * NT + lane] = make_uint2(__float_as_uint(keep3[0]), __float_as_uint(keep3[1]));
        R[9 * NT + lane] = make_uint2(__float_as_uint(keep3[2]), (unsigned int)l2o);
        keys[tid] = ~0ull;
        keys[NT + tid] = ~0ull;                                            // row 1: the runner-up entry (distance, leaf)
        R[2 * NT + lane] = make_uint2(FMAXB, 0xFFFFFFFFu);                 // row 2: (bound on the rest, position of the winner)
        ((unsigned int*)(R + 6 * NT))[WAVE + lane] = FMAXB;                // row 6, second half: smallest skipped box bound
    }
    // (the fold of one lane's part of a search into the owner's record: see the comment where the loop below calls it)
    auto fold_part = [&](int owner, float wb, int wi, int wp, float b2, int l2, float b3, unsigned int mlb) {
        const unsigned long long mykey = ((unsigned long long)__float_as_uint(wb) << 32) | (unsigned int)wi;
        const unsigned long long old = __hip_atomic_fetch_min(keys + owner, mykey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned int* bp = (unsigned int*)(lbq + 2 * NT + owner);     // {rest, position}
        unsigned int rest = __float_as_uint(b3);
        unsigned long long e = ((unsigned long long)__float_as_uint(b2) << 32) | (unsigned int)l2;
        if (old < mykey) {
            const unsigned long long w = ((unsigned long long)__float_as_uint(wb) << 32) | (unsigned int)(wp >> 3);
            const unsigned long long lo = w < e ? w : e, hi = w < e ? e : w;
            if ((unsigned int)lo != (unsigned int)hi) rest = min(rest, (unsigned int)(hi >> 32));
            e = lo;
        } else if (old > mykey && old != ~0ull) rest = min(rest, (unsigned int)(old >> 32));
        {
            const unsigned long long o2 = __hip_atomic_fetch_min(keys + NT + owner, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long lo = o2 < e ? o2 : e, hi = o2 < e ? e : o2;
            if (hi != ~0ull && (unsigned int)lo != (unsigned int)hi) rest = min(rest, (unsigned int)(hi >> 32));
        }
        __hip_atomic_fetch_min(bp, rest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_min((unsigned int*)(lbq + 6 * NT + (owner & ~63)) + WAVE + (owner & 63), mlb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_load(keys + owner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == mykey) bp[1] = (unsigned int)wp;      // (XW: see the check at the end)
    };
    // ---- ONE walker in the wave (every converged launch that has a walking query at all, most waves of the late iterations): no hand-over,
    // no seed path -- the wave searches the tree for that one query level-synchronously from the root, TWO levels per dependent load.  An
    // item is (node n of level L, child j): its lane loads n and n's child j in the same trip, tests box j of n and, if that survives, the
    // four boxes in the child; the surviving grandchildren are the nodes of the next step (four items each).  A walk of Lq = 8 levels is
    // four trips and one for the leaves, against a dozen dependent loads of the general path (tools/dev_walk_trace.py: 7.3-7.9 us for the
    // one query that makes a converged launch last 19.7 instead of 14.2 us).  The frontier lives in the wave's 64-entry table: more than
    // 16 nodes on a level (64 leaves at the end) and the wave takes the general path instead -- nothing has been folded by then.  Every
    // box is tested against the seed's radius (a seeded walk is a range query: the radius is all but final), every evaluated leaf is
    // folded into the walker's record like a helper lane's, every pruned box goes to the skipped-bound minimum: the same record, the same
    // exact neighbour, bounds that are valid in the same way.
    bool lone_done = false;
    const unsigned long long wm0 = __ballot(need_walk);
    if (ICP_LONE_WALK && MODE == 1 && DIM == 3 && sizeof(MaskT) == 4 && ICP_SHARE_SPREAD && Lq > 0 && __popcll(wm0) == 1 && wm0 == __ballot(need_walk && bpos >= 0)) {
        int* tbl = (int*)(R + 6 * NT);
        const int wl = (int)__ffsll((long long)wm0) - 1;
        QueryPt<DIM> lq;
#pragma unroll
        for (int a = 0; a < DIM; a++) { const float v = __shfl(p[a], wl, WAVE); lq.p2[a].x = v; lq.p2[a].y = v; }
        const float sb = __shfl(best, wl, WAVE); const int si = __shfl(bi, wl, WAVE), sp = __shfl(bpos, wl, WAVE);
        const float thr_u = fminf(sb * ICP_PRUNE_SLACK, FLT_MAX);
        unsigned int lmlb = FMAXB;
        int nf = 1, L = 0; bool ok = true;
        if (lane == 0) tbl[0] = 0;
        while (L < Lq && ok) {
            const int items = nf * 4;
            const bool act = lane < items;
            const int n = act ? tbl[lane >> 2] : 0, j = lane & 3;
            f2 a01, a23;
            quad_lb_at<DIM>(bv, (0x55555555u & ((1u << (2 * L)) - 1u)) + (unsigned int)n, lq, a01, a23);
            const float lbj = j == 0 ? a01.x : j == 1 ? a01.y : j == 2 ? a23.x : a23.y;
            const bool alive = act && !(lbj > thr_u);
            if (act && !alive) lmlb = min(lmlb, __float_as_uint(lbj));
            const int c = 4 * n + j;
            if (L + 2 <= Lq) {                                            // (wave-uniform) the children are nodes: their boxes in the same trip
                f2 g01, g23;
                quad_lb_at<DIM>(bv, (0x55555555u & ((1u << (2 * (L + 1))) - 1u)) + (unsigned int)(act ? c : 0), lq, g01, g23);
                const bool s0 = alive && !(g01.x > thr_u), s1 = alive && !(g01.y > thr_u), s2 = alive && !(g23.x > thr_u), s3 = alive && !(g23.y > thr_u);
                if (alive) lmlb = min(min(lmlb, min(s0 ? NONE : __float_as_uint(g01.x), s1 ? NONE : __float_as_uint(g01.y))), min(s2 ? NONE : __float_as_uint(g23.x), s3 ? NONE : __float_as_uint(g23.y)));
                const unsigned long long m0 = __ballot(s0), m1 = __ballot(s1), m2 = __ballot(s2), m3 = __ballot(s3);
                const int c0 = __popcll(m0), c1 = __popcll(m1), c2 = __popcll(m2), nn = c0 + c1 + c2 + __popcll(m3);
                ok = nn <= (L + 2 < Lq ? 16 : WAVE);                       // (uniform)
                if (ok) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    if (s0) tbl[__popcll(m0 & below)] = 4 * c;
                    if (s1) tbl[c0 + __popcll(m1 & below)] = 4 * c + 1;
                    if (s2) tbl[c0 + c1 + __popcll(m2 & below)] = 4 * c + 2;
                    if (s3) tbl[c0 + c1 + c2 + __popcll(m3 & below)] = 4 * c + 3;
                }
                nf = nn; L += 2;
            } else {                                                       // the children are the leaves
                const unsigned long long m = __ballot(alive);
                if (alive) tbl[__popcll(m & ((1ull << lane) - 1ull))] = c;
                nf = __popcll(m); L += 1;
            }
        }
        if (ok) {
            if (lane < nf) {
                const int leaf = tbl[lane];
                float xb = sb, x2 = FLT_MAX, x3 = FLT_MAX; int xi = si, xp = sp, xl = -1;
                leaf_eval<DIM>(bv.leaves + leaf, leaf, lq.p2, xb, xi, xp, x2, xl, x3);
                fold_part(wbase + wl, xb, xi, xp, x2, xl, x3, lmlb);
            } else __hip_atomic_fetch_min((unsigned int*)(lbq + 6 * NT + wbase) + WAVE + wl, lmlb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // (XW: the wave has not registered as searching -- it neither offers nor needs help)
            lone_done = true;
        }
#if ICP_DEBUG_TIMES
        if (lane == 0) GX_COUNT(ok ? 11 : 12, 1);                         // development builds: lone searches completed / started over on the general path
#endif
    }
    QueryPt<DIM> qp;
    float wb; int wi, wp;
    if constexpr (DIM == 3 && MODE == 1 && sizeof(MaskT) == 4 && ICP_LONE_WALK) {
        // (read back from the rows written above rather than kept in registers across the lone walker's search: six registers of the budget)
        const uint2 ra_ = R[3 * NT + lane], rb_ = R[4 * NT + lane], rc_ = R[5 * NT + lane], rd_ = R[7 * NT + lane];
        float pq[3] = {__uint_as_float(rc_.y), __uint_as_float(rd_.x), __uint_as_float(rd_.y)};
        make_query<DIM>(bv, pq, qp);
        wb = __uint_as_float(ra_.x); wi = (int)ra_.y; wp = (int)rb_.x;
    } else {
        make_query<DIM>(bv, p, qp);
        wb = best; wi = bi; wp = bpos;
    }
    unsigned int touched = 0u;
    if (!HELP && need_walk && !lone_done) {
        if (ICP_SEED_DESCENT && wp < 0) {                                 // first iteration: a greedy descent yields a real candidate (see knn_walk)
            int idx = 0;
            for (int L = 0; L < Lq; L++) {
                f2 l01, l23;
                quad_lb_at<DIM>(bv, (0x55555555u & ((1u << (2 * L)) - 1u)) + (unsigned int)idx, qp, l01, l23);
                const float m = fminf(fminf(l01.x, l01.y), fminf(l23.x, l23.y));
                const int c = (l01.x == m) ? 0 : (l01.y == m) ? 1 : (l23.x == m) ? 2 : 3;
                idx = (idx << 2) | c;
            }
            float u2 = FLT_MAX, u3 = FLT_MAX; int ul = -1;
            leaf_eval<DIM>(bv.leaves + idx, idx, qp.p2, wb, wi, wp, u2, ul, u3);
        }
    }
    // Few walkers, all with a seed: the seed's root-to-leaf path is known, so its Lq nodes need not be visited one after the other.
    // Every level of every walker's path goes to an idle lane at once (the walker itself takes the seed's leaf): one node step in
    // parallel instead of Lq dependent ones in front of everything else.  What remains are the off-path children that survive the
    // seed's bound -- handed on below like any other parked subtree.
    const unsigned long long wm = __ballot(need_walk);
    const int W = __popcll(wm);
#if ICP_DEBUG_TIMES && ICP_DEBUG_WALK_TRACE
    const bool tracer = !HELP && !PROXY && W > 0 && W <= 3 && lane == (int)__ffsll((long long)wm) - 1;
    if (tracer) { g_walk_trace[0] = (unsigned int)W; g_walk_trace[2] = (unsigned int)wall_clock64(); }
#endif
    if (MODE == 1 && lane == 0 && !lone_done) __hip_atomic_fetch_add(xc + 0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // nreg: this wave has queries that search
    const bool spread = !HELP && !PROXY && ICP_SHARE_SPREAD && Lq > 0 && W > 0 && W * (Lq + 1) <= WAVE && wm == __ballot(need_walk && wp >= 0);
    if (ICP_PREFETCH_PATH && !spread && need_walk && wp >= 0) touched = quad_prefetch_path<DIM>(bv, wp >> 3);
    float b2 = FLT_MAX, b3 = FLT_MAX; int l2 = -1;
    unsigned int mlb = FMAXB;
    int owner = need_walk ? tid : -1;                                     // whose query this lane is searching for (thread of the block); -1: idle
    QuadStateT<MaskT> st; st.L = PROXY ? start_L : 0; st.idx = PROXY ? start_idx : 0; st.pending = 0; st.alive = need_walk;
    float thr = fminf(wb * ICP_PRUNE_SLACK, FLT_MAX);
    // one node step on the child bounds of node (st.L, st.idx): nearest surviving child next, the other survivors parked
    auto descend = [&](const f2& l01, const f2& l23) {
        const float m = fminf(fminf(l01.x, l01.y), fminf(l23.x, l23.y));
        const bool s0 = !(l01.x > thr), s1 = !(l01.y > thr), s2 = !(l23.x > thr), s3 = !(l23.y > thr);
        mlb = min(min(mlb, min(s0 ? NONE : __float_as_uint(l01.x), s1 ? NONE : __float_as_uint(l01.y))), min(s2 ? NONE : __float_as_uint(l23.x), s3 ? NONE : __float_as_uint(l23.y)));
        if (!(m > thr)) {
            const bool c0 = l01.x == m, c1 = l01.y == m, c2 = l23.x == m;
            int c = 3; c = c2 ? 2 : c; c = c1 ? 1 : c; c = c0 ? 0 : c;
            const unsigned int pend = ((s0 ? 1u : 0u) | (s1 ? 2u : 0u) | (s2 ? 4u : 0u) | (s3 ? 8u : 0u)) & ~(1u << c);
            st.pending |= (MaskT)pend << (4 * st.L);
            st.idx = (st.idx << 2) | c; st.L++;
        } else st.alive = false;
        quad_pop_bits(st);
    };
    if (spread && !lone_done) {
        int* tbl = (int*)(R + 6 * NT);
        if (need_walk) tbl[__builtin_amdgcn_mbcnt_hi((unsigned int)(wm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)wm, 0u))] = lane;
        // Two paths per walker when the lanes suffice (ICP_SPREAD_TWO): besides the path to the seed's leaf (A) the path to the leaf the
        // RUNNER-UP of the last search lives in (B; l2o on entry = the second leaf of the two-leaf tier, -1: none).  A query that searches
        // although ICP has converged sits between two (or more) targets: the subtree that holds the other one survives every bound, and a
        // helper lane would walk down to it level by level -- Lq - D dependent steps below the level D where the paths part.  Here the
        // levels of B below D go to idle lanes as well, and B's leaf to one more: roles 0 .. Lq - 1 = the nodes of A (a node both paths
        // pass through skips BOTH on-path children), Lq .. 2 Lq - 1 = the nodes of B (idle where B still runs with A), 2 Lq = leaf B.
        // Every subtree that hangs off either path is tested by exactly one lane, every on-path child by the lane of the next level.
        const bool two = ICP_SPREAD_TWO && W * (2 * Lq + 2) <= WAVE;        // (wave-uniform)
        const int roles = two ? 2 * Lq + 1 : Lq;
#if ICP_DEBUG_TIMES && ICP_DEBUG_WALK_TRACE
        if (tracer) g_walk_trace[1] = two ? 2u : 1u;
#endif
        const unsigned long long im = ~wm;
        const int ri = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(im >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)im, 0u));
        const int r = (ri * ((65536 + roles - 1) / roles)) >> 16, role = ri - r * roles;           // ri / roles, ri % roles (exact for ri < 64, roles <= 33)
        bool take = !need_walk && r < W;
        const int src = take ? tbl[r] : lane;
        float q[DIM];
#pragma unroll
        for (int a = 0; a < DIM; a++) q[a] = __shfl(qp.p2[a].x, src, WAVE);
        const float sb = __shfl(wb, src, WAVE); const int si = __shfl(wi, src, WAVE), sp = __shfl(wp, src, WAVE);
        const int sl2 = two ? __shfl(l2o, src, WAVE) : -1;
        const int leafA = sp >> 3;
        const bool onB = role >= Lq;                                          // a role of path B
        const int L = role == 2 * Lq ? Lq : (onB ? role - Lq : role);
        const int myleaf = onB ? sl2 : leafA;
        const int nodeA = leafA >> (2 * (Lq - L)), nodeB = sl2 >> (2 * (Lq - L));      // (L == Lq: the leaves themselves)
        if (onB && (sl2 < 0 || nodeB == nodeA)) take = false;                // B runs with A here (or there is no B): A's lane has the node
        if (take) {
#pragma unroll
            for (int a = 0; a < DIM; a++) { qp.p2[a].x = q[a]; qp.p2[a].y = q[a]; }
            wb = sb; wi = si; wp = sp; owner = wbase + src;
            thr = fminf(wb * ICP_PRUNE_SLACK, FLT_MAX);
            st.L = L; st.idx = myleaf >> (2 * (Lq - L)); st.alive = true;
            if (L < Lq) {
                const int skip = (myleaf >> (2 * (Lq - L - 1))) & 3;         // the child of my node that lies on my path: the next level's lane has it
                const int skipB = (!onB && sl2 >= 0 && nodeB == nodeA) ? (sl2 >> (2 * (Lq - L - 1))) & 3 : skip;      // ... and B's, where B passes through this node too
                f2 l01, l23;
                quad_lb_at<DIM>(bv, (0x55555555u & ((1u << (2 * L)) - 1u)) + (unsigned int)st.idx, qp, l01, l23);
                const float inf = __uint_as_float(NONE);
                l01.x = (skip == 0 || skipB == 0) ? inf : l01.x; l01.y = (skip == 1 || skipB == 1) ? inf : l01.y;
                l23.x = (skip == 2 || skipB == 2) ? inf : l23.x; l23.y = (skip == 3 || skipB == 3) ? inf : l23.y;
                descend(l01, l23);
            }                                                                 // (role 2 Lq: leaf B, evaluated by the loop below like any adopted leaf)
        }
        if (need_walk) { st.L = Lq; st.idx = wp >> 3; }                   // the walker itself: straight to the seed's leaf
    }
#if ICP_DEBUG_TIMES && ICP_DEBUG_WALK_TRACE
    if (tracer) g_walk_trace[3] = (unsigned int)wall_clock64();
#endif
    bool polling = false;                                                 // XW, wave-uniform: this wave is counted in idlew
    int polls = 0, trips = 0;
    for (; !lone_done;) {                                                 // (a lone walker's search is over already)
        trips++;
#if ICP_DEBUG_TIMES && ICP_DEBUG_WALK_TRACE
        if (tracer && trips < 56) g_walk_trace[7 + trips] = (unsigned int)wall_clock64();
#endif
        if (!st.alive && owner >= 0) {
            // this lane's (part of the) search is over: fold it into the owner's record.  Winner: 64-bit minimum of (distance, index).
            // Runner-up entry (distance, leaf): 64-bit minimum as well; whatever loses there -- and is not in the same leaf as what beat
            // it: such a point is bounded by that entry for as long as it stands, and by its distance in the rest once it falls -- goes to
            // the bound on the rest.  A winner of mine that loses is an entry of its own leaf; a winner I dethrone has no leaf on record:
            // straight to the rest (smaller bound than needed, for this one iteration).
            fold_part(owner, wb, wi, wp, b2, l2, b3, mlb);
            if (XW && (owner >> 6) != myw) __hip_atomic_fetch_sub(xc + 4 + (owner >> 6), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            owner = -1;
        }
        if (!__any(st.alive)) {
            if (!XW) break;
            // MODE 1: are this wave's queries complete (no part of them on the board or in another wave's lanes)?  Then it leaves at once --
            // what it can do for the others comes after its own pairs are weighed (xw_help)
            const bool complete = MODE == 1 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(xc + 4 + myw, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0;
            // otherwise the idle wave takes what is on the board (lane i looks at slot i)
            const int sv = complete ? 0 : __hip_atomic_load(xstate + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            bool mine = false;
            if (sv == 1) { int expect = 1; mine = __hip_atomic_compare_exchange_strong(xstate + lane, &expect, 3, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if (mine) {
                const uint2 ia = xa[lane], ib = xb[lane]; const int ip = xp[lane];
                __hip_atomic_store(xstate + lane, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                owner = (int)(ia.x & 0xFFFFu);
                if ((owner >> 6) == myw) __hip_atomic_fetch_sub(xc + 4 + myw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // back home
                const uint2 c = lbq[5 * NT + owner], d = lbq[7 * NT + owner];
                qp.p2[0].x = __uint_as_float(c.y); qp.p2[0].y = qp.p2[0].x; qp.p2[1].x = __uint_as_float(d.x); qp.p2[1].y = qp.p2[1].x; qp.p2[2 % DIM].x = __uint_as_float(d.y); qp.p2[2 % DIM].y = qp.p2[2 % DIM].x;
                wb = __uint_as_float(ib.x); wi = (int)ib.y; wp = ip;
                b2 = FLT_MAX; b3 = FLT_MAX; l2 = -1; mlb = FMAXB;
                thr = fminf(wb * ICP_PRUNE_SLACK, FLT_MAX);
                st.L = (int)(ia.x >> 16); st.idx = (int)ia.y; st.pending = 0; st.alive = true;
            }
            bool gmine = false;
            if (ICP_GX && XW && GXON && !complete && !__any(mine) && __builtin_amdgcn_readfirstlane(__hip_atomic_load(xc + 20, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) > 0) {
                // GX: what this block posted -- lane i looks at slot i.  A result that has come back is folded like the result of a lane of
                // this wave (the lane "finishes" that part right here); a slot nobody claimed is taken back and searched here.
                if (lane < GX_SLOTS && __hip_atomic_load(gstate + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 1) {
                    unsigned long long* sl = gx->slots + ((size_t)gx_block * GX_SLOTS + lane) * GX_GRANULES;
                    const unsigned long long r3 = gx_load(sl + 8), cl = gx_load(sl + 4);
                    int expect = 1;
                    if (r3 != GX_EMPTY) {
                        if (__hip_atomic_compare_exchange_strong(gstate + lane, &expect, 2, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            // (the four result granules are written together: the others are a moment away at most; one at a time -- registers)
                            auto wait_granule = [&](const unsigned long long* g) { unsigned long long v = gx_load(g); for (int sp = 0; v == GX_EMPTY && sp < (1 << 18); sp++) v = gx_load(g); if (v == GX_EMPTY && fault) atomicOr(fault, 1); return v; };
                            owner = (int)(ga[lane].x & 0xFFFFu);
                            wp = (int)(unsigned int)r3;
                            { const unsigned long long r0 = wait_granule(sl + 5); wi = (int)(unsigned int)r0; wb = __uint_as_float((unsigned int)(r0 >> 32)); }
                            { const unsigned long long r1 = wait_granule(sl + 6); l2 = (int)(unsigned int)r1; b2 = __uint_as_float((unsigned int)(r1 >> 32)); }
                            { const unsigned long long r2 = wait_granule(sl + 7); b3 = __uint_as_float((unsigned int)r2); mlb = (unsigned int)(r2 >> 32); }
                            st.alive = false; st.pending = 0;
                            gmine = true; GX_COUNT(3, 1);
                        }
                    } else if (cl == 1ull) {
                        if (__hip_atomic_compare_exchange_strong(gstate + lane, &expect, 2, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            unsigned long long ce = 1ull;
                            if (__hip_atomic_compare_exchange_strong(sl + 4, &ce, 3ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                const uint2 ia = ga[lane], ib = gb[lane];
                                owner = (int)(ia.x & 0xFFFFu);
                                const uint2 c = lbq[5 * NT + owner], d = lbq[7 * NT + owner];
                                qp.p2[0].x = __uint_as_float(c.y); qp.p2[0].y = qp.p2[0].x; qp.p2[1].x = __uint_as_float(d.x); qp.p2[1].y = qp.p2[1].x; qp.p2[2 % DIM].x = __uint_as_float(d.y); qp.p2[2 % DIM].y = qp.p2[2 % DIM].x;
                                wb = __uint_as_float(ib.x); wi = (int)ib.y; wp = gp[lane];
                                b2 = FLT_MAX; b3 = FLT_MAX; l2 = -1; mlb = FMAXB;
                                thr = fminf(wb * ICP_PRUNE_SLACK, FLT_MAX);
                                st.L = (int)(ia.x >> 16); st.idx = (int)ia.y; st.pending = 0; st.alive = true;
                                gmine = true; GX_COUNT(2, 1);
                            } else __hip_atomic_store(gstate + lane, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // a helper was faster: its result will come
                        }
                    }
                    if (gmine) {
                        if ((owner >> 6) == myw) __hip_atomic_fetch_sub(xc + 4 + myw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // back home
                        gx_store(sl + 0, GX_EMPTY); gx_store(sl + 1, GX_EMPTY); gx_store(sl + 2, GX_EMPTY); gx_store(sl + 3, GX_EMPTY); gx_store(sl + 4, 0ull);      // the slot as it was found
                        gx_store(sl + 5, GX_EMPTY); gx_store(sl + 6, GX_EMPTY); gx_store(sl + 7, GX_EMPTY); gx_store(sl + 8, GX_EMPTY);
                    }
                }
            }
            if (__any(mine) || __any(gmine)) {
                if (polling) { polling = false; if (lane == 0) __hip_atomic_fetch_sub(xc + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            } else {
                if (MODE == 1) {
                    if (complete) { if (lane == 0) __hip_atomic_fetch_add(xc + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }      // ndone
                } else {
                    const int ndone = __builtin_amdgcn_readfirstlane(__hip_atomic_load(xc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)), nreg = __builtin_amdgcn_readfirstlane(__hip_atomic_load(xc + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (ndone >= nreg) break;                             // every wave that searched is complete
                }
                if (!polling) { polling = true; if (lane == 0) __hip_atomic_fetch_add(xc + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                if (++polls > XW_SPIN_LIMIT) { if (fault && lane == 0) atomicOr(fault, 1); break; }      // (cannot happen: every part is held by a running lane or lies on the board)
                __builtin_amdgcn_s_sleep(ICP_XW_SLEEP);
                continue;
            }
        }
        bool served = true;                                               // wave-uniform: no idle lane is left without work
#pragma unroll 1
        for (int round = 0; round < ICP_SHARE_ROUNDS; round++) {
            const unsigned long long im = __ballot(owner < 0);
            if (!im) break;
            const bool can = st.alive && st.pending != 0;
            const unsigned long long dm = __ballot(can);
            served = __popcll(im) <= __popcll(dm);
            if (!dm) break;
            {
                const int n = min(__popcll(im), __popcll(dm));
                int* tbl = (int*)(R + 6 * NT);
                int dL = 0, dIdx = 0;
                const int rd = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)dm, 0u));      // donors below me
                if (can && rd < n) {
                    const int low = sizeof(MaskT) == 8 ? __ffsll((long long)st.pending) - 1 : __ffs((int)st.pending) - 1;            // shallowest parked child: the largest subtree
                    const int lv = low >> 2;
                    st.pending &= ~((MaskT)1 << low);
                    dL = lv + 1; dIdx = ((st.idx >> (2 * (st.L - lv))) << 2) | (low & 3);
                    tbl[rd] = lane;
                }
                const int ri = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(im >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)im, 0u));
                const bool take = owner < 0 && ri < n;
                const int src = take ? tbl[ri] : lane;
                float q[DIM];
#pragma unroll
                for (int a = 0; a < DIM; a++) q[a] = __shfl(qp.p2[a].x, src, WAVE);
                const float sb = __shfl(wb, src, WAVE); const int si = __shfl(wi, src, WAVE), sp = __shfl(wp, src, WAVE), so = __shfl(owner, src, WAVE);
                const int sL = __shfl(dL, src, WAVE), sI = __shfl(dIdx, src, WAVE);
                if (take) {
#pragma unroll
                    for (int a = 0; a < DIM; a++) { qp.p2[a].x = q[a]; qp.p2[a].y = q[a]; }
                    wb = sb; wi = si; wp = sp; owner = so;
                    b2 = FLT_MAX; b3 = FLT_MAX; l2 = -1; mlb = FMAXB;
                    thr = fminf(wb * ICP_PRUNE_SLACK, FLT_MAX);
                    st.L = sL; st.idx = sI; st.pending = 0; st.alive = true;
                    if (XW && (so >> 6) != myw) __hip_atomic_fetch_add(xc + 4 + (so >> 6), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // one more part of that wave's query in foreign hands
                }
            }
        }
        const int idlew = (XW && served) ? __builtin_amdgcn_readfirstlane(__hip_atomic_load(xc + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) : 0;
        if (ICP_GX && XW && GXON) {
            if (trips == (gx->start_trips >> 1) && lane == 0) __hip_atomic_store(xc + 21, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // this block walks for long: its wave 0 will look for work elsewhere afterwards
            if (served && idlew == 0 && trips >= gx->start_trips && (trips & 7) == 0) {
                // GX: nobody in this block is idle and this wave has been at it for long: its lanes post their shallowest parked subtree in
                // the block's outbox, for a block that is through with everything
                const bool can = st.alive && st.pending != 0;
                const unsigned long long dm = __ballot(can);
                if (dm) {
                    int base = 0;
                    if (lane == 0) base = __hip_atomic_fetch_add(xc + 20, __popcll(dm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base < GX_SLOTS) {
                        const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)dm, 0u));
                        if (can && slot < GX_SLOTS) {
                            const int low = sizeof(MaskT) == 8 ? __ffsll((long long)st.pending) - 1 : __ffs((int)st.pending) - 1;
                            const int lv = low >> 2;
                            st.pending &= ~((MaskT)1 << low);
                            const unsigned int w0 = (unsigned int)owner | ((unsigned int)(lv + 1) << 16), w1 = (unsigned int)(((st.idx >> (2 * (st.L - lv))) << 2) | (low & 3));
                            ga[slot] = make_uint2(w0, w1); gb[slot] = make_uint2(__float_as_uint(wb), (unsigned int)wi); gp[slot] = wp;
                            __hip_atomic_fetch_add(xc + 4 + (owner >> 6), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(xc + 12 + (owner >> 6), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(gstate + slot, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            unsigned long long* sl = gx->slots + ((size_t)gx_block * GX_SLOTS + slot) * GX_GRANULES;
                            gx_store(sl + 0, gx_pack(w0, w1)); gx_store(sl + 1, gx_pack(__float_as_uint(wb), (unsigned int)wi));
                            gx_store(sl + 2, gx_pack((unsigned int)wp, __float_as_uint(qp.p2[0].x))); gx_store(sl + 3, gx_pack(__float_as_uint(qp.p2[1].x), __float_as_uint(qp.p2[2 % DIM].x)));
                            gx_store(sl + 4, 1ull);
                            GX_COUNT(0, 1);
                        }
                        if (lane == 0) atomicMax(gx->hdr + 2 * gx_block, (unsigned int)min(base + __popcll(dm), GX_SLOTS));
                    }
                }
            }
        }
        if (XW && served && idlew > 0) {
            // another wave of the block waits for work and this wave's own lanes are all busy: lanes that still have parked subtrees put
            // their shallowest one on the board
            const bool can = st.alive && st.pending != 0;
            const unsigned long long dm = __ballot(can);
            if (dm) {
                const int sv = __hip_atomic_load(xstate + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned long long fm = __ballot(sv == 0);
                const int n = min(__popcll(dm), __popcll(fm));
                int* tbl = (int*)(R + 6 * NT);
                const int rf = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)fm, 0u));
                if (sv == 0 && rf < n) tbl[rf] = lane;                    // the rf-th free slot
                const int rd = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)dm, 0u));
                if (can && rd < n) {
                    const int slot = tbl[rd];
                    int expect = 0;
                    if (__hip_atomic_compare_exchange_strong(xstate + slot, &expect, 2, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        const int low = sizeof(MaskT) == 8 ? __ffsll((long long)st.pending) - 1 : __ffs((int)st.pending) - 1;
                        const int lv = low >> 2;
                        st.pending &= ~((MaskT)1 << low);
                        xa[slot] = make_uint2((unsigned int)owner | ((unsigned int)(lv + 1) << 16), (unsigned int)(((st.idx >> (2 * (st.L - lv))) << 2) | (low & 3)));
                        xb[slot] = make_uint2(__float_as_uint(wb), (unsigned int)wi);
                        xp[slot] = wp;
                        __hip_atomic_fetch_add(xc + 4 + (owner >> 6), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(xc + 12 + (owner >> 6), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // that wave's queries had outside help
                        __hip_atomic_store(xstate + slot, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        while (st.alive && st.L < Lq) {
            f2 l01, l23;
            quad_lb_at<DIM>(bv, (0x55555555u & ((1u << (2 * st.L)) - 1u)) + (unsigned int)st.idx, qp, l01, l23);
            descend(l01, l23);
            // Back to the hand-over as soon as it has something to do: while every idle lane found work last time (lanes are what is
            // scarce) when a lane runs out of work; otherwise (lanes idle, parked subtrees scarce) when a lane parks one.  Measured against
            // handing over only between leaves: iteration 0 0.142 -> 0.126 ms, iterations 1-9 0.065 -> 0.061, 10-16 0.033 -> 0.028.
            if (served ? __any(!st.alive) : __any(st.pending != 0)) break;
        }
        if (st.alive && st.L == Lq) {                                      // (a lane that left the loop above early is still at a node)
            leaf_eval<DIM>(bv.leaves + st.idx, st.idx, qp.p2, wb, wi, wp, b2, l2, b3);
            thr = fminf(wb * ICP_PRUNE_SLACK, FLT_MAX);
            st.alive = false;
            quad_pop_bits(st);
        }
    }
    if (XW && polling && lane == 0) __hip_atomic_fetch_sub(xc + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if ICP_DEBUG_TIMES && ICP_DEBUG_WALK_TRACE
    if (tracer) { g_walk_trace[4] = (unsigned int)trips; g_walk_trace[5] = (unsigned int)polls; g_walk_trace[6] = (unsigned int)wall_clock64(); }
#endif
    asm volatile("" ::"v"(touched));
    if (HELP) return;
    if (PROXY) {                                                          // raw: what the owner folds like the result of one of its own lanes
        const unsigned long long key = keys[tid], key2 = keys[NT + tid];
        const uint2 b = R[2 * NT + lane];
        best = __uint_as_float((unsigned int)(key >> 32)); bi = (int)(unsigned int)key; bpos = (int)b.y;
        lb_others = key2 == ~0ull ? FLT_MAX : __uint_as_float((unsigned int)(key2 >> 32)); l2o = key2 == ~0ull ? -1 : (int)(unsigned int)key2;
        lb3 = __uint_as_float(b.x);
        keep3[0] = __uint_as_float(((const unsigned int*)(R + 6 * NT))[WAVE + lane]);
        return;
    }
    {
        const uint2 c = R[5 * NT + lane], d = R[7 * NT + lane], e = R[8 * NT + lane], f = R[9 * NT + lane];
        p[0] = __uint_as_float(c.y); p[1] = __uint_as_float(d.x); p[2] = __uint_as_float(d.y);
        keep3[0] = __uint_as_float(e.x); keep3[1] = __uint_as_float(e.y); keep3[2] = __uint_as_float(f.x);
    }
    if (need_walk) {
        const unsigned long long key = keys[tid], key2 = keys[NT + tid];
        const uint2 b = R[2 * NT + lane];
        best = __uint_as_float((unsigned int)(key >> 32)); bi = (int)(unsigned int)key; bpos = (int)b.y;
        const float sk = __uint_as_float(((const unsigned int*)(R + 6 * NT))[WAVE + lane]);
        const float rest = fminf(__uint_as_float(b.x), sk);
        // (a winner dethroned during the merge went straight to the rest: it can be nearer than the runner-up entry.  No entry at all: the
        //  initial key reads as a NaN distance and fminf returns the other operand)
        lb_others = sqrt_dn(fminf(__uint_as_float((unsigned int)(key2 >> 32)), rest));
        lb3 = sqrt_dn(rest);
        l2o = key2 == ~0ull ? -1 : (int)(unsigned int)key2;
        if (XW && __builtin_amdgcn_readfirstlane(__hip_atomic_load(xc + 12 + myw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
            // lanes of other waves folded into this record: the position written beside the key may belong to a fold that lost
            if (bpos < 0 || bv.recs[bpos].idx != bi) bpos = bv.pos_of[bi];
        }
    } else {
        const uint2 a = R[3 * NT + lane], b = R[4 * NT + lane], c = R[5 * NT + lane];
        best = __uint_as_float(a.x); bi = (int)a.y; bpos = (int)b.x; lb_others = __uint_as_float(b.y); lb3 = __uint_as_float(c.x); l2o = (int)R[9 * NT + lane].y;
    }
}


// XW, a wave that has nothing of its own left to do (its pairs are weighed, only the block's sums remain): help the block's other waves
// until every wave that searched is complete.  The wave's rows 3.. are not touched (the caller parks its pair there).
template <int DIM, int NT, class MaskT>
__device__ __forceinline__ void xw_help(const BvhViewT<DIM>& bv, uint2* __restrict__ lbq, int tid, int* fault, const GxParams* gx, int gx_block) {
    float p[DIM], k3[3] = {0.f, 0.f, 0.f}, best = FLT_MAX, lbo = 0.f, lb3 = 0.f; int bi = -1, bpos = -1, l2 = -1;
#pragma unroll
    for (int q = 0; q < DIM; q++) p[q] = 0.f;
    knn_walk_shared<DIM, NT, MaskT, 2>(bv, p, k3, false, best, bi, bpos, lbo, lb3, l2, lbq, tid, fault, gx, gx_block);
}

// GX: wave 0 of a block that is through with everything looks into the outboxes of other blocks (64 of them per round, one per lane),
// claims a posted subtree, searches it with the whole wave and writes the result back; it leaves after gx.empty_rounds rounds in a row
// without a claim.
constexpr int GX_MAX_ROUNDS = 1 << 14;
template <int DIM, int NT, class MaskT>
__device__ __forceinline__ void gx_help(const BvhViewT<DIM>& bv, const GxParams& gx, int my_block, int nblocks, uint2* __restrict__ lbq, int tid) {
    const int lane = tid & 63;
    int empty = 0;
    if (lane == 0) GX_COUNT(4, 1);
    for (int r = 0; r < GX_MAX_ROUNDS && empty < gx.empty_rounds; r++) {
        if (lane == 0) GX_COUNT(5, 1);
        const unsigned int v = ((unsigned int)my_block * 97u + (unsigned int)r * 64u + (unsigned int)lane + 1u) % (unsigned int)nblocks;
        bool got = false;
        unsigned long long* sl = nullptr;
        if ((int)v != my_block) {
            const unsigned long long h = gx_load((const unsigned long long*)gx.hdr + v);      // (posted, lowest slot that may be unclaimed)
            const unsigned int posted = (unsigned int)h, hint = (unsigned int)(h >> 32);
            if (hint < posted && hint < (unsigned int)GX_SLOTS) {
                sl = gx.slots + ((size_t)v * GX_SLOTS + hint) * GX_GRANULES;
                const unsigned long long cl = gx_load(sl + 4);
                unsigned long long ce = 1ull;
                if (cl == 1ull && __hip_atomic_compare_exchange_strong(sl + 4, &ce, 2ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) got = true;
                if (got || cl >= 2ull) atomicMax(gx.hdr + 2 * v + 1, hint + 1u);      // (claimed by somebody: the next prober starts one further)
            }
        }
        if (!__any(got)) { empty++; __builtin_amdgcn_s_sleep(16); continue; }
        empty = 0;
        if (got) GX_COUNT(1, 1);
        if (lane == 0) GX_COUNT(6, 1);
        float p[DIM], k3[3] = {0.f, 0.f, 0.f}, best = FLT_MAX, lbo = 0.f, lb3 = 0.f; int bi = -1, bpos = -1, l2 = -1, L = 0, idx = 0;
#pragma unroll
        for (int q = 0; q < DIM; q++) p[q] = 0.f;
        if (got) {
            unsigned long long g0 = GX_EMPTY, g1 = GX_EMPTY, g2 = GX_EMPTY, g3 = GX_EMPTY;
            for (int sp = 0; sp < (1 << 18); sp++) {                      // (item and claim word were written together)
                g0 = gx_load(sl + 0); g1 = gx_load(sl + 1); g2 = gx_load(sl + 2); g3 = gx_load(sl + 3);
                if (g0 != GX_EMPTY && g1 != GX_EMPTY && g2 != GX_EMPTY && g3 != GX_EMPTY) break;
            }
            L = (int)(((unsigned int)g0 >> 16) & 0xFFu); idx = (int)(unsigned int)(g0 >> 32);
            best = __uint_as_float((unsigned int)g1); bi = (int)(unsigned int)(g1 >> 32);
            bpos = (int)(unsigned int)g2; p[0] = __uint_as_float((unsigned int)(g2 >> 32)); p[1] = __uint_as_float((unsigned int)g3); p[2 % DIM] = __uint_as_float((unsigned int)(g3 >> 32));
            if (g0 == GX_EMPTY || g1 == GX_EMPTY || g2 == GX_EMPTY || g3 == GX_EMPTY || L < 1 || L > bv.Lq || bpos < 0) got = false;      // (never seen; an unusable item stays claimed and the owner's bounded wait reports it)
        }
        knn_walk_shared<DIM, NT, MaskT, 3>(bv, p, k3, got, best, bi, bpos, lbo, lb3, l2, lbq, tid, nullptr, nullptr, 0, L, idx);
        if (got) {
            gx_store(sl + 5, gx_pack((unsigned int)bi, __float_as_uint(best))); gx_store(sl + 6, gx_pack((unsigned int)l2, __float_as_uint(lbo)));
            gx_store(sl + 7, gx_pack(__float_as_uint(lb3), __float_as_uint(k3[0]))); gx_store(sl + 8, gx_pack((unsigned int)bpos, 1u));
        }
    }
}

__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, WAVE));
    return v;
}

// The tree walk proper for query p, starting from the seed (best, bi, bpos); returns the lower bound on the distance to every
// target other than the winner.  NT = threads of the block (layout of the LDS stacks).
template <int DIM, int NT>
__device__ __forceinline__ float knn_walk(const BvhViewT<DIM>& bv, const float* p, float& best, int& bi, int& bpos, float& lb3, int& l2, uint2* __restrict__ lbq, int tid, int* dbg_out = nullptr) {
    int dbg_nodes = 0, dbg_leaves = 0;
    QueryPt<DIM> qp;
    make_query<DIM>(bv, p, qp);
    float b2 = FLT_MAX, b3 = FLT_MAX, minlb = FLT_MAX; l2 = -1;
    unsigned int touched = 0u;
    if (ICP_SEED_DESCENT && bpos < 0 && bv.Lq > 0) {
        // No candidate yet (first iteration): one greedy root-to-leaf descent -- nearest child at every level, nothing parked --
        // yields a real candidate first.  The proper walk below then prunes from the root on; without it every level parks
        // three siblings that are popped and discarded later, which costs more instructions than these Lq extra node visits.
        int idx = 0;
        for (int L = 0; L < bv.Lq; L++) {
            f2 l01, l23;
            quad_lb_at<DIM>(bv, (0x55555555u & ((1u << (2 * L)) - 1u)) + (unsigned int)idx, qp, l01, l23);
            const float m = fminf(fminf(l01.x, l01.y), fminf(l23.x, l23.y));
            const int c = (l01.x == m) ? 0 : (l01.y == m) ? 1 : (l23.x == m) ? 2 : 3;
            idx = (idx << 2) | c;
        }
        float u2 = FLT_MAX, u3 = FLT_MAX; int ul = -1;
        leaf_eval<DIM>(bv.leaves + idx, idx, qp.p2, best, bi, bpos, u2, ul, u3);      // (the walk re-evaluates this leaf: the bound bookkeeping stays in one place)
    }
    if (ICP_PREFETCH_PATH && bpos >= 0) touched = quad_prefetch_path<DIM>(bv, bpos >> 3);
    if (bv.Lq <= 8) {                                     // uniform: up to 8 levels (524 288 targets) the pending bits fit 32 bits
        QuadStateT<unsigned int> st; st.L = 0; st.idx = 0; st.pending = 0u; st.alive = true;
        quad_run<DIM>(bv, qp, st, best, bi, bpos, b2, l2, b3, minlb, dbg_nodes, dbg_leaves);
    } else {
        QuadStateT<unsigned long long> st; st.L = 0; st.idx = 0; st.pending = 0ull; st.alive = true;
        quad_run<DIM>(bv, qp, st, best, bi, bpos, b2, l2, b3, minlb, dbg_nodes, dbg_leaves);
    }
#if ICP_DEBUG_STEPS
    if (dbg_out) *dbg_out = dbg_nodes | (dbg_leaves << 16);
#endif
    asm volatile("" ::"v"(touched));
    lb3 = sqrt_dn(fminf(b3, minlb));
    return sqrt_dn(fminf(b2, minlb));
}

// One query per lane (k < 0: none), the whole wave together: the lanes without a walk of their own help with the others'.
template <int DIM>
__device__ __forceinline__ void knn_bvh_query(const KnnParams& kp, const BvhViewT<DIM>& bv, int k, uint2* __restrict__ lbq, int tid,
                                              float& best, int& bi, int& bpos) {
    float p[DIM];
#pragma unroll
    for (int q = 0; q < DIM; q++) p[q] = 0.f;
    if (k >= 0) knn_load_query<DIM>(kp, k, p);
    best = FLT_MAX; bi = -1; bpos = -1;
    float lb_others = 0.f, lb3 = 0.f; int l2 = -1;  // lower bounds on the (real) distance from p to every target except bi / outside bi's leaf and leaf l2
    bool need_walk = false;
    if (k >= 0 && finite3(p[0], p[1], p[2]) && bv.n_valid > 0) {
        // (a query verified here re-anchors: lb_others is already L - delta, and it bounds everything outside the neighbour's leaf too)
        if (knn_try_verify<DIM>(kp, bv, k, p, best, bi, bpos, lb_others)) lb3 = lb_others;
        else need_walk = true;
    }
#if ICP_SHARE_WALKS
    if (__any(need_walk)) {
        float none[3] = {0.f, 0.f, 0.f};
        if (bv.Lq <= 8) knn_walk_shared<DIM, BVH_THREADS, unsigned int>(bv, p, none, need_walk, best, bi, bpos, lb_others, lb3, l2, lbq, tid);
        else knn_walk_shared<DIM, BVH_THREADS, unsigned long long>(bv, p, none, need_walk, best, bi, bpos, lb_others, lb3, l2, lbq, tid);
    }
#else
    if (need_walk) lb_others = knn_walk<DIM, BVH_THREADS>(bv, p, best, bi, bpos, lb3, l2, lbq, tid);
#endif
    if (k >= 0) knn_store_state<DIM>(kp, k, p, best, bpos, lb_others, lb3, l2);
}

// Which query does this lane serve?  Position t of the (Morton-sorted) query order, in XCD-contiguous slices.
__device__ __forceinline__ int knn_bvh_lane_query(const KnnParams& kp, const int* __restrict__ qorder, int tid) {
    const int t = xcd_contiguous_block(blockIdx.x, gridDim.x) * BVH_THREADS + tid;
    if (t >= kp.n) return -1;
    return qorder ? qorder[t] : t;                        // spatially sorted queries: neighbouring lanes walk similar paths
}

template <int DIM>
__global__ __launch_bounds__(BVH_THREADS) void k_knn_bvh(const KnnParams kp, const BvhViewT<DIM> bv, const int* __restrict__ qorder) {
    extern __shared__ uint2 bvh_lbq[];                    // [ICP_SHARE_ROWS][BVH_THREADS]: the shared walk's records
    const int tid = threadIdx.x;
    const int k = knn_bvh_lane_query(kp, qorder, tid);
    float best; int bi, bpos;
    knn_bvh_query<DIM>(kp, bv, k, bvh_lbq, tid, best, bi, bpos);
    if (k < 0) return;
    icp_match_t m;
    if (best <= kp.max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }
    kp.out[k] = m;
}


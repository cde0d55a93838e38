// dev_common.hpp -- shared constants, pose state, transforms, upload conversion kernels.
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;
constexpr int KNN_CH = 16;          // targets per filter chunk (one s_load_dwordx16 per coordinate)
constexpr int NSUM = 40;            // doubles per block partial (row stride of the partial / total arrays)
constexpr int NSUM_USED = 34;       // rows the producers write and the reducer folds: count, 3 + 3 sums, 27 metric sums
constexpr int SUM_N = 0, SUM_S = 1, SUM_D = 4, SUM_M = 7;   // count, sum s, sum d, metric-specific block
constexpr int POST_THREADS = 256;

// Device-resident pose: column-major 4x4 (Eigen layout) + row-major (R^-1)^T for the normals.
struct PoseState {
    float pose[16];
    float nmat[9];
    float mean_s[3];       // unweighted means of the current valid correspondences (symmetric ICP)
    float mean_d[3];
    int fault;             // set by the device when a bounded wait ran out (k_reduce_solve's hand-over): the run reports ICP_ERR_HIP
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

// A resident source cloud becomes the target of the next pair (consecutive scan pairs share a scan: icp_batch_run): plane by plane, with the
// target's +inf padding.  src / dst: 6 planes (x y z nx ny nz), grid.y = plane.
struct Planes6 { const float* s[6]; float* d[6]; };
__global__ void k_copy_planes_pad(const Planes6 pl, int n, int pad_to) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    const float* __restrict__ s = nullptr; float* __restrict__ d = nullptr;
#pragma unroll
    for (int q = 0; q < 6; q++) if (q == k) { s = pl.s[q]; d = pl.d[q]; }
    if (!s || !d) return;
    if (i < n) d[i] = s[i];
    else if (i < pad_to && k < 3) d[i] = INFINITY;
}
__global__ void k_fill_u64(unsigned long long* p, int n, unsigned long long v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// Set-up passes over a freshly uploaded cloud, on the device (round 1 looped over every point on the host for these):
//   k_mark_finite : flag[i] = point i finite (&& normal finite, when normals are given)  -- the finite filter of the index build
//                   (non-finite targets can never win the argmin) and PointCloud::getCoarseResolution's validity test (PointCloud.h:334)
//   k_bbox        : bounding box of the finite points as ordered-bit unsigned min / max (one atomic per block and bound)
//   k_stride_flags: flag of every `factor`-th point, for the multi-resolution selections
__global__ void k_mark_finite(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                              const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz, int n, uint8_t* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool ok = finite3(x[i], y[i], z[i]);
    if (ok && nx) ok = finite3(nx[i], ny[i], nz[i]);
    flag[i] = ok ? 1 : 0;
}
__device__ __forceinline__ unsigned int ordered_bits_u(float f) { const unsigned int u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float from_ordered_bits_u(unsigned int u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
__global__ __launch_bounds__(256) void k_bbox(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n, unsigned int* __restrict__ box /* [6]: min xyz, max xyz; preset to ~0 / 0 */) {
    __shared__ unsigned int sm[4][6];
    unsigned int lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float a = x[i], b = y[i], c = z[i];
        if (finite3(a, b, c)) {
            const unsigned int o[3] = {ordered_bits_u(a), ordered_bits_u(b), ordered_bits_u(c)};
#pragma unroll
            for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], o[k]); hi[k] = max(hi[k], o[k]); }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], (unsigned int)__shfl_down((int)lo[k], off, WAVE)); hi[k] = max(hi[k], (unsigned int)__shfl_down((int)hi[k], off, WAVE)); }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { for (int k = 0; k < 3; k++) { sm[w][k] = lo[k]; sm[w][3 + k] = hi[k]; } }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        unsigned int v = sm[0][k];
        for (int ww = 1; ww < 4; ww++) v = k < 3 ? min(v, sm[ww][k]) : max(v, sm[ww][k]);
        if (k < 3) atomicMin(box + k, v); else atomicMax(box + k, v);
    }
}
__global__ void k_stride_flags(const uint8_t* __restrict__ flag, int n, int factor, int count, uint8_t* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) out[j] = flag[(size_t)j * factor];
}
struct MulBy { int f; __host__ __device__ int operator()(int j) const { return j * f; } };

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

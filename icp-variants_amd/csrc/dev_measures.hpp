// dev_measures.hpp -- RMSE / benchmark error, selection, back-projection, transforms.
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.
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

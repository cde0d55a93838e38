// launch-floor micro-benchmark: back-to-back dependent kernels in one stream, timed with events over many launches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty(int* p) { if (p == nullptr) p[0] = 1; }
__global__ void k_trip1(const float* a, float* out, int n) { int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) { float v = a[t]; if (v == 123.f) out[t] = v; } }
__global__ void k_trip2(const int* idx, const float* a, float* out, int n) { int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) { float v = a[idx[t]]; if (v == 123.f) out[t] = v; } }
__global__ void k_trip3(const int* idx, const int* idx2, const float* a, float* out, int n) { int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) { float v = a[idx2[idx[t]]]; if (v == 123.f) out[t] = v; } }
__global__ void k_store(float* out, int n) { int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) out[t] = 1.f; }
__global__ void k_partial(double* out, int nb) { __shared__ double s[2]; if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = 1.0; __syncthreads(); if (threadIdx.x < 34) out[threadIdx.x * nb + blockIdx.x] = s[0] + s[1]; }
template <class F> float timeit(hipStream_t st, int reps, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f();
    hipStreamSynchronize(st);
    hipEventRecord(a, st); for (int i = 0; i < reps; i++) f(); hipEventRecord(b, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}
int main() {
    const int n = 370488, nb = (n + 127) / 128;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float *a, *out; int *idx, *idx2; double* part;
    hipMalloc(&a, n * 4); hipMalloc(&out, n * 4); hipMalloc(&idx, n * 4); hipMalloc(&idx2, n * 4); hipMalloc(&part, (size_t)nb * 34 * 8);
    std::vector<int> h(n); for (int i = 0; i < n; i++) h[i] = (int)(((long long)i * 7919) % n);
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(idx2, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(a, 0, n * 4);
    printf("empty 1 block      %.2f us\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (int*)out); }));
    printf("empty %d blocks  %.2f us\n", nb, timeit(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(nb), dim3(128), 0, st, (int*)out); }));
    printf("empty %d blocks + 10 KB LDS  %.2f us\n", nb, timeit(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(nb), dim3(128), 10240, st, (int*)out); }));
    printf("1 trip              %.2f us\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_trip1, dim3(nb), dim3(128), 0, st, a, out, n); }));
    printf("2 trips             %.2f us\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_trip2, dim3(nb), dim3(128), 0, st, idx, a, out, n); }));
    printf("3 trips             %.2f us\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_trip3, dim3(nb), dim3(128), 0, st, idx, idx2, a, out, n); }));
    printf("store 4 B/lane      %.2f us\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_store, dim3(nb), dim3(128), 0, st, out, n); }));
    printf("partials 34/block   %.2f us\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_partial, dim3(nb), dim3(128), 0, st, part, nb); }));
    printf("pair: 3 trips + partials, then 34-block empty   %.2f us\n", timeit(st, 2000, [&] { hipLaunchKernelGGL(k_trip3, dim3(nb), dim3(128), 0, st, idx, idx2, a, out, n); hipLaunchKernelGGL(k_empty, dim3(34), dim3(256), 0, st, (int*)out); }));
    return 0;
}

// does a long straight-line kernel pay for cold instruction fetches at every launch?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Chain { static __device__ __forceinline__ float run(float v) { v = Chain<N - 1>::run(v); return v * (1.0f + N * 1e-7f) + (float)N * 1.25e-3f; } };
template <> struct Chain<0> { static __device__ __forceinline__ float run(float v) { return v; } };
template <int N> __global__ void k_line(const float* a, float* out, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= n) return;
    float v = Chain<N>::run(a[t]); if (v == 123.f) out[t] = v;
}
template <int N> __global__ void k_loop(const float* a, float* out, int n) {   // the same arithmetic in a loop: small code
    int t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= n) return;
    float v = a[t];
#pragma unroll 1
    for (int i = 1; i <= N; i++) v = v * (1.0f + i * 1e-7f) + (float)i * 1.25e-3f;
    if (v == 123.f) out[t] = v;
}
template <class F> float timeit(hipStream_t st, int reps, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f();
    hipStreamSynchronize(st);
    hipEventRecord(a, st); for (int i = 0; i < reps; i++) f(); hipEventRecord(b, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}
__global__ void k_small(float* out) { if (out == nullptr) out[0] = 1.f; }
int main() {
    const int n = 370488, nb = (n + 127) / 128;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float *a, *out; hipMalloc(&a, n * 4); hipMalloc(&out, n * 4); hipMemset(a, 0, n * 4);
#define T(N) printf("straight %5d fma: %.2f us   loop: %.2f us   straight alternating with a small kernel: %.2f us\n", N, \
      timeit(st, 1000, [&] { hipLaunchKernelGGL(k_line<N>, dim3(nb), dim3(128), 0, st, a, out, n); }), \
      timeit(st, 1000, [&] { hipLaunchKernelGGL(k_loop<N>, dim3(nb), dim3(128), 0, st, a, out, n); }), \
      timeit(st, 1000, [&] { hipLaunchKernelGGL(k_line<N>, dim3(nb), dim3(128), 0, st, a, out, n); hipLaunchKernelGGL(k_small, dim3(34), dim3(256), 0, st, out); }));
    T(250) T(500) T(1000) T(2000) T(4000)
    return 0;
}

// Does hipExtAnyOrderLaunch let the next launch on a stream start while the one before it still runs (gfx950, ROCm 7.2)?  And what
// does a chain of launches cost per link when the dependency between them is carried by memory instead of by the launch boundary?
//   test 1: A (one block) waits -- bounded -- for a flag that only B sets; B is enqueued behind A on the same stream.
//           in-order: A runs out of polls (B cannot start).  any-order: A sees the flag.
//   test 2: a chain shaped like the merged ICP loop: block 0 of launch i waits until all worker blocks of launch i - 1 have signed off,
//           then publishes "go i"; the worker blocks of launch i wait for "go i", work for ~W us, sign off.  Period per launch with the
//           launch boundary as the dependency (in-order; the waits are then satisfied at once) against any-order launches.
// Every wait is bounded: a waiter that runs out of polls raises fail[0] and leaves.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait(unsigned long long* flag, int* res, int limit) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    int ok = 0;
    for (int s = 0; s < limit; s++) { if (ld(flag) == 1ull) { ok = 1; break; } __builtin_amdgcn_s_sleep(8); }
    res[0] = ok; res[1] = (int)(wall_clock64() - t0);
}
__global__ void k_set(unsigned long long* flag) { if (threadIdx.x == 0) st(flag, 1ull); }

// chain: go[i] (16 replicas, 4 KB apart); done[(i & 1)][b] = i + 1 written by worker b of launch i (a value, not a counter: nothing to re-arm)
constexpr int REPL = 16, RSTRIDE = 520;      // in 8-byte words
__global__ __launch_bounds__(256) void k_link(int i, int nworkers, unsigned long long* go, unsigned long long* done, int* fail, int work_clocks, int limit) {
    const int b = blockIdx.x;
    if (b == 0) {                                           // the "reducer": waits for every worker of launch i - 1, publishes go[i]
        bool ok = true;
        if (i > 0) {
            const unsigned long long* d = done + (size_t)((i - 1) & 1) * nworkers;
            for (int w = threadIdx.x; w < nworkers; w += 256) {
                bool okw = false;
                for (int s = 0; s < limit && !okw; s++) { okw = ld(d + w) == (unsigned long long)i; if (!okw) __builtin_amdgcn_s_sleep(2); }
                ok = ok && okw;
            }
        }
        if (!ok) atomicOr(fail, 1);
        __syncthreads();
        if (threadIdx.x < REPL) st(go + (size_t)threadIdx.x * RSTRIDE + i, 1ull);
        return;
    }
    const unsigned long long* g = go + (size_t)((b * 4 + (threadIdx.x >> 6)) % REPL) * RSTRIDE + i;
    bool ok = false;
    for (int s = 0; s < limit && !ok; s++) { ok = ld(g) == 1ull; if (!ok) __builtin_amdgcn_s_sleep(4); }
    if (!ok) { if ((threadIdx.x & 63) == 0) atomicOr(fail, 2); }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < work_clocks) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    if (threadIdx.x == 0) st(done + (size_t)(i & 1) * nworkers + (b - 1), (unsigned long long)(i + 1));
}
int main(int argc, char** argv) {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long* flag; int* res; hipMalloc(&flag, 8); hipMalloc(&res, 8);
    for (int any = 0; any < 2; any++) {
        hipMemsetAsync(flag, 0, 8, s); hipMemsetAsync(res, 0, 8, s);
        const int limit = 100000;                           // ~ 100 000 x (64 x 8 clocks + a trip) = tens of ms
        hipExtLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, flag, res, limit);
        hipExtLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, s, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0, flag);
        hipStreamSynchronize(s);
        int h[2]; hipMemcpy(h, res, 8, hipMemcpyDeviceToHost);
        printf("test 1 (%s): waiter %s the flag after %.1f us\n", any ? "any-order" : "in-order ", h[0] ? "SAW" : "never saw", h[1] * 0.01);
    }
    const int L = 400, nworkers = 1448;
    unsigned long long* go; unsigned long long* done; int* fail;
    hipMalloc(&go, (size_t)REPL * RSTRIDE * 8); hipMalloc(&done, (size_t)2 * nworkers * 8); hipMalloc(&fail, 4);
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e0, e1, ej; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    const char* names[3] = {"one stream, in-order ", "one stream, any-order", "two streams, no order"};
    for (int work_us : {0, 5, 10, 40}) for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f; int hf = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipMemsetAsync(go, 0, (size_t)REPL * RSTRIDE * 8, s); hipMemsetAsync(done, 0, (size_t)2 * nworkers * 8, s); hipMemsetAsync(fail, 0, 4, s);
            hipStreamSynchronize(s); hipStreamSynchronize(s2);
            hipEventRecord(e0, s);
            if (mode == 2) { hipEventRecord(ej, s); hipStreamWaitEvent(s2, ej, 0); }
            for (int i = 0; i < L; i++)
                hipExtLaunchKernelGGL(k_link, dim3(nworkers + 1), dim3(256), 0, (mode == 2 && (i & 1)) ? s2 : s, nullptr, nullptr, (mode == 1 && i > 0) ? hipExtAnyOrderLaunch : 0, i, nworkers, go, done, fail, work_us * 100, 400000);
            if (mode == 2) { hipEventRecord(ej, s2); hipStreamWaitEvent(s, ej, 0); }
            hipEventRecord(e1, s); hipStreamSynchronize(s); hipStreamSynchronize(s2);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
        }
        printf("test 2: work %2d us, %s: %.2f us per launch (fail=%d)\n", work_us, names[mode], best * 1000.f / L, hf);
    }
    return 0;
}

// what does a cross-stream dependency cost against an in-stream one?  A on S1 -> B on S2 -> A on S1 ... vs A, B, A, B on one stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_a(int* p) { if (p == nullptr) p[0] = 1; }
__global__ void k_b(int* p) { if (p == nullptr) p[0] = 2; }
__global__ void k_spin(volatile int* flag, int want) { if (threadIdx.x == 0) { int n = 0; while (*flag < want && n < (1 << 24)) { __builtin_amdgcn_s_sleep(2); n++; } } }
__global__ void k_bump(int* flag) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(flag, 1); }
int main() {
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    const int N = 2000;
    hipEvent_t ea[N], eb[N];
    for (int i = 0; i < N; i++) { hipEventCreateWithFlags(&ea[i], hipEventDisableTiming); hipEventCreateWithFlags(&eb[i], hipEventDisableTiming); }
    auto now = [] { return std::chrono::high_resolution_clock::now(); };
    for (int rep = 0; rep < 2; rep++) {
        hipDeviceSynchronize();
        auto t0 = now();
        for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_a, dim3(2895), dim3(128), 0, s1, d); hipLaunchKernelGGL(k_b, dim3(34), dim3(256), 0, s1, d); }
        hipStreamSynchronize(s1);
        auto t1 = now();
        printf("one stream, A then B:              %.2f us per pair\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        t0 = now();
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(k_a, dim3(2895), dim3(128), 0, s1, d); hipEventRecord(ea[i], s1); hipStreamWaitEvent(s2, ea[i], 0);
            hipLaunchKernelGGL(k_b, dim3(34), dim3(256), 0, s2, d); hipEventRecord(eb[i], s2); hipStreamWaitEvent(s1, eb[i], 0);
        }
        hipStreamSynchronize(s1); hipStreamSynchronize(s2);
        t1 = now();
        printf("two streams, event ping-pong:      %.2f us per pair\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        // B resident early, waiting in-kernel for A's end (flag bumped by a tiny kernel after A in S1 stands for A's last block); S1 waits for B by event
        hipMemset(d, 0, 64); hipDeviceSynchronize();
        t0 = now();
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(k_spin, dim3(34), dim3(256), 0, s2, (volatile int*)d, i + 1); hipEventRecord(eb[i], s2);
            hipLaunchKernelGGL(k_a, dim3(2895), dim3(128), 0, s1, d); hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, s1, d);
            hipStreamWaitEvent(s1, eb[i], 0);
        }
        hipStreamSynchronize(s1); hipStreamSynchronize(s2);
        t1 = now();
        printf("B resident + spinning on a flag:   %.2f us per triple (A, bump, B)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
    return 0;
}

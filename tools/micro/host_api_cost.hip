// Host-side cost of the HIP calls a frame is made of (no GPU wait inside the timed loops).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Big { char b[400]; };
__global__ void k_small(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 1u << 30) *p = 1; }
__global__ void k_big(Big a, int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 1u << 30) *p = a.b[0]; }

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int N = 2000;
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    Big big{};
    // warm up
    for (int i = 0; i < 100; ++i) { hipLaunchKernelGGL(k_small, dim3(64), dim3(64), 0, s1, nullptr); hipLaunchKernelGGL(k_big, dim3(64), dim3(64), 0, s1, big, nullptr); }
    hipStreamSynchronize(s1);
    double t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(64), 0, s1, nullptr);
    double t1 = now();
    hipStreamSynchronize(s1);
    printf("launch (8 B args)      : %.2f us per call\n", (t1 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(64), dim3(64), 0, s1, big, nullptr);
    t1 = now();
    hipStreamSynchronize(s1);
    printf("launch (408 B args)    : %.2f us per call\n", (t1 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) hipEventRecord(ev[i], s1);
    t1 = now();
    hipStreamSynchronize(s1);
    printf("hipEventRecord         : %.2f us per call\n", (t1 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) hipStreamWaitEvent(s2, ev[i], 0);
    t1 = now();
    hipStreamSynchronize(s2);
    printf("hipStreamWaitEvent     : %.2f us per call\n", (t1 - t0) / N);
    // a frame-like mix: 6 launches + 5 records + 5 waits
    t0 = now();
    for (int i = 0; i < N / 10; ++i) {
        for (int k = 0; k < 3; ++k) { hipLaunchKernelGGL(k_big, dim3(64), dim3(64), 0, s1, big, nullptr); hipEventRecord(ev[i * 10 + k], s1); hipStreamWaitEvent(s2, ev[i * 10 + k], 0); hipLaunchKernelGGL(k_big, dim3(64), dim3(64), 0, s2, big, nullptr); }
    }
    t1 = now();
    hipDeviceSynchronize();
    printf("mix of 6 launches + 3 records + 3 waits: %.2f us per group\n", (t1 - t0) / (N / 10));
    return 0;
}

// Micro-benchmark: cost of ordering two in-order chains on two HIP streams, per "frame":
//   chain A: 4 short kernels, then signal;  chain B: wait for A's signal of the same frame, then 2 longer kernels.
// Variants: (0) single stream, everything serial; (1) hipEvent record / hipStreamWaitEvent;
//           (2) hipStreamWriteValue64 / hipStreamWaitValue64 on signal memory; (3) in-kernel bounded spin on a flag.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void busy(long long cycles, unsigned long long* sink)
{
    const long long t0 = wall_clock64();
    unsigned long long acc = 0;
    while (wall_clock64() - t0 < cycles) acc += 1;
    if (acc == 0xFFFFFFFFFFFFull) *sink = acc;
}

__global__ void set_flag(unsigned long long* flag, unsigned long long v)
{
    __threadfence();
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void busy_after_flag(long long cycles, unsigned long long* sink, unsigned long long* flag, unsigned long long v, int* err)
{
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < v) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 22)) { *err = 1; break; }
        }
    }
    __syncthreads();
    const long long t0 = wall_clock64();
    unsigned long long acc = 0;
    while (wall_clock64() - t0 < cycles) acc += 1;
    if (acc == 0xFFFFFFFFFFFFull) *sink = acc;
}

int main()
{
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned long long* sink; CK(hipMalloc((void**)&sink, 8));
    unsigned long long* sig = nullptr;
    hipError_t se = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("signal memory: %s\n", hipGetErrorString(se));
    unsigned long long* flag; CK(hipMalloc((void**)&flag, 8)); CK(hipMemset(flag, 0, 8));
    if (sig) CK(hipMemset(sig, 0, 8));
    int* err; CK(hipMalloc((void**)&err, 4)); CK(hipMemset(err, 0, 4));
    const int frames = 200;
    const long long cA = 20 * 100, cB = 80 * 100;   // clock64 counts at 100 MHz here? measured below
    // calibrate clock64 rate
    {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, sa, 1000000LL, sink);
        CK(hipStreamSynchronize(sa));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("1e6 clock64 ticks = %.1f us (incl. launch)\n", us);
    }
    std::vector<hipEvent_t> ev(frames);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int variant = 0; variant < 4; ++variant) {
        if (variant == 2 && !sig) continue;
        CK(hipMemset(flag, 0, 8));
        if (sig) CK(hipMemset(sig, 0, 8));
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < frames; ++k) {
            hipStream_t b = variant == 0 ? sa : sb;
            for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, sa, cA, sink);
            if (variant == 1) { CK(hipEventRecord(ev[k], sa)); CK(hipStreamWaitEvent(sb, ev[k], 0)); }
            if (variant == 2) { CK(hipStreamWriteValue64(sa, sig, (uint64_t)(k + 1), 0)); CK(hipStreamWaitValue64(sb, sig, (uint64_t)(k + 1), hipStreamWaitValueGte, ~0ull)); }
            if (variant == 3) {
                hipLaunchKernelGGL(set_flag, dim3(1), dim3(64), 0, sa, flag, (unsigned long long)(k + 1));
                hipLaunchKernelGGL(busy_after_flag, dim3(64), dim3(64), 0, b, cB, sink, flag, (unsigned long long)(k + 1), err);
                hipLaunchKernelGGL(busy, dim3(64), dim3(64), 0, b, cB, sink);
            } else {
                for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(busy, dim3(64), dim3(64), 0, b, cB, sink);
            }
        }
        CK(hipStreamSynchronize(sa));
        CK(hipStreamSynchronize(sb));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        int h = 0; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        printf("variant %d: %.1f us/frame (spin timeout flag %d)\n", variant, us / frames, h);
    }
    return 0;
}

// Micro-benchmark: does hipExtLaunchKernelGGL(..., flags = hipExtAnyOrderLaunch) let two kernels of ONE stream overlap on this
// runtime / gfx950?  (hip_ext.h says the flag "is not supported on AMD GFX9xx boards".)  Two kernels that each sleep 100 us on one
// workgroup, back to back on one stream: ~200 us in order, ~100 us if the second one may start before the first has ended.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/anyorder_probe.hip -o anyorder_probe && ./anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void nap(long long ticks, long long* out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
int main()
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    long long* d = nullptr;
    CK(hipMalloc(&d, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
        float best = 1e9f;
        long long h[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s));
            hipExtLaunchKernelGGL(nap, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, 10000ll, d);
            hipExtLaunchKernelGGL(nap, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, 10000ll, d + 2);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
        }
        printf("{\"second_launch_flags\": %d, \"two_100us_kernels_us\": %.1f, \"second_started_us_after_first_started\": %.1f}\n", flags, best * 1e3f, (h[2] - h[0]) / 100.0);
    }
    return 0;
}

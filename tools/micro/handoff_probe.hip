// Micro-benchmark: what a frame-granular hand-off between two persistent chain kernels costs on this runtime.
//  (1) does hipStreamWaitValue32 see a value a KERNEL wrote (signal memory / device memory / mapped host memory), and how long
//      after the write does the next kernel of the waiting stream start?
//  (2) a consumer kernel polling a flag a producer kernel publishes with an agent-scope store: publish -> observe latency,
//      with the consumer resident on another CU (both kernels run at the same time on two streams).
// All times on the device's 100 MHz wall clock.   hipcc --offload-arch=gfx950 -O2 handoff_probe.hip -o handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(unsigned* flag, unsigned v, long long delay_ticks, long long* stamp)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) {
        stamp[0] = wall_clock64();
        __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void stamp_kernel(long long* stamp) { if (threadIdx.x == 0) stamp[1] = wall_clock64(); }

// frames: the producer publishes frame numbers 1..n every `period` ticks; the consumer polls and stamps when it saw each
__global__ void producer_frames(unsigned* flag, int n, long long period, long long* t_pub)
{
    long long t = wall_clock64();
    for (int k = 1; k <= n; ++k) {
        while (wall_clock64() - t < period) __builtin_amdgcn_s_sleep(2);
        t = wall_clock64();
        if (threadIdx.x == 0) {
            t_pub[k] = t;
            __hip_atomic_store(flag, (unsigned)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void consumer_frames(unsigned* flag, int n, int sleep_arg, long long* t_seen, int* err)
{
    for (int k = 1; k <= n; ++k) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)k) {
                if (sleep_arg == 1) __builtin_amdgcn_s_sleep(1);
                else if (sleep_arg == 8) __builtin_amdgcn_s_sleep(8);
                else if (sleep_arg == 32) __builtin_amdgcn_s_sleep(32);
                if (++spins > (1u << 24)) { *err = 1; break; }
            }
            t_seen[k] = wall_clock64();
        }
        __syncthreads();
    }
}

int main()
{
    int can = -1;
    hipError_t ae = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("{\"attr_can_use_stream_wait_value\": %d, \"attr_err\": \"%s\"}\n", can, hipGetErrorString(ae));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    long long* stamp; CK(hipMalloc((void**)&stamp, 64)); CK(hipMemset(stamp, 0, 64));
    // ---- (1) stream wait value on three kinds of memory, written by a kernel
    for (int kind = 0; kind < 3; ++kind) {
        unsigned* flag = nullptr;
        unsigned* host_view = nullptr;
        hipError_t e;
        const char* name = kind == 0 ? "signal" : (kind == 1 ? "device" : "mapped_host");
        if (kind == 0) e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
        else if (kind == 1) e = hipMalloc((void**)&flag, 8);
        else { e = hipHostMalloc((void**)&host_view, 8, hipHostMallocMapped); if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&flag, host_view, 0); }
        if (e != hipSuccess) { printf("{\"wait_value\": \"%s\", \"alloc\": \"%s\"}\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        if (kind == 2) *host_view = 0; else CK(hipMemset(flag, 0, 8));
        CK(hipDeviceSynchronize());
        double lat[5];
        int ok = 1;
        for (int rep = 0; rep < 5 && ok; ++rep) {
            CK(hipMemset(stamp, 0, 64));
            if (kind == 2) *host_view = 0; else CK(hipMemset(flag, 0, 8));
            CK(hipDeviceSynchronize());
            e = hipStreamWaitValue32(sb, flag, 7u, hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { printf("{\"wait_value\": \"%s\", \"wait_call\": \"%s\"}\n", name, hipGetErrorString(e)); (void)hipGetLastError(); ok = 0; break; }
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, sb, stamp);
            hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, sa, flag, 7u, 20000ll /* 200 us */, stamp);
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sb));
            long long h[2]; CK(hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost));
            lat[rep] = (double)(h[1] - h[0]) * 0.01;
        }
        if (ok) printf("{\"wait_value\": \"%s\", \"kernel_write_to_next_kernel_us\": [%.1f, %.1f, %.1f, %.1f, %.1f]}\n", name, lat[0], lat[1], lat[2], lat[3], lat[4]);
    }
    // ---- (2) flag polled by a resident consumer kernel
    {
        unsigned* flag; CK(hipMalloc((void**)&flag, 8));
        const int n = 200;
        long long *t_pub, *t_seen; CK(hipMalloc((void**)&t_pub, 8 * (n + 1))); CK(hipMalloc((void**)&t_seen, 8 * (n + 1)));
        int* err; CK(hipMalloc((void**)&err, 4)); CK(hipMemset(err, 0, 4));
        for (int sl : {0, 1, 8, 32}) {
            CK(hipMemset(flag, 0, 8));
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(consumer_frames, dim3(1), dim3(256), 0, sb, flag, n, sl, t_seen, err);
            hipLaunchKernelGGL(producer_frames, dim3(1), dim3(256), 0, sa, flag, n, 1500ll /* 15 us */, t_pub);
            CK(hipDeviceSynchronize());
            std::vector<long long> p(n + 1), s(n + 1);
            CK(hipMemcpy(p.data(), t_pub, 8 * (n + 1), hipMemcpyDeviceToHost));
            CK(hipMemcpy(s.data(), t_seen, 8 * (n + 1), hipMemcpyDeviceToHost));
            double sum = 0, mx = 0;
            for (int k = 20; k <= n; ++k) { const double d = (double)(s[k] - p[k]) * 0.01; sum += d; if (d > mx) mx = d; }
            int h = 0; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
            printf("{\"poll_sleep\": %d, \"publish_to_seen_us_mean\": %.2f, \"max\": %.2f, \"timeout\": %d}\n", sl, sum / (n - 19), mx, h);
        }
    }
    return 0;
}

// pipe_probe.hip -- do two HIP streams share a dispatch pipe?  (experiment, not part of the library)
// A dispatch that cannot be placed completely (more workgroups than the chip holds) keeps its queue's pipe busy until the
// last workgroup is placed; a tiny kernel on another stream completes at once if its queue sits on another pipe and only
// after the big one has been placed if it shares the pipe.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/pipe_probe.hip -o /tmp/pipe_probe && /tmp/pipe_probe hhnlN
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void blocker(long long ticks)
{
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = 0;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void tiny(int* p) { if (threadIdx.x == 0) *p = 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const char* order = argc > 1 ? argv[1] : "hhnlN";
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    std::vector<hipStream_t> st;
    for (const char* c = order; *c; ++c) {
        hipStream_t s;
        if (*c == 'N') hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        else hipStreamCreateWithPriority(&s, hipStreamNonBlocking, *c == 'h' ? greatest : (*c == 'l' ? least : (least + greatest) / 2));
        st.push_back(s);
    }
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipFuncSetAttribute(reinterpret_cast<const void*>(blocker), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    int* d;
    hipMalloc(&d, 4);
    const int n = (int)st.size();
    // warm every stream (queues are created lazily)
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[i], d); }
    hipDeviceSynchronize();
    printf("order %s, %d CUs; entry (A, B) = microseconds until a tiny kernel on B completes while A places 3 x %d one-per-CU workgroups of 100 us\n", order, cus, cus);
    for (int a = 0; a < n; ++a) {
        printf("%c%d:", order[a], a);
        for (int b = 0; b < n; ++b) {
            if (a == b) { printf("     -"); continue; }
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipDeviceSynchronize();
                const double t0 = now_us();
                hipLaunchKernelGGL(blocker, dim3(3 * cus), dim3(64), 150 * 1024, st[a], 10000ll);   // 100 us per workgroup at 100 MHz
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[b], d);
                hipStreamSynchronize(st[b]);
                const double t1 = now_us();
                if (t1 - t0 < best) best = t1 - t0;
            }
            printf(" %5.0f", best);
        }
        printf("\n");
    }
    return 0;
}

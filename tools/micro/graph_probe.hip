// graph_probe.hip -- would a hipGraph shorten a chain of dependent kernel launches?  (experiment, not part of the library)
// A batch of the engine is ~12 dependent launches whose arguments (a 400-byte block of pointers, ring slots and counters) change
// with every batch.  Timed: the chain as plain stream launches, as one hipGraphLaunch of the captured chain, and as that graph
// with every kernel node's parameters replaced before each launch -- host time per chain and device time from the first
// kernel's start to the last one's end (event pair on the stream).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/graph_probe.hip -o /tmp/graph_probe && /tmp/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { char block[400]; int* out; };
__global__ void step_kernel(Args a)
{
    // ~3 us of work for one workgroup
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 300) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && a.block[0] == 77) *a.out = 1;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    constexpr int kChain = 12, kReps = 200;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* out;
    CK(hipMalloc(&out, 4));
    Args args{};
    args.out = out;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto measure = [&](const char* name, auto&& submit) -> int {
        for (int i = 0; i < 20; ++i) submit(i);
        CK(hipStreamSynchronize(s));
        double host = 0.0, dev = 0.0;
        for (int r = 0; r < kReps; ++r) {
            CK(hipEventRecord(e0, s));
            const double t0 = now_us();
            submit(r);
            host += now_us() - t0;
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            dev += 1e3 * ms;
        }
        printf("{\"chain\": \"%s\", \"kernels\": %d, \"host_us_per_chain\": %.1f, \"device_us_per_chain\": %.1f, \"device_us_per_kernel\": %.2f}\n", name, kChain,
               host / kReps, dev / kReps, dev / kReps / kChain);
        return 0;
    };
    // (a) plain launches
    if (measure("stream launches", [&](int r) {
            args.block[1] = (char)r;
            for (int k = 0; k < kChain; ++k) hipLaunchKernelGGL(step_kernel, dim3(1), dim3(64), 0, s, args);
        })) return 1;
    // (b) captured graph, launched as it is
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < kChain; ++k) hipLaunchKernelGGL(step_kernel, dim3(1), dim3(64), 0, s, args);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    if (measure("graph launch", [&](int) { (void)hipGraphLaunch(exec, s); })) return 1;
    // (c) the graph with new kernel arguments in every node before every launch
    size_t n_nodes = 0;
    CK(hipGraphGetNodes(graph, nullptr, &n_nodes));
    std::vector<hipGraphNode_t> nodes(n_nodes);
    CK(hipGraphGetNodes(graph, nodes.data(), &n_nodes));
    if (measure("graph launch + node parameter updates", [&](int r) {
            args.block[1] = (char)r;
            void* kargs[1] = {&args};
            hipKernelNodeParams p{};
            p.func = reinterpret_cast<void*>(step_kernel);
            p.gridDim = dim3(1);
            p.blockDim = dim3(64);
            p.kernelParams = kargs;
            for (hipGraphNode_t n : nodes) (void)hipGraphExecKernelNodeSetParams(exec, n, &p);
            (void)hipGraphLaunch(exec, s);
        })) return 1;
    return 0;
}

// Micro-benchmark: the skeleton of a batch's MASK CHAIN run two ways --
//   (A) one launch per frame (what the engine does: the kernel boundary is the dependency between frame t - 1 and frame t), and
//   (B) ONE resident grid of workers that claim tasks by TICKET from a queue filled in dependency order (frame-major): task
//       (t, object, band) waits until all bands of (t - 1, object) have signalled, so it only ever waits for tasks with smaller
//       tickets -- claimed by workers that already run -- and needs no co-residency of the whole chain (rounds 2 - 3's persistent
//       chain did).
// A task is the skeleton of a mask-frame workgroup: read the band's words of the object's bit plane of frame t - 1, one dependent
// load per non-empty word from a cold "flow" image, OR the word into the plane of frame t some words further on (across the band
// boundary: data flows between workgroups), + `spin` microseconds of arithmetic that stands for the walk's bookkeeping.
// Both variants must leave the same planes (checked bit for bit).  The grid's tasks exchange data INSIDE a kernel across the eight
// XCDs' L2 caches: with agent-scope fences (FENCES=1: buffer_wbl2 / buffer_inv around every task) or -- the default -- with
// every word that crosses a task boundary moved by agent-scope atomics (loads of the source plane, ORs into the destination, the
// counters), which go to the coherence point and need no cache maintenance; the flow image is read-only and read normally.
// STOP RULE (set before the run): the resident grid is worth building into the engine only if it shortens this chain by >= 15 %
// at 8 AND at 64 objects; the engine's other chains would then be moved onto the same grid.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/ticket_grid_probe.hip -o ticket_grid_probe && ./ticket_grid_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kW = 9600;          // 32-bit words of a 640 x 480 bit plane
constexpr int kBands = 24;        // workgroups per object and frame (the engine's 20-row bands)
constexpr int kPer = kW / kBands; // 400 words per band
constexpr int kShift = 37;        // a word's bits land 37 words further on (crosses a band boundary for a tenth of the words)
constexpr int kThreads = 256;

struct Args {
    unsigned* planes;       // [T + 1][O][kW]   plane 0 = the delivered mask, planes 1 .. T zeroed
    const uint4* flow;      // [T][O][kW]       one 16-byte element per word, cold
    unsigned* done;         // [T][O]           bands of (t, o) that have signalled
    unsigned* next;         // ticket counter
    int O, T;
    int spin_ticks;         // 100 MHz ticks of stand-in arithmetic per task
    int fences;             // grid variant: agent-scope fences around a task instead of atomic loads of the source plane
    unsigned* error;        // set if a worker gave up waiting (a watchdog: this probe must never hang a box)
};

template <bool COHERENT_LOADS>
__device__ __forceinline__ void task_body(const Args& a, int t, int o, int b)
{
    const unsigned* src = a.planes + ((size_t)t * a.O + o) * kW;
    unsigned* dst = a.planes + ((size_t)(t + 1) * a.O + o) * kW;
    const uint4* fl = a.flow + ((size_t)t * a.O + o) * kW;
    const long long t0 = wall_clock64();
    for (int i = threadIdx.x; i < kPer; i += kThreads) {
        const int w = b * kPer + i;
        const unsigned v = COHERENT_LOADS ? __hip_atomic_load(&src[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[w];
        if (v) {
            const uint4 f = fl[w];                                  // the dependent round trip to the flow image
            const int tw = (w + kShift + (int)(f.x & 1u)) % kW;      // (f is zero: + kShift)
            atomicOr(&dst[tw], (v << 1) | (v >> 31));
        }
    }
    while (wall_clock64() - t0 < a.spin_ticks) __builtin_amdgcn_s_sleep(1);
}

// (A) one launch per frame: grid (kBands, O)
__global__ __launch_bounds__(kThreads) void frame_kernel(Args a, int t) { task_body<false>(a, t, blockIdx.y, blockIdx.x); }

// (B) resident workers
__global__ __launch_bounds__(kThreads) void grid_kernel(Args a)
{
    __shared__ unsigned s_ticket;
    const unsigned total = (unsigned)a.T * a.O * kBands;
    const long long k0 = wall_clock64();
    for (;;) {
        if (wall_clock64() - k0 > 50000000ll) { if (threadIdx.x == 0) atomicOr(a.error, 2u); return; }   // (half a second: never hang a box)
        if (threadIdx.x == 0) s_ticket = atomicAdd(a.next, 1u);
        __syncthreads();
        const unsigned ticket = s_ticket;
        __syncthreads();
        if (ticket >= total) return;
        const int t = ticket / (a.O * kBands), r = ticket % (a.O * kBands), o = r / kBands, b = r % kBands;
        if (t > 0) {
            if (threadIdx.x == 0) {
                const long long w0 = wall_clock64();
                while (__hip_atomic_load(&a.done[(t - 1) * a.O + o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)kBands) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - w0 > 20000000ll) { atomicOr(a.error, 1u); break; }   // 0.2 s
                }
            }
            __syncthreads();
            if (a.fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (every wave: buffer_inv)
        }
        if (a.fences) {
            task_body<false>(a, t, o, b);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                 // (buffer_wbl2)
        } else {
            task_body<true>(a, t, o, b);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");             // (the ORs acknowledged: no cache maintenance)
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&a.done[t * a.O + o], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static double median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int T = 6;
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 11;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("{\"what\": \"mask-chain skeleton: %d frames x objects x %d bands, launch per frame (A) vs ticket-ordered resident grid (B)\", \"cus\": %d, \"runs\": [\n", T, kBands, cus);
    bool first = true;
    for (int O : {8, 64}) {
        for (int spin_us : {0, 5}) {
            Args a{};
            a.O = O; a.T = T; a.spin_ticks = spin_us * 100;
            const size_t plane_n = (size_t)(T + 1) * O * kW, flow_n = (size_t)T * O * kW;
            CK(hipMalloc(&a.planes, plane_n * 4));
            uint4* flow = nullptr;
            CK(hipMalloc(&flow, flow_n * sizeof(uint4)));
            CK(hipMemset(flow, 0, flow_n * sizeof(uint4)));
            a.flow = flow;
            CK(hipMalloc(&a.done, (size_t)T * O * 4));
            CK(hipMalloc(&a.next, 4));
            CK(hipMalloc(&a.error, 4));
            CK(hipMemset(a.error, 0, 4));
            a.fences = getenv("FENCES") ? atoi(getenv("FENCES")) : 0;
            // the delivered mask: a blob of ~26 k pixels (rows 150 .. 320, five words per row)
            std::vector<unsigned> mask((size_t)O * kW, 0u);
            for (int o = 0; o < O; ++o)
                for (int row = 150; row < 320; ++row)
                    for (int c = 7; c < 12; ++c) mask[(size_t)o * kW + row * 20 + c] = 0xFFFFFFFFu >> ((row + o) % 3);
            auto reset = [&]() -> int {
                CK(hipMemsetAsync(a.planes, 0, plane_n * 4, s));
                CK(hipMemcpyAsync(a.planes, mask.data(), mask.size() * 4, hipMemcpyHostToDevice, s));
                CK(hipMemsetAsync(a.done, 0, (size_t)T * O * 4, s));
                CK(hipMemsetAsync(a.next, 0, 4, s));
                CK(hipStreamSynchronize(s));
                return 0;
            };
            std::vector<unsigned> ref(plane_n), got(plane_n);
            std::vector<float> ta, tb[3];
            const int w1 = getenv("WORKERS") ? atoi(getenv("WORKERS")) : cus;
            const int workers[3] = {w1, 2 * cus, 4 * cus};
            for (int rep = 0; rep < reps; ++rep) {
                if (reset()) return 1;
                if (getenv("VERBOSE")) fprintf(stderr, "objects %d spin %d rep %d: launches\n", O, spin_us, rep);
                CK(hipEventRecord(e0, s));
                for (int t = 0; t < T; ++t) hipLaunchKernelGGL(frame_kernel, dim3(kBands, O), dim3(kThreads), 0, s, a, t);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) ta.push_back(ms * 1e3f);
                if (rep == 0) CK(hipMemcpy(ref.data(), a.planes, plane_n * 4, hipMemcpyDeviceToHost));
                for (int k = 0; k < 3; ++k) {
                    if (reset()) return 1;
                    const int G = std::min(workers[k], T * O * kBands);
                    if (getenv("VERBOSE")) fprintf(stderr, "  grid of %d workers\n", G);
                    CK(hipEventRecord(e0, s));
                    hipLaunchKernelGGL(grid_kernel, dim3(G), dim3(kThreads), 0, s, a);
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) tb[k].push_back(ms * 1e3f);
                    unsigned err = 0;
                    CK(hipMemcpy(&err, a.error, 4, hipMemcpyDeviceToHost));
                    if (err) {
                        unsigned nx = 0;
                        std::vector<unsigned> dn((size_t)T * O);
                        CK(hipMemcpy(&nx, a.next, 4, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(dn.data(), a.done, dn.size() * 4, hipMemcpyDeviceToHost));
                        unsigned long long sum = 0;
                        for (unsigned v : dn) sum += v;
                        printf("WATCHDOG objects %d workers %d: error %u, tickets handed out %u of %d, bands signalled %llu, kernel %.1f ms\n", O, G, err, nx, T * O * kBands, sum, ms);
                        return 3;
                    }
                    if (rep == 0) {
                        CK(hipMemcpy(got.data(), a.planes, plane_n * 4, hipMemcpyDeviceToHost));
                        if (std::memcmp(ref.data(), got.data(), plane_n * 4) != 0) { printf("MISMATCH objects %d workers %d\n", O, G); return 2; }
                    }
                }
            }
            const double A = median(ta);
            printf("%s {\"objects\": %d, \"spin_us_per_task\": %d, \"launch_per_frame_us\": %.1f, \"per_frame_us\": %.2f", first ? "" : ",\n", O, spin_us, A, A / T);
            for (int k = 0; k < 3; ++k) {
                const double B = median(tb[k]);
                printf(", \"grid_%dx_cus_us\": %.1f, \"grid_%dx_vs_launches\": %.3f", workers[k] / cus, B, workers[k] / cus, B / A);
            }
            printf(", \"fences\": %d, \"planes_identical\": true}", a.fences);
            fflush(stdout);
            first = false;
            CK(hipFree(a.planes)); CK(hipFree(flow)); CK(hipFree(a.done)); CK(hipFree(a.next)); CK(hipFree(a.error));
        }
    }
    printf("\n]}\n");
    return 0;
}

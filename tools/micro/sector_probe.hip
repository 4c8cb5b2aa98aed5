// sector_probe.hip -- what does the memory system give a kernel shaped like the flow measurement (K1)?  (experiment, not part
// of the library)
//   (1) device-to-device copy: the streaming figure the roofline's 8 TB/s is usually compared with;
//   (2) random 64-byte sectors: every thread reads L words at hashed, sector-aligned offsets of a 4 GiB buffer -- from a launch
//       of K1's size (768 k sectors) up to a saturating one;
//   (3) the skeleton of K1's launch: G workgroups of 1024 threads, each workgroup reads a 38 400-byte bit plane of its own
//       (three 16-byte loads per thread), meets at two barriers (where K1 scans), ~730 of its threads then gather one depth
//       sample (4 bytes) and one flow sample (8 bytes) 35 mask pixels apart inside a 150 x 170 window of the workgroup's own
//       640 x 480 images, two more barriers (where K1 compacts), 20 bytes written per sample -- no arithmetic at all: the
//       floor of that launch shape.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/sector_probe.hip -o /tmp/sector_probe && /tmp/sector_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int L>
__global__ __launch_bounds__(1024) void random_sectors(const unsigned* buf, unsigned n_sectors_mask, int active, unsigned* out, unsigned salt)
{
    if ((int)threadIdx.x >= active) return;
    const unsigned t = blockIdx.x * 1024u + threadIdx.x;
    unsigned v[L];
#pragma unroll
    for (int j = 0; j < L; ++j) v[j] = buf[(size_t)(hash32(t * L + j + salt) & n_sectors_mask) * 16];
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) s += v[j];
    if (s == 0x12345678u) out[t] = s;   // (never: keeps the loads)
}

__global__ __launch_bounds__(1024) void k1_skeleton(const uint4* planes, const float* depth, const float2* flow, uint4* recs, int n_cand,
                                                    int gathers, int barriers)
{
    __shared__ unsigned s_x[16];
    const int g = blockIdx.x;
    const uint4* p = planes + (size_t)g * 2400;
    uint4 q[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) q[j] = p[min((int)threadIdx.x * 3 + j, 2399)];
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) c += __popc(q[j].x) + __popc(q[j].y) + __popc(q[j].z) + __popc(q[j].w);
    if (barriers) {
        if ((threadIdx.x & 63) == 0) s_x[threadIdx.x >> 6] = c;
        __syncthreads();
        c += s_x[(threadIdx.x >> 6) ^ 1];
        __syncthreads();
    }
    float z = 0.f;
    float2 f = make_float2(0.f, 0.f);
    const int i = (int)threadIdx.x;
    if (i < n_cand && gathers) {
        const int pix = i * 35, row = 150 + pix / 150, col = 240 + pix % 150 + (int)(c & 1u);
        const size_t at = (size_t)g * 640 * 480 + (size_t)row * 640 + col;
        z = depth[at];
        f = flow[at];
    }
    if (barriers) {
        if ((threadIdx.x & 63) == 0) s_x[threadIdx.x >> 6] = __float_as_uint(z);
        __syncthreads();
        c += s_x[(threadIdx.x >> 6) ^ 1];
        __syncthreads();
    }
    if (i < n_cand) {
        // 20 bytes per record in K1; here 16 + the 4 of a neighbour, same number of write sectors
        recs[(size_t)g * 1024 + i] = make_uint4(c, __float_as_uint(z), __float_as_uint(f.x), __float_as_uint(f.y));
    }
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> double median_us(F&& launch, int reps = 9)
    {
        std::vector<float> ms(reps);
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(a, 0));
            launch(r);
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms[r], a, b));
        }
        std::sort(ms.begin(), ms.end());
        return 1e3 * ms[reps / 2];
    }
    // n launches back to back between one pair of events: per launch, without the events' own cost
    template <class F> double chain_us(F&& launch, int n = 10)
    {
        float ms = 0.f;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        for (int r = 0; r < n; ++r) launch(r);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        return 1e3 * ms / n;
    }
};

__global__ void empty_kernel() {}

int main()
{
    Timer T;
    const size_t GiB = 1ull << 30;
    unsigned *buf, *out;
    CK(hipMalloc(&buf, 4 * GiB));
    CK(hipMalloc(&out, 4 * GiB));
    CK(hipMemset(buf, 1, 4 * GiB));
    CK(hipMemset(out, 0, 4 * GiB));
    CK(hipDeviceSynchronize());
    {
        const double one = T.median_us([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); });
        const double ten = T.chain_us([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); });
        printf("{\"probe\": \"empty_kernel\", \"event_pair_us\": %.2f, \"back_to_back_us\": %.2f}\n", one, ten);
    }
    // (1)
    {
        const double us = T.median_us([&](int) { CK(hipMemcpyAsync(out, buf, 2 * GiB, hipMemcpyDeviceToDevice, 0)); });
        printf("{\"probe\": \"copy_d2d\", \"bytes\": %zu, \"us\": %.1f, \"read_plus_write_GBs\": %.0f}\n", 2 * GiB, us, 2.0 * 2 * GiB / us * 1e-3);
    }
    // (2)
    const unsigned mask = (unsigned)(4 * GiB / 64 - 1);
    auto sectors = [&](int grid, int active, int L) {
        auto go = [&](int r) {
            const unsigned salt = 0x9e3779b9u * (unsigned)(r + 1);
            if (L == 1) hipLaunchKernelGGL(random_sectors<1>, dim3(grid), dim3(1024), 0, 0, buf, mask, active, out, salt);
            else if (L == 2) hipLaunchKernelGGL(random_sectors<2>, dim3(grid), dim3(1024), 0, 0, buf, mask, active, out, salt);
            else if (L == 8) hipLaunchKernelGGL(random_sectors<8>, dim3(grid), dim3(1024), 0, 0, buf, mask, active, out, salt);
            else hipLaunchKernelGGL(random_sectors<16>, dim3(grid), dim3(1024), 0, 0, buf, mask, active, out, salt);
        };
        const double us = T.median_us(go), us_chain = T.chain_us(go);
        const double n = (double)grid * active * L;
        printf("{\"probe\": \"random_sectors\", \"workgroups\": %d, \"active_threads\": %d, \"loads_per_thread\": %d, \"sectors\": %.0f, "
               "\"us\": %.2f, \"us_back_to_back\": %.2f, \"Gsectors_per_s\": %.1f, \"GBs_at_64B\": %.0f}\n", grid, active, L, n, us, us_chain,
               n / us_chain * 1e-3, n * 64 / us_chain * 1e-3);
    };
    sectors(512, 750, 2);      // K1's launch: 768 k sectors
    sectors(512, 750, 1);
    sectors(384, 750, 2);
    sectors(2048, 750, 2);
    sectors(2048, 1024, 8);
    sectors(8192, 1024, 8);
    sectors(8192, 1024, 16);
    // (3)
    {
        const int Gmax = 2048;
        uint4* planes;
        float* depth;
        float2* flow;
        uint4* recs;
        CK(hipFree(out));
        CK(hipFree(buf));
        CK(hipMalloc(&planes, (size_t)Gmax * 38400));
        CK(hipMalloc(&depth, (size_t)Gmax * 640 * 480 * 4));
        CK(hipMalloc(&flow, (size_t)Gmax * 640 * 480 * 8));
        CK(hipMalloc(&recs, (size_t)Gmax * 1024 * 16));
        CK(hipMemset(planes, 0x11, (size_t)Gmax * 38400));
        CK(hipMemset(depth, 0, (size_t)Gmax * 640 * 480 * 4));
        CK(hipMemset(flow, 0, (size_t)Gmax * 640 * 480 * 8));
        CK(hipDeviceSynchronize());
        // every repetition touches other images than the one before (K1 reads every image once): rotate through the
        // 2048 image sets
        for (int G : {128, 384, 512, 1024, 2048}) {
            for (int variant = 0; variant < 4; ++variant) {
                const int gathers = variant != 1, barriers = variant != 2, n_cand = variant == 3 ? 1024 : 730;
                auto go = [&](int r) {
                    const int first = (r * G) % (Gmax - G + 1);
                    hipLaunchKernelGGL(k1_skeleton, dim3(G), dim3(1024), 0, 0, planes + (size_t)first * 2400, depth + (size_t)first * 640 * 480,
                                       flow + (size_t)first * 640 * 480, recs + (size_t)first * 1024, n_cand, gathers, barriers);
                };
                const double us_pair = T.median_us(go), us = T.chain_us(go, 4);
                const double declared = (double)G * (38400.0 + n_cand * 12.0 + n_cand * 20.0);
                printf("{\"probe\": \"k1_skeleton\", \"workgroups\": %d, \"variant\": \"%s\", \"us_event_pair\": %.2f, \"us\": %.2f, \"declared_GBs\": %.0f, \"frac_of_8TBs\": %.3f}\n", G,
                       variant == 0 ? "plane + barriers + gathers + records" : variant == 1 ? "no gathers" : variant == 2 ? "no barriers" : "1024 candidates",
                       us_pair, us, declared / us * 1e-3, declared / us * 1e-3 / 8000.0);
            }
        }
        // an empty launch of the same shape
        const double us = T.median_us([&](int) { hipLaunchKernelGGL(k1_skeleton, dim3(512), dim3(1024), 0, 0, planes, depth, flow, recs, 0, 0, 0); });
        printf("{\"probe\": \"k1_skeleton\", \"workgroups\": 512, \"variant\": \"plane only\", \"us\": %.2f}\n", us);
    }
    return 0;
}

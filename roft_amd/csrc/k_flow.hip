// k_flow.hip -- masked optical-flow + depth measurement of the velocity filter (gfx950).
//
// Reference: ImageOpticalFlowMeasurement<T>::freeze  include/ROFT/ImageOpticalFlowMeasurement.hpp:231-283
//   C      = cv::findNonZero(previous_segmentation_)            (row-major list)
//   cand   = C[0], C[R], C[2R], ...   R = subsampling radius (float-accumulated index, exact < 2^24)
//   keep   = flow finite and |.| < 1e9 (OpticalFlowUtilities.h:19-22) and 0 < Z < depth_maximum
//   y_i    = flow / scale;  H_i = T * [interaction matrix rows]  (:272-282)
//
// MI355X design: one workgroup of 16 waves per (object, frame of the batch) -- the measurement of a frame needs only
// the previous frame's mask plane, depth and the flow, none of the filter state, so all frames of a batch run in ONE
// launch (grid n_obj x T).  The previous frame's `obj` bit plane (W*H/8 bytes) is held in registers (or staged in LDS).  Plane words are in row-major order, so one block scan over per-thread popcounts of
// contiguous word chunks gives every word its starting rank; a word holds candidate c iff c*R falls into its rank
// interval, and the pixel is the (c*R - start)-th set bit of the word.  The candidate pixels go through a small
// list (LDS, or the global scratch for very large masks), then ONE thread per candidate gathers depth and flow --
// all gathers of an object are in flight together, and nothing but the ~N_mask/R candidate pixels is read from
// the images.  Output order equals the reference's (candidates in rank order, invalid ones dropped).
#include "plane_rank.h"

namespace roft {

__device__ __forceinline__ void flow_at(const void* data, const DevFlowFmt& f, int row, int col, float& dx,
                                        float& dy)
{
    size_t idx = ((size_t)row * (size_t)f.cols + (size_t)col);
    if (f.type == ROFT_FLOW_S16C2) {
        short2 p = reinterpret_cast<const short2*>(data)[idx];
        dx = (float)p.x / f.scale;
        dy = (float)p.y / f.scale;
    } else {
        float2 p = reinterpret_cast<const float2*>(data)[idx];
        dx = p.x / f.scale;
        dy = p.y / f.scale;
    }
}

__device__ __forceinline__ bool is_flow_valid(float fx, float fy)
{
    return !isnan(fx) && !isnan(fy) && fabs((double)fx) < 1e9 && fabs((double)fy) < 1e9;
}

// phase stamps (build with -DROFT_K1_PROFILE): K1TICK(i) stores the 100 MHz wall clock ticks since the previous stamp
#ifdef ROFT_K1_PROFILE
#define K1TICK(i) do { __syncthreads(); if (threadIdx.x == 0) { long long _t = wall_clock64(); st.dbg[i] = _t - k1_t0; k1_t0 = _t; } } while (0)
#else
#define K1TICK(i) do {} while (0)
#endif

constexpr int kFlowThreads = 1024;
constexpr int kCandLds = 2048;   // candidate work items kept in LDS; larger candidate sets go through a.cand

// position of the k-th (0-based) set bit of x, k < popc(x): five popcount halvings, no data-dependent loop
__device__ __forceinline__ int select_bit(uint32_t x, int k)
{
    int pos = 0;
#pragma unroll
    for (int wdt = 16; wdt >= 1; wdt >>= 1) {
        const int cnt = __popc((x >> pos) & ((1u << wdt) - 1u));
        if (k >= cnt) { k -= cnt; pos += wdt; }
    }
    return pos;
}

// One plane word with starting rank `rank`: candidate ci is the set bit of rank next = ci * R (hpp:237); a word holds
// it iff next falls into [rank, rank + popc).  Only a work item {word bits, word index, bit rank inside the word} is
// queued here -- the threads that own the dense words of the mask would otherwise serialise the pixel arithmetic of
// all their candidates.  Advances rank / next / ci.
__device__ __forceinline__ void emit_word(uint32_t bits, int w, int& rank, int& next, int& ci, int radius, uint2* list)
{
    const int pc = __popc(bits);
    if (next < rank + pc) {
        list[ci++] = make_uint2(bits, ((uint32_t)w << 5) | (uint32_t)(next - rank));
        next += radius;
        if (radius < 32) {   // strides below the word size: several candidates per word
#pragma nounroll
            while (next < rank + pc) {
                list[ci++] = make_uint2(bits, ((uint32_t)w << 5) | (uint32_t)(next - rank));
                next += radius;
                asm volatile("" : "+v"(next));   // keeps the loop a plain loop (no trip-count division, no unrolling)
            }
        }
    }
    rank += pc;
}

// One thread per candidate: pixel of the work item, gather depth + flow, validity, ordered compaction (rank order is
// thread order).
__device__ __forceinline__ int gather_candidates(const EngineArrays& a, const float* depth, const void* flow, int slot,
                                                 const uint2* list, int C, double depth_max, int* s_wave)
{
    const int W = a.cam.W, wpr = a.cam.wpr;
    FlowRec* recs = a.recs + (size_t)slot * a.cand_cap;
    int base = 0;
    for (int c0 = 0; c0 < C; c0 += blockDim.x) {
        const int ci = c0 + threadIdx.x;
        FlowRec r;
        bool ok = false;
        if (ci < C) {
            const uint2 item = list[ci];
            const int w = (int)(item.y >> 5);
            const int v = w / wpr, u = (w - v * wpr) * 32 + select_bit(item.x, (int)(item.y & 31u));
            // (round 6: the two gathers as NON-TEMPORAL loads -- ~500 k scattered sectors per launch, each read once, that need not
            //  push the other chains' lines out of the L2: 17.3 us against 18.0 per launch alone in four alternated runs; next to
            //  the other chains inside the spread of the windows, +1 % in the 20-step window)
            const float z = __builtin_nontemporal_load(&depth[(size_t)v * W + u]);
            float dx, dy;
            {
                const size_t fidx = ((size_t)(v / a.ffmt.grid) * (size_t)a.ffmt.cols + (size_t)(u / a.ffmt.grid));
                if (a.ffmt.type == ROFT_FLOW_S16C2) {
                    const unsigned pr = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(flow) + fidx);
                    dx = (float)(short)(pr & 0xFFFFu) / a.ffmt.scale;
                    dy = (float)(short)(pr >> 16) / a.ffmt.scale;
                } else {
                    typedef float f2v __attribute__((ext_vector_type(2)));   // (one 8-byte load)
                    const f2v pf = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(flow) + fidx);
                    dx = pf.x / a.ffmt.scale;
                    dy = pf.y / a.ffmt.scale;
                }
            }
            ok = is_flow_valid(dx, dy) && z > 0 && (double)z < depth_max;
            r.u = u; r.v = v; r.z = z; r.dx = dx; r.dy = dy;
        }
        int total;
        const int pos = block_exclusive_scan(ok ? 1 : 0, s_wave, &total);
        if (ok) recs[base + pos] = r;
        base += total;
    }
    return base;
}

// The control-block fields of the kernel, fetched together at its start: the empty asm pins all four loads before the
// first barrier (the compiler would otherwise sink each to its first use and pay the memory latency once per field).
struct FlowCtrl {
    int vel_stage, slot_prev;
    const float* depth;
    const void* flow;
};

// Timing runs (EngineArrays::k1_span): when this workgroup started and when its last wave is through, on the 100 MHz wall
// clock all workgroups share -- the host takes the launch's span, first workgroup in to last workgroup out, from them.
__device__ __forceinline__ void span_out(const EngineArrays& a, int slot, long long t0)
{
    if (!a.k1_span) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        a.k1_span[2 * slot] = (unsigned long long)t0;
        a.k1_span[2 * slot + 1] = (unsigned long long)wall_clock64();
    }
}

// Plane words held in registers: thread t owns the PER4 consecutive 16-byte groups starting at t * PER4 (no LDS copy
// of the plane at all).  Needs plane_words % 4 == 0 and plane_words / 4 <= PER4 * kFlowThreads.
template <int PER4>
__global__ __launch_bounds__(kFlowThreads) void flow_measure_kernel(EngineArrays a, double depth_max, int radius)
{
    ROFT_RESIDENT(a, RK_FLOW_MEASURE);
    __shared__ int s_wave[16];
    __shared__ uint2 s_item[kCandLds];
    const int obj = blockIdx.x, slot = blockIdx.y * a.n_obj + obj;   // slot = (frame of the batch, object)
    const FrameCtrl& c = a.ctrl[slot];
    const long long span_t0 = a.k1_span ? wall_clock64() : 0;   // (timing runs: see span_out)
#ifdef ROFT_K1_PROFILE
    ObjState& st = a.state[obj];
#endif
#ifdef ROFT_K1_PROFILE
    long long k1_t0 = wall_clock64();
#endif
    // 1. this thread's words of the previous frame's obj plane (unconditional loads, all in flight together) -- and the
    //    control block fields in the SAME round trip: inside the engine the plane's ring slot follows from the frame's
    //    position in the batch (EngineArrays::slot_prev0), not from the control block
    FlowCtrl k{c.vel_stage, c.slot_prev, c.depth_prev, c.flow[0]};
    const int n4 = (int)(a.plane_words / 4), i0 = (int)threadIdx.x * PER4;
    uint4 q[PER4];
    if (a.slot_prev0 >= 0) {
        const uint4* g4 = reinterpret_cast<const uint4*>(a.planes + plane_offset(a, obj, (a.slot_prev0 + (int)blockIdx.y) % kPlaneSlots, 1));
#pragma unroll
        for (int j = 0; j < PER4; ++j) q[j] = g4[min(i0 + j, n4 - 1)];
        asm volatile("" : "+v"(k.vel_stage), "+v"(k.slot_prev), "+v"(k.depth), "+v"(k.flow), "+v"(q[0].x));
    } else {
        asm volatile("" : "+v"(k.vel_stage), "+v"(k.slot_prev), "+v"(k.depth), "+v"(k.flow));
        const uint4* g4 = reinterpret_cast<const uint4*>(a.planes + plane_offset(a, obj, k.slot_prev, 1));
#pragma unroll
        for (int j = 0; j < PER4; ++j) q[j] = g4[min(i0 + j, n4 - 1)];
    }
    if (!k.vel_stage) {
        if (threadIdx.x == 0) a.npts[slot] = -1;
        span_out(a, slot, span_t0);
        return;
    }
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < PER4; ++j) {
        if (i0 + j >= n4) q[j] = make_uint4(0u, 0u, 0u, 0u);
        cnt += __popc(q[j].x) + __popc(q[j].y) + __popc(q[j].z) + __popc(q[j].w);
    }
    K1TICK(1);
    // 2. starting rank of the thread's chunk (plane words are in row-major order)
    int M;
    int rank = block_exclusive_scan(cnt, s_wave, &M);
    const int C = (M + radius - 1) / radius;
    K1TICK(2);
    // 3. candidate work items of the chunk -> list, 4. gathers + compaction.  Two instances so that the common
    //    case addresses the list as LDS and not through flat pointers.
    auto tail = [&](uint2* list) {
        if (cnt) {
            int ci = (rank + radius - 1) / radius, next = ci * radius;
#pragma unroll
            for (int j = 0; j < PER4; ++j) {
                const int w = (i0 + j) * 4;
                emit_word(q[j].x, w, rank, next, ci, radius, list);
                emit_word(q[j].y, w + 1, rank, next, ci, radius, list);
                emit_word(q[j].z, w + 2, rank, next, ci, radius, list);
                emit_word(q[j].w, w + 3, rank, next, ci, radius, list);
            }
        }
        __syncthreads();   // the list is written and read by this workgroup only
        K1TICK(3);
        return gather_candidates(a, k.depth, k.flow, slot, list, C, depth_max, s_wave);
    };
    const int n = (C <= kCandLds) ? tail(s_item) : tail(reinterpret_cast<uint2*>(a.cand + (size_t)slot * a.cand_cap));
    K1TICK(5);
    if (threadIdx.x == 0) a.npts[slot] = n;
    span_out(a, slot, span_t0);
}

// Any plane size.  LDS = true: the plane is staged in dynamic LDS, every thread walks a contiguous chunk of its words.
// LDS = false (planes beyond the LDS, e.g. 1920 x 1080: 259 KB): the same two passes -- popcounts of the thread's chunk, then
// its candidate items once the block scan has given the chunk its starting rank -- read the plane from memory both times (the
// second pass finds it in the L2): no image size is refused (the reference scans any cv::Mat, hpp:231-256).
template <bool LDS>
__global__ __launch_bounds__(kFlowThreads) void flow_measure_lds_kernel(EngineArrays a, double depth_max, int radius)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_wave[16];
    __shared__ uint2 s_item[kCandLds];
    const int obj = blockIdx.x, slot = blockIdx.y * a.n_obj + obj;
    const FrameCtrl& c = a.ctrl[slot];
    if (!c.vel_stage) {
        if (threadIdx.x == 0) a.npts[slot] = -1;
        return;
    }
    const uint32_t* gplane = a.planes + plane_offset(a, obj, c.slot_prev, 1);
    uint32_t* s_plane = reinterpret_cast<uint32_t*>(smem);
    if (LDS) {
        for (size_t i = threadIdx.x; i < a.plane_words; i += blockDim.x) s_plane[i] = gplane[i];
        __syncthreads();
    }
    const uint32_t* plane = LDS ? s_plane : gplane;
    const int n_words = (int)a.plane_words;
    const int per = (n_words + blockDim.x - 1) / blockDim.x;
    const int w0 = min(n_words, (int)threadIdx.x * per), w1 = min(n_words, w0 + per);
    int cnt = 0;
    for (int w = w0; w < w1; ++w) cnt += __popc(plane[w]);
    int M;
    int rank = block_exclusive_scan(cnt, s_wave, &M);
    const int C = (M + radius - 1) / radius;
    uint2* list = (C <= kCandLds) ? s_item : reinterpret_cast<uint2*>(a.cand + (size_t)slot * a.cand_cap);
    if (cnt) {
        int ci = (rank + radius - 1) / radius, next = ci * radius;
        for (int w = w0; w < w1; ++w) emit_word(plane[w], w, rank, next, ci, radius, list);
    }
    __syncthreads();
    const int n = gather_candidates(a, c.depth_prev, c.flow[0], slot, list, C, depth_max, s_wave);
    if (threadIdx.x == 0) a.npts[slot] = n;
}

template <int PER4>
static void launch_flow_reg(const EngineArrays& a, double depth_max, int radius, hipStream_t s, hipEvent_t start,
                            hipEvent_t stop)
{
    hipExtLaunchKernelGGL(flow_measure_kernel<PER4>, dim3(a.n_obj, a.T), dim3(kFlowThreads), 0, s, start, stop, 0, a,
                          depth_max, radius);
}

void launch_flow_measure(const EngineArrays& a, double depth_max, int radius, hipStream_t s, hipEvent_t start,
                         hipEvent_t stop)
{
    const size_t n4 = a.plane_words / 4;
    const int per4 = (int)((n4 + kFlowThreads - 1) / kFlowThreads);
    if (a.plane_words % 4 == 0 && per4 <= 10) {
        if (per4 <= 1) launch_flow_reg<1>(a, depth_max, radius, s, start, stop);
        else if (per4 <= 2) launch_flow_reg<2>(a, depth_max, radius, s, start, stop);
        else if (per4 <= 3) launch_flow_reg<3>(a, depth_max, radius, s, start, stop);
        else if (per4 <= 4) launch_flow_reg<4>(a, depth_max, radius, s, start, stop);
        else if (per4 <= 6) launch_flow_reg<6>(a, depth_max, radius, s, start, stop);
        else if (per4 <= 8) launch_flow_reg<8>(a, depth_max, radius, s, start, stop);
        else launch_flow_reg<10>(a, depth_max, radius, s, start, stop);
        return;
    }
    const size_t lds = (a.plane_words * 4 + 15) & ~(size_t)15;
    const size_t lds_cap = 160 * 1024 - 256 - kCandLds * sizeof(uint2) - 128;
    if (lds <= lds_cap) {
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(flow_measure_lds_kernel<true>), (int)lds_cap);
        hipExtLaunchKernelGGL(flow_measure_lds_kernel<true>, dim3(a.n_obj, a.T), dim3(kFlowThreads), lds, s, start, stop, 0, a,
                              depth_max, radius);
    } else {
        hipExtLaunchKernelGGL(flow_measure_lds_kernel<false>, dim3(a.n_obj, a.T), dim3(kFlowThreads), 0, s, start, stop, 0, a,
                              depth_max, radius);
    }
}

// ---- records -> (uv, y, H) exactly as the reference assembles them (hpp:258-283) -------------
__global__ __launch_bounds__(256) void expand_yh_kernel(const FlowRec* recs, const int* n, DevCamera cam, double dt,
                                                        int32_t* uv, double* y, double* Hm, int cap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = min(*n, cap);
    if (i >= N) return;
    const FlowRec r = recs[i];
    uv[2 * i] = r.u;
    uv[2 * i + 1] = r.v;
    y[2 * i] = (double)r.dx;
    y[2 * i + 1] = (double)r.dy;
    const double z = (double)r.z;
    const double uu = (r.u - cam.cx);
    const double vv = (r.v - cam.cy);
    double* h = Hm + (size_t)12 * i;
    h[0] = (cam.fx / z) * dt;
    h[1] = 0.0 * dt;
    h[2] = (-uu / z) * dt;
    h[3] = (-uu * vv / cam.fy) * dt;
    h[4] = (cam.fx + uu * uu / cam.fx) * dt;
    h[5] = (-vv * cam.fx / cam.fy) * dt;
    h[6] = 0.0 * dt;
    h[7] = (cam.fy / z) * dt;
    h[8] = (-vv / z) * dt;
    h[9] = (-(cam.fy + vv * vv / cam.fy)) * dt;
    h[10] = (vv * uu / cam.fx) * dt;
    h[11] = (uu * cam.fy / cam.fx) * dt;
}

void launch_expand_yh(const FlowRec* recs, const int* n, DevCamera cam, double dt, int32_t* uv, double* y, double* H,
                      int cap, hipStream_t s)
{
    hipLaunchKernelGGL(expand_yh_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, recs, n, cam, dt, uv, y, H, cap);
}

}  // namespace roft

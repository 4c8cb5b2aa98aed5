// k_mask.hip -- segmentation mask kernels (gfx950).
//
// Reference behaviour reproduced (hsp-iit/roft v1.2.1):
//   ImageSegmentationOFAidedSource<T>::step_frame / map   include/ROFT/ImageSegmentationOFAidedSource.hpp:127-281
//   cv::remap(mask_, mask_, map, INTER_LINEAR, BORDER_CONSTANT) with an integer map (:215, :225)
//   ImageSegmentationMeasurement::freeze threshold (>1 -> 255)  src/roft-lib/src/ImageSegmentationMeasurement.cpp:65
//
// MI355X design: a mask lives in HBM as two 1-bit planes (W*H/8 bytes each instead of W*H):
//   nz  = raw value != 0  (what cv::findNonZero sees inside the OF-aided source)
//   obj = raw value  > 1  (what every consumer sees after the threshold)
// The mask of frame k is the source of frame k+1: the recursion is sequential per object and frame.  Per frame of a
// batch the chain is mask_ingest_kernel on the frames that deliver masks (u8 -> planes, counts), then ONE launch of
// mask_chain_kernel that walks the frames:
//  * binary masks (no pixel of value 1, i.e. nz == obj -- decided on the device at ingest): every source pixel carries
//    the same value, so the reference's "later writer wins" map + remap is an order-free OR of the target bits.  S
//    workgroups per object (grid S x n_obj, S * n_obj ~ the CU count) each walk a share of the source's 64-pixel groups,
//    OR into an LDS plane of their own (W*H/8 bytes, LDS atomics) and flush its non-zero words with global atomicOr into
//    the destination, which the frame before left zeroed.  No map, no gather; only the obj plane of a
//    binary mask is written and read.
//  * general masks ({0, 1, 255}): mask_general_kernel, one persistent workgroup per object at the end of the batch,
//    handles the frames whose source is not binary: the winner among the sources of a target is the one with the LARGEST
//    linear index = an atomicMax on a W*H int32 map whose zero value doubles as "unmapped -> sample mask(0,0)" exactly
//    like the zero-initialised cv::Mat map (:237); the gather reads every entry inside the targets' bounding box with
//    an atomic exchange (read + clear), so the map is never memset and never read through a stale L1 line.
// The per-frame decisions (mode, source, flow count, binary or not) are made once, by the chain kernel, and recorded in
// MaskRec rows that carry the state from frame to frame and from batch to batch.
#include <algorithm>

#include "roft_device.h"

namespace roft {

// (int)float as evaluated by the reference's x86-64 build (cvttss2si): NaN / out of range give
// INT_MIN, which then fails the `< 0` bounds test.  AMD's v_cvt_i32_f32 would saturate / give 0.
__device__ __forceinline__ int trunc_int_x86(float x)
{
    if (!(x > -2147483904.0f && x < 2147483648.0f)) return INT32_MIN;
    return (int)x;
}

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

// the same element in two steps -- load, then decode -- so that several loads can be in flight before the first use
template <int FT>
__device__ __forceinline__ uint2 flow_raw(const void* data, size_t idx)
{
    if (FT == ROFT_FLOW_S16C2) return make_uint2(reinterpret_cast<const uint32_t*>(data)[idx], 0u);
    return reinterpret_cast<const uint2*>(data)[idx];
}

template <int FT>
__device__ __forceinline__ void flow_decode(uint2 raw, float scale, float& dx, float& dy)
{
    if (FT == ROFT_FLOW_S16C2) {
        dx = (float)(short)(raw.x & 0xFFFFu) / scale;
        dy = (float)(short)(raw.x >> 16) / scale;
    } else {
        dx = __uint_as_float(raw.x) / scale;
        dy = __uint_as_float(raw.y) / scale;
    }
}

// ---- ingest: raw u8 mask -> (nz, obj) bit planes ---------------------------------------------------
// One thread converts 64 consecutive pixels: four 16-byte loads, two 64-bit masks built in registers,
// two coalesced 8-byte stores.
__device__ __forceinline__ void bytes_to_bits(uint32_t w, int shift, unsigned long long& nz, unsigned long long& ob)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t v = (w >> (8 * k)) & 0xFFu;
        nz |= (unsigned long long)(v != 0u) << (shift + k);
        ob |= (unsigned long long)(v > 1u) << (shift + k);
    }
}

__device__ __forceinline__ void ingest_group(const uint4* src, int g, uint2* nz, uint2* ob, int& count, int& ones)
{
    unsigned long long bnz = 0, bob = 0;
    uint4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = src[(size_t)g * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bytes_to_bits(v[q].x, 16 * q, bnz, bob);
        bytes_to_bits(v[q].y, 16 * q + 4, bnz, bob);
        bytes_to_bits(v[q].z, 16 * q + 8, bnz, bob);
        bytes_to_bits(v[q].w, 16 * q + 12, bnz, bob);
    }
    nz[g] = make_uint2((uint32_t)bnz, (uint32_t)(bnz >> 32));
    ob[g] = make_uint2((uint32_t)bob, (uint32_t)(bob >> 32));
    count += __popcll(bnz);
    ones += __popcll(bnz & ~bob);
}

// grid: (ceil(W*H/64/256), n_obj); frame t of the batch
__global__ __launch_bounds__(256) void mask_ingest_kernel(EngineArrays a, int t)
{
    const int obj = blockIdx.y;
    const FrameCtrl& c = frame_ctrl(a, t, obj);
    if (!c.has_new_mask) return;
    const int n_grp = (a.cam.W * a.cam.H) >> 6;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    int count = 0, ones = 0;
    if (g < n_grp)
        ingest_group(reinterpret_cast<const uint4*>(c.new_mask), g,
                     reinterpret_cast<uint2*>(a.planes + plane_offset(a, obj, a.slot_new + t, 0)),
                     reinterpret_cast<uint2*>(a.planes + plane_offset(a, obj, a.slot_new + t, 1)), count, ones);
    for (int off = 32; off > 0; off >>= 1) { count += __shfl_xor(count, off, 64); ones += __shfl_xor(ones, off, 64); }
    if ((threadIdx.x & 63) == 0 && count) {
        MaskRec& r = a.mrec[(size_t)(t + 1) * a.n_obj + obj];
        atomicAdd(&r.new_count, count);
        if (ones) atomicAdd(&r.new_ones, ones);
    }
}

void launch_mask_ingest(const EngineArrays& a, int t, hipStream_t s, hipEvent_t stop)
{
    const int n_grp = a.cam.W * a.cam.H / 64;
    hipExtLaunchKernelGGL(mask_ingest_kernel, dim3((n_grp + 255) / 256, a.n_obj), dim3(256), 0, s, nullptr, stop, 0, a, t);
}

__global__ void mask_reset_kernel(EngineArrays a)
{
    mask_reset_tables(a, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

void launch_mask_reset(const EngineArrays& a, hipStream_t s)
{
    hipLaunchKernelGGL(mask_reset_kernel, dim3(((a.T + 1) * a.n_obj + 255) / 256), dim3(256), 0, s, a);
}

// ---- mask chain ------------------------------------------------------------------------------------
constexpr int kMaskThreads = 1024;
constexpr int kMaskWaves = kMaskThreads / 64;
// (64-pixel groups whose walks through the flows are in flight together in one wave -- chase_groups' NCH: a pixel's
//  walk is a chain of dependent loads, the chains of different groups are independent)

struct MaskShared {
    int bbox[4];
    int n_list;
    const void* flows[kMaxFlowHist];
};

// Geometry the walks need (a handful of scalars instead of the whole EngineArrays block)
struct ChaseGeo {
    int W, H, cols, grid;
    float grid_f, scale;
    float inv_grid, inv_scale;   // exact reciprocals when grid and scale are powers of two (they are: 1 | 4, 1 | 32)
    int mode;                    // 2: grid 1 and scale 1; 1: multiply by the reciprocals (bit-identical to the divisions); 0: divide
};

__host__ __device__ inline ChaseGeo make_chase_geo(const DevCamera& cam, const DevFlowFmt& f)
{
    ChaseGeo g;
    g.W = cam.W; g.H = cam.H; g.cols = f.cols; g.grid = f.grid;
    g.grid_f = (float)f.grid; g.scale = f.scale;
    g.inv_grid = 1.0f / g.grid_f; g.inv_scale = 1.0f / f.scale;
    int e = 0;
    const bool p2g = (f.grid & (f.grid - 1)) == 0;
    const bool p2s = f.scale > 0.0f && frexpf(f.scale, &e) == 0.5f;
    g.mode = (p2g && p2s) ? ((f.grid == 1 && f.scale == 1.0f) ? 2 : 1) : 0;
    return g;
}

// (int)x of the reference's x86-64 build as far as the bounds tests and indices can tell: NaN and |x| >= 2^31 give
// INT_MIN there (cvttss2si), i.e. "out of the image"; here NaN -> -2 and the value is clamped to [-2, 2^24] before the
// conversion, which is out of the image as well (W*H < 2^24) and exact in between, (-1, 0) -> 0 included.
// (v_med3_f32 returns min3 of its operands when one of them is a NaN, and v_min_f32 returns the other operand: a NaN
//  becomes -2 without a separate test; the NaN-flow cases of tests/test_parity_gpu.py pin this.)
__device__ __forceinline__ int trunc_clamped(float x)
{
    return (int)__builtin_amdgcn_fmed3f(x, -2.0f, 16777216.0f);
}

#define ROFT_GLOBAL __attribute__((address_space(1)))
// Walks of the source pixels of one frame.  `list` (LDS) holds the non-empty 64-pixel groups of the source plane as
// (row << 16 | column) of their first pixel -- the division by the image width is done once per group by the list
// pass, one group per thread, instead of by every wave that walks the group; wave w owns entries w, w + 16, ... (the object's rows spread over all waves), prefetches the plane
// words of up to 64 of them with one load (lane i <-> the wave's i-th entry) and chases them NCH at a time: the
// flow reads of a wave are row-contiguous (64 x 8 B).  A surviving pixel is handed to `hit(target, x, y, source)`.
// One CU walks a whole object, so the instruction count per pixel and flow matters as much as the load latency:
//  * per-group work (row / column of the group) is wave-uniform;
//  * a pixel that is not set, or left the image, carries t_x = NaN from then on -- NaN survives every flow addition and
//    converts to "out of the image", so there is no per-walk activity flag to keep (and a NaN flow drops the pixel the
//    same way, as cvttss2si does in the reference);
//  * the float -> int conversions are clamps; MODE 2: grid 1 and scale 1 (CV_32FC2), the flow element of a pixel is
//    the pixel itself; MODE 1: grid and scale are powers of two, the divisions are exact reciprocal multiplies;
//    MODE 0: true divisions.
template <int FT, int NCH, int MODE, class Hit>
__device__ __forceinline__ void chase_groups(const ChaseGeo g, const uint2* plane2_, const uint32_t* list, int n_list,
                                             int n_flows, bool clear00, const void* const* flows, Hit hit,
                                             const uint2* words = nullptr)
{
    const int W = g.W, H = g.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const ROFT_GLOBAL uint2* plane2 = (const ROFT_GLOBAL uint2*)plane2_;
    const float nan = __uint_as_float(0x7FC00000u);
    for (int e0 = wave; e0 < n_list; e0 += kMaskWaves * 64) {
        const int my_e = e0 + lane * kMaskWaves;
        uint32_t my_yx = 0u;
        uint2 mine = make_uint2(0u, 0u);
        if (my_e < n_list) {
            my_yx = list[my_e];
            if (words) {   // (LDS copy kept by the list pass: no second trip to memory)
                mine = words[my_e];
            } else {
                const int my_grp = (int)(((my_yx >> 16) * (uint32_t)W + (my_yx & 0xFFFFu)) >> 6);
                // (agent-coherent: inside the chain kernel the word may come from a workgroup on another XCD)
                const unsigned long long w = __hip_atomic_load((const ROFT_GLOBAL unsigned long long*)(plane2 + my_grp),
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mine = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
            }
        }
        unsigned long long pending = __ballot(my_e < n_list);
        while (pending) {
            float t_x[NCH], t_y[NCH];
            int src[NCH];
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                t_x[u] = nan;
                t_y[u] = 0.0f;
                src[u] = 0;
                if (pending) {
                    const int it = __builtin_ctzll(pending);
                    pending &= pending - 1;
                    const uint32_t yx = (uint32_t)__builtin_amdgcn_readlane((int)my_yx, it);
                    unsigned long long bits = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine.y, it) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)mine.x, it);
                    if (clear00 && yx == 0u) bits &= ~1ull;                 // mask_.at<uchar>(0,0) = 0
                    // row / column of the group's first pixel: wave-uniform; a group may straddle rows when W % 64 != 0
                    const int y0 = (int)(yx >> 16), x0 = (int)(yx & 0xFFFFu);
                    int px = x0 + lane, py = y0;
                    if (W & 63) {
                        if (px >= W) { px -= W; ++py; }
                        if (px >= W) { px -= W; ++py; }
                    }
                    src[u] = y0 * W + x0 + lane;
                    t_x[u] = ((bits >> lane) & 1ull) ? (float)px : nan;
                    t_y[u] = (float)py;
                }
            }
            // flows in chronological order: oldest buffered first (flows[n_flows-1]) ... current (flows[0])
            for (int j = n_flows - 1; j >= 0; --j) {
                // (wave-uniform base pointer in scalar registers: the loads need only a 32-bit offset per lane)
                const unsigned long long fl_bits = (unsigned long long)flows[j];
                const ROFT_GLOBAL unsigned char* fl = (const ROFT_GLOBAL unsigned char*)(
                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(fl_bits >> 32)) << 32) |
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)fl_bits));
                uint2 raw[NCH];
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int ix = trunc_clamped(t_x[u]), iy = trunc_clamped(t_y[u]);
                    const bool in = (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H;
                    t_x[u] = in ? t_x[u] : nan;   // left the image: the pixel is dropped (hpp:262-266)
                    int fr, fc;
                    if (MODE == 2) { fr = iy; fc = ix; }
                    else if (MODE == 1) { fr = trunc_clamped(t_y[u] * g.inv_grid); fc = trunc_clamped(t_x[u] * g.inv_grid); }
                    else { fr = trunc_clamped(t_y[u] / g.grid_f); fc = trunc_clamped(t_x[u] / g.grid_f); }
                    // dropped pixels read element (0, 0): the loads stay unconditional and in flight together
                    const uint32_t off = in ? (uint32_t)(fr * g.cols + fc) * (FT == ROFT_FLOW_S16C2 ? 4u : 8u) : 0u;
                    if (FT == ROFT_FLOW_S16C2) {
                        raw[u] = make_uint2(*(const ROFT_GLOBAL uint32_t*)(fl + off), 0u);
                    } else {
                        const unsigned long long w = *(const ROFT_GLOBAL unsigned long long*)(fl + off);
                        raw[u] = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
                    }
                }
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    float dx, dy;
                    if (FT == ROFT_FLOW_S16C2) { dx = (float)(short)(raw[u].x & 0xFFFFu); dy = (float)(short)(raw[u].x >> 16); }
                    else { dx = __uint_as_float(raw[u].x); dy = __uint_as_float(raw[u].y); }
                    if (MODE == 1) { dx *= g.inv_scale; dy *= g.inv_scale; }
                    else if (MODE == 0) { dx /= g.scale; dy /= g.scale; }
                    t_x[u] += dx;
                    t_y[u] += dy;
                }
            }
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                const int ix = trunc_clamped(t_x[u]), iy = trunc_clamped(t_y[u]);
                if ((unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H) hit(iy * W + ix, ix, iy, src[u]);
            }
        }
    }
}

// One flow (every frame between two mask deliveries: mode 1): the walk of a pixel is a single step from an integer
// position, so nothing but the loaded flow elements has to stay in registers while the loads are in flight -- two
// VGPRs per group -- and ALL groups of a wave (about a dozen at 64 objects) go out in one round: the frame pays one
// memory latency for its flow.  Position, bit and target are (re)computed when the data is back.  Same arithmetic as
// chase_groups with n_flows == 1, operation by operation.  Needs the plane words of the listed groups in LDS (`words`).
constexpr int kSingleWalks = 12;   // groups per wave whose flow loads are in flight together (16: slower, register pressure)
template <int FT, int MODE>
__device__ __forceinline__ void walk_single(const ChaseGeo g, const uint32_t* list_, const uint2* words_, int n_list, bool clear00,
                                            const void* flow, ROFT_LDS uint32_t* tgt)
{
    constexpr int NCH = kSingleWalks;
    // (LDS pointers as such: through generic pointers every read of the list is a flat load, and a flat load waits for
    //  ALL outstanding memory operations -- the flow loads would go out one at a time)
    const ROFT_LDS uint32_t* const list = (const ROFT_LDS uint32_t*)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(uint32_t)(uintptr_t)(const ROFT_LDS uint32_t*)list_);
    const ROFT_LDS uint32_t* const words = (const ROFT_LDS uint32_t*)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(uint32_t)(uintptr_t)(const ROFT_LDS uint32_t*)reinterpret_cast<const uint32_t*>(words_));
    const int W = g.W, H = g.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long fl_bits = (unsigned long long)flow;
    const ROFT_GLOBAL unsigned char* fl = (const ROFT_GLOBAL unsigned char*)(
        ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(fl_bits >> 32)) << 32) |
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)fl_bits));
    for (int e0 = wave; e0 < n_list; e0 += NCH * kMaskWaves) {
        uint2 raw[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int e = e0 + u * kMaskWaves;
            raw[u] = make_uint2(0u, 0u);
            if (e < n_list) {   // (wave-uniform)
                const uint32_t yx = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[e]);
                int px = (int)(yx & 0xFFFFu) + lane, py = (int)(yx >> 16);
                if (W & 63) {
                    if (px >= W) { px -= W; ++py; }
                    if (px >= W) { px -= W; ++py; }
                }
                int fr, fc;
                if (MODE == 2) { fr = py; fc = px; }
                else { fr = trunc_clamped((float)py * g.inv_grid); fc = trunc_clamped((float)px * g.inv_grid); }
                const uint32_t off = (uint32_t)(fr * g.cols + fc) * (FT == ROFT_FLOW_S16C2 ? 4u : 8u);
                if (FT == ROFT_FLOW_S16C2) {
                    raw[u] = make_uint2(*(const ROFT_GLOBAL uint32_t*)(fl + off), 0u);
                } else {
                    const unsigned long long w = *(const ROFT_GLOBAL unsigned long long*)(fl + off);
                    raw[u] = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int e = e0 + u * kMaskWaves;
            if (e < n_list) {
                const uint32_t yx = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[e]);
                unsigned long long bits = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)words[2 * e + 1]) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)words[2 * e]);
                if (clear00 && yx == 0u) bits &= ~1ull;                 // mask_.at<uchar>(0,0) = 0
                int px = (int)(yx & 0xFFFFu) + lane, py = (int)(yx >> 16);
                if (W & 63) {
                    if (px >= W) { px -= W; ++py; }
                    if (px >= W) { px -= W; ++py; }
                }
                float dx, dy;
                if (FT == ROFT_FLOW_S16C2) { dx = (float)(short)(raw[u].x & 0xFFFFu); dy = (float)(short)(raw[u].x >> 16); }
                else { dx = __uint_as_float(raw[u].x); dy = __uint_as_float(raw[u].y); }
                if (MODE == 1) { dx *= g.inv_scale; dy *= g.inv_scale; }
                const float t_x = (float)px + dx, t_y = (float)py + dy;
                const int ix = trunc_clamped(t_x), iy = trunc_clamped(t_y);
                if (((bits >> lane) & 1ull) && (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H) {
                    const int tp = iy * W + ix;
                    (void)__hip_atomic_fetch_or(tgt + (tp >> 5), 1u << (tp & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
}

// binary source: OR-scatter into the LDS plane.  kBinaryWalks walks in flight per wave: a workgroup's share of an object
// is about a dozen groups per wave, so all their flow reads go out together and the frame pays ONE memory latency per
// flow instead of one per eight groups.
constexpr int kBinaryWalks = 8;
template <int FT>
__device__ __noinline__ void propagate_binary(ChaseGeo g, const uint2* plane2, const uint32_t* list, int n_list, int n_flows,
                                              bool clear00, const void* const* flows, uint32_t* s_tgt, const uint2* words)
{
    // (workgroup-uniform LDS address into a scalar register: arguments of a function arrive in vector registers)
    ROFT_LDS uint32_t* const tgt = (ROFT_LDS uint32_t*)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(uint32_t)(uintptr_t)(ROFT_LDS uint32_t*)s_tgt);
    if (n_flows == 1 && words && g.mode != 0) {
        if (g.mode == 2) walk_single<FT, 2>(g, list, words, n_list, clear00, flows[0], tgt);
        else walk_single<FT, 1>(g, list, words, n_list, clear00, flows[0], tgt);
        return;
    }
    auto hit = [tgt](int tp, int, int, int) {
        (void)__hip_atomic_fetch_or(tgt + (tp >> 5), 1u << (tp & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if (g.mode == 2) chase_groups<FT, kBinaryWalks, 2>(g, plane2, list, n_list, n_flows, clear00, flows, hit, words);
    else if (g.mode == 1) chase_groups<FT, kBinaryWalks, 1>(g, plane2, list, n_list, n_flows, clear00, flows, hit, words);
    else chase_groups<FT, 4, 0>(g, plane2, list, n_list, n_flows, clear00, flows, hit, words);
}

// general source: map of the winning (largest) source index per target + the targets' bounding box (LDS bbox[4])
template <int FT>
__device__ __noinline__ void propagate_general(ChaseGeo g, const uint2* plane2, const uint32_t* list, int n_list, int n_flows,
                                               bool clear00, const void* const* flows, int32_t* map, int* s_bbox)
{
    int bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = -1, by1 = -1;
    auto hit = [map, &bx0, &by0, &bx1, &by1](int tp, int ix, int iy, int p) {
        atomicMax(&map[tp], p);
        bx0 = min(bx0, ix); bx1 = max(bx1, ix);
        by0 = min(by0, iy); by1 = max(by1, iy);
    };
    if (g.mode == 2) chase_groups<FT, 4, 2>(g, plane2, list, n_list, n_flows, clear00, flows, hit);
    else if (g.mode == 1) chase_groups<FT, 4, 1>(g, plane2, list, n_list, n_flows, clear00, flows, hit);
    else chase_groups<FT, 4, 0>(g, plane2, list, n_list, n_flows, clear00, flows, hit);
    for (int off = 32; off > 0; off >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, off, 64)); by0 = min(by0, __shfl_xor(by0, off, 64));
        bx1 = max(bx1, __shfl_xor(bx1, off, 64)); by1 = max(by1, __shfl_xor(by1, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && bx1 >= 0) {
        atomicMin(&s_bbox[0], bx0); atomicMin(&s_bbox[1], by0);
        atomicMax(&s_bbox[2], bx1); atomicMax(&s_bbox[3], by1);
    }
}

// copies / fills of plane words by the whole workgroup: 16-byte accesses when pointers and length allow, 8-byte ones
// otherwise (plane offsets and shares are multiples of two words: W*H % 64 == 0)
__device__ __forceinline__ void plane_copy(uint32_t* d, const uint32_t* s, size_t n_words)
{
    if (((n_words & 3) | ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15)) == 0)
        for (size_t i = threadIdx.x; i < n_words / 4; i += blockDim.x) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    else
        for (size_t i = threadIdx.x; i < n_words / 2; i += blockDim.x) reinterpret_cast<uint2*>(d)[i] = reinterpret_cast<const uint2*>(s)[i];
}

__device__ __forceinline__ void plane_fill(uint32_t* d, uint32_t v, size_t n_words)
{
    if (((n_words & 3) | (reinterpret_cast<uintptr_t>(d) & 15)) == 0)
        for (size_t i = threadIdx.x; i < n_words / 4; i += blockDim.x) reinterpret_cast<uint4*>(d)[i] = make_uint4(v, v, v, v);
    else
        for (size_t i = threadIdx.x; i < n_words / 2; i += blockDim.x) reinterpret_cast<uint2*>(d)[i] = make_uint2(v, v);
}

// The same through agent-coherent accesses (sc1: write-through stores, loads that miss the caches above the coherence
// point), 8 bytes at a time: what the workgroups of an object, spread over XCDs with an L2 each, exchange INSIDE the
// persistent chain kernel goes through these and through atomics only -- no cache write-back / invalidate per frame.
__device__ __forceinline__ unsigned long long coh_load64(const void* p)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void coh_store64(void* p, unsigned long long v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void plane_copy_coherent(uint32_t* d, const uint32_t* s, size_t n_words)
{
    for (size_t i = threadIdx.x; i < n_words / 2; i += blockDim.x) coh_store64(d + 2 * i, coh_load64(s + 2 * i));
}

__device__ __forceinline__ void plane_fill_coherent(uint32_t* d, uint32_t v, size_t n_words)
{
    const unsigned long long vv = ((unsigned long long)v << 32) | v;
    for (size_t i = threadIdx.x; i < n_words / 2; i += blockDim.x) coh_store64(d + 2 * i, vv);
}

// The decisions of frame t (ImageSegmentationOFAidedSource::step_frame, hpp:169-226) from the state after frame t-1.
__device__ __forceinline__ MaskRec decide_frame(const MaskRec& prev, const MaskRec& cur, int new_slot, const FrameCtrl& c,
                                                int frames_between, int flow_aided)
{
    MaskRec r = cur;   // new_count / new_ones of this frame
    const int new_count = c.has_new_mask ? r.new_count : 0;
    const int new_binary = r.new_ones == 0;
    if (flow_aided) {
        r.mode = decide_mode(c, new_slot, prev.fbuf_n, new_count, frames_between, r.src_slot, r.n_flows);
        r.fbuf_n = next_fbuf(c, prev.fbuf_n, new_count, r.mode, frames_between);
    } else {  // no flow-aided segmentation: the delivered mask is used as is, otherwise the last one persists
        r.mode = 0;
        r.src_slot = c.has_new_mask ? new_slot : c.slot_prev;
        r.n_flows = 0;
        r.fbuf_n = 0;
    }
    r.src_binary = (r.src_slot >= kSlotNew) ? new_binary : prev.binary;
    r.binary = r.src_binary;
    return r;
}

// Barrier among the nq workgroups of one object inside the persistent chain kernel: `counter` (zeroed by the control block
// upload of the batch) counts arrivals, `target` = nq * (barriers so far + 1).  What the workgroups exchange goes through
// agent-coherent accesses (sc1 stores / loads, device-scope atomics): a thread only has to wait for its own stores and
// atomics to be acknowledged before the workgroup arrives.  In two halves, so that what a workgroup can do for the next
// frame without the others' results -- control block, decisions, zeroing -- runs while the arrivals travel: arrive (one
// atomic by thread 0) ... wait (thread 0 polls).
// FORWARD PROGRESS: the poll ends only if the object's other workgroups run, i.e. the nq * n_obj workgroups of the launch
// must become resident together.  HIP promises no dispatch order; what the launch relies on is (i) nq * n_obj <= the CUs
// the launch may fill (launch_mask_chain: three quarters of the device's CUs, one workgroup fills a CU's register file;
// nq = 1, no barrier at all, when that cannot hold), (ii) the hardware dispatching the workgroups of ONE launch in
// order and (iii) no kernel of the other chains ever waiting for this one while it holds CUs.  Several engines -- or
// processes -- running mask chains on one device at once can break (i); therefore the poll is bounded: after ~2 s the
// workgroup raises EngineArrays::dev_error (pinned host memory: the host turns it into ROFT_ERR_DEVICE at its next
// synchronisation, roft_engine.h) and leaves the kernel, and so does every workgroup that sees the flag raised.
__device__ __forceinline__ void object_arrive(unsigned* counter)
{
    __builtin_amdgcn_s_waitcnt(0);   // every store / atomic of this thread acknowledged
    __syncthreads();
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// returns false when the barrier was abandoned (dev_error raised)
__device__ __forceinline__ bool object_wait(unsigned* counter, unsigned target, int* dev_error, int* s_ok)
{
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0u) {   // ~ every 50 us: somebody else gave up, or two seconds have passed (100 MHz clock)
                const bool raised = dev_error && __hip_atomic_load(dev_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
                if (raised || wall_clock64() - t0 > 200000000ll) {
                    if (dev_error && !raised) __hip_atomic_store(dev_error, ROFT_DEV_ERROR_MASK_BARRIER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ok = 0;
                    break;
                }
            }
        }
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

// The binary-mask chain of a batch: ONE launch walks the T frames.  grid: (S, n_obj).  Workgroup q of an object owns
// the 64-pixel groups q, q + S, ... of the source and the words [q, q+1) * plane_words / S of the planes it copies /
// fills / zeroes; the S workgroups of an object meet at a barrier in memory (object_arrive / object_wait) between two frames (the mask of frame t is
// the source of frame t + 1), objects never wait for each other.
// dynamic LDS: [plane_words] OR target | list of this workgroup's non-empty groups [| their plane words]
#ifdef ROFT_MASK_PROFILE
#define MTICK(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 4) { long long _t = wall_clock64(); a.state[blockIdx.y].dbg[blockIdx.x * 8 + (i)] += _t - m_t0; m_t0 = _t; } } while (0)
#else
#define MTICK(i) do {} while (0)
#endif

template <int FT>
__global__ __launch_bounds__(kMaskThreads) void mask_chain_kernel(EngineArrays a, int frames_between, int flow_aided,
                                                                  int list_cap, int keep_words)
{
#ifdef ROFT_MASK_PROFILE
    long long m_t0 = wall_clock64();
    if (threadIdx.x < 8 && blockIdx.x < 4) a.state[blockIdx.y].dbg[blockIdx.x * 8 + threadIdx.x] = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ MaskShared S;
    uint32_t* s_tgt = reinterpret_cast<uint32_t*>(smem);
    uint32_t* s_list = reinterpret_cast<uint32_t*>(smem + ((a.plane_words * 4 + 15) & ~(size_t)15));
    const int obj = blockIdx.y, q = blockIdx.x, nq = gridDim.x;
    // plane words of the listed groups (behind the list, when the launch reserved the room)
    uint2* s_words = keep_words ? reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(s_list) + (((size_t)list_cap * 4 + 15) & ~(size_t)15))
                                : nullptr;
    const int W = a.cam.W, H = a.cam.H, n_grp = (W * H) >> 6;
    const int tid = threadIdx.x, lane = tid & 63;
    __shared__ FrameCtrl s_c;
    __shared__ int s_barrier_ok;
    __shared__ MaskRec s_rec[2];   // [0] state after the frame before (frame 0: the carry), [1] this frame's counters
    static_assert(sizeof(MaskRec) == 32, "two 16-byte loads per record");
    // share of the plane words of this workgroup, in 16-byte units when the planes are 16-byte aligned
    const int unit = (a.plane_words & 3) ? 2 : 4;
    const int n_units = (int)(a.plane_words / unit);
    const int u0 = (int)((long long)n_units * q / nq) * unit, u1 = (int)((long long)n_units * (q + 1) / nq) * unit;
    const ChaseGeo geo = make_chase_geo(a.cam, a.ffmt);
    unsigned n_barriers = 0, general = 0u;
    // What a frame needs before it can read its source: control block and state records -> LDS with one load per thread,
    // the decisions, the obj plane of the NEXT frame's slot zeroed for that frame's OR flush (nobody reads that slot
    // any more: its last user is kPlaneSlots frames back), the LDS plane zeroed.  Runs between the arrival at the
    // barrier behind the frame before and the wait for the others.
    auto prologue = [&](int t) -> MaskRec {
        stage_ctrl(&s_c, frame_ctrl(a, t, obj));
        if (tid >= 128 && tid < 132) {
            const int k = tid - 128;   // 0, 1: the carry (first frame only; later the record of the frame before); 2, 3: this frame's
            if (k >= 2)
                reinterpret_cast<uint4*>(s_rec)[k] = reinterpret_cast<const uint4*>(a.mrec + (size_t)(t + 1) * a.n_obj + obj)[k & 1];
            else if (t == 0)
                reinterpret_cast<uint4*>(s_rec)[k] = reinterpret_cast<const uint4*>(a.mrec_carry + obj)[k & 1];
        }
        plane_fill(s_tgt, 0u, a.plane_words);
        __syncthreads();
        const MaskRec r = decide_frame(s_rec[0], s_rec[1], a.slot_new + t, s_c, frames_between, flow_aided);
        MTICK(0);
        if (q == 0 && tid == 0) a.mrec[(size_t)(t + 1) * a.n_obj + obj] = r;   // (every workgroup computes the same record)
        plane_fill_coherent(a.planes + plane_offset(a, obj, (s_c.slot_cur + 1) % kPlaneSlots, 1) + u0, 0u, (size_t)(u1 - u0));
        if (tid < kMaxFlowHist) S.flows[tid] = (tid < r.n_flows) ? s_c.flow[tid] : nullptr;
        if (tid == 0) S.n_list = 0;
        __syncthreads();
        MTICK(1);
        return r;
    };
    MaskRec r = prologue(0);
    for (int t = 0; t < a.T; ++t) {
        const FrameCtrl& c = s_c;
        const uint32_t* src = a.planes + plane_offset(a, obj, r.src_slot, 1);
        uint32_t* dst = a.planes + plane_offset(a, obj, c.slot_cur, 1);
        if (!r.src_binary) {
            general |= 1u << t;   // three-valued source: mask_general_kernel
        } else if (r.mode == 0) {
            plane_copy_coherent(dst + u0, src + u0, (size_t)(u1 - u0));
        } else if (r.mode == 2 && (src[0] & 1u)) {
            // mask(0,0) set in a new-mask frame: every target, mapped or not, samples a set pixel
            plane_fill_coherent(dst + u0, ~0u, (size_t)(u1 - u0));
        } else {
            // This workgroup's non-empty 64-pixel groups of the source (g = q + nq i) -> list (any order: the scatter
            // is order-free), then their walks; in chunks of list_cap groups when the LDS next to the plane cannot list
            // the whole share at once (a 1280x720 plane with one workgroup per object).
            const uint2* plane2 = reinterpret_cast<const uint2*>(src);
            const int share = (n_grp - q + nq - 1) / nq;
            for (int c0 = 0; c0 < share; c0 += list_cap) {
                if (c0 > 0) {
                    __syncthreads();   // the walks of the chunk before have read the list
                    if (tid == 0) S.n_list = 0;
                    __syncthreads();
                }
                const int c1 = min(share, c0 + list_cap);
                for (int i0 = c0; i0 < c1; i0 += kMaskThreads) {
                    const int i = i0 + tid, g = q + nq * i;
                    bool ne = false;
                    uint2 w = make_uint2(0u, 0u);
                    if (i < c1) {
                        const unsigned long long ww = coh_load64(plane2 + g);
                        w = make_uint2((uint32_t)ww, (uint32_t)(ww >> 32));
                        ne = ww != 0ull;
                    }
                    const unsigned long long b = __ballot(ne);
                    int base = 0;
                    if (lane == 0 && b) base = atomicAdd(&S.n_list, __popcll(b));
                    base = __shfl(base, 0, 64);
                    if (ne) {
                        const int e = base + __popcll(b & ((1ull << lane) - 1ull));
                        const int p0 = g * 64, y0 = p0 / W;
                        s_list[e] = ((uint32_t)y0 << 16) | (uint32_t)(p0 - y0 * W);
                        if (s_words) s_words[e] = w;
                    }
                }
                __syncthreads();
                MTICK(2);
                propagate_binary<FT>(geo, plane2, s_list, S.n_list, r.n_flows, r.mode == 1, S.flows, s_tgt, s_words);
            }
            __syncthreads();
            MTICK(3);
            // flush: the non-zero words of this workgroup's plane into the (zeroed) destination
            for (int i = tid; i < (int)a.plane_words; i += kMaskThreads) {
                const uint32_t v = s_tgt[i];
                if (v) atomicOr(&dst[i], v);
            }
            MTICK(4);
        }
        if (t + 1 == a.T) {
            if (q == 0 && tid == 0) a.mask_general[obj] = general;
            break;
        }
        // the next frame reads this frame's planes and ORs into the slot zeroed by this frame's prologue
        if (nq > 1) object_arrive(a.mask_sync + obj);
        else { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); }   // (one workgroup, one CU: its L1 is written through)
        if (tid == 0) s_rec[0] = r;   // (the staging barrier of the prologue publishes it)
        r = prologue(t + 1);
        if (nq > 1 && !object_wait(a.mask_sync + obj, (unsigned)nq * ++n_barriers, a.dev_error, &s_barrier_ok)) return;
        MTICK(5);
    }
}

// One persistent workgroup per object at the end of the batch's mask chain: the frames whose source is three-valued.
// dynamic LDS: list of the non-empty groups
template <int FT>
__global__ __launch_bounds__(kMaskThreads) void mask_general_kernel(EngineArrays a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ MaskShared S;
    uint32_t* s_list = reinterpret_cast<uint32_t*>(smem);
    const int obj = blockIdx.x;
    const int W = a.cam.W, H = a.cam.H, npix = W * H, n_grp = npix >> 6;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned todo = a.mask_general[obj];   // (almost always 0: every mask the reference's sources deliver is binary)
    for (int t = 0; t < a.T; ++t) {
        if (!((todo >> t) & 1u)) continue;
        const MaskRec r = a.mrec[(size_t)(t + 1) * a.n_obj + obj];
        const FrameCtrl& c = frame_ctrl(a, t, obj);
        const uint32_t* snz = a.planes + plane_offset(a, obj, r.src_slot, 0);
        const uint32_t* sob = a.planes + plane_offset(a, obj, r.src_slot, 1);
        uint32_t* dnz = a.planes + plane_offset(a, obj, c.slot_cur, 0);
        uint32_t* dob = a.planes + plane_offset(a, obj, c.slot_cur, 1);
        if (r.mode == 0) {
            plane_copy(dnz, snz, a.plane_words);
            plane_copy(dob, sob, a.plane_words);
        } else {
            // background = mask(0,0) of the source: unmapped targets sample it; forced to 0 in mode 1
            const bool bg_nz = (r.mode == 2) && (snz[0] & 1u), bg_ob = (r.mode == 2) && (sob[0] & 1u);
            if (tid < kMaxFlowHist) S.flows[tid] = (tid < r.n_flows) ? c.flow[tid] : nullptr;
            if (tid == 0) { S.n_list = 0; S.bbox[0] = INT32_MAX; S.bbox[1] = INT32_MAX; S.bbox[2] = -1; S.bbox[3] = -1; }
            __syncthreads();
            const uint2* plane2 = reinterpret_cast<const uint2*>(snz);
            for (int g0 = 0; g0 < n_grp; g0 += kMaskThreads) {
                const int g = g0 + tid;
                bool ne = false;
                if (g < n_grp) { const uint2 w = plane2[g]; ne = (w.x | w.y) != 0u; }
                const unsigned long long b = __ballot(ne);
                int base = 0;
                if (lane == 0 && b) base = atomicAdd(&S.n_list, __popcll(b));
                base = __shfl(base, 0, 64);
                if (ne) {
                    const int p0 = g * 64, y0 = p0 / W;
                    s_list[base + __popcll(b & ((1ull << lane) - 1ull))] = ((uint32_t)y0 << 16) | (uint32_t)(p0 - y0 * W);
                }
            }
            __syncthreads();
            int32_t* map = a.map + (size_t)obj * npix;
            propagate_general<FT>(make_chase_geo(a.cam, a.ffmt), plane2, s_list, S.n_list, r.n_flows, r.mode == 1, S.flows, map, S.bbox);
            __syncthreads();
            const int bx0 = S.bbox[0], by0 = S.bbox[1], bx1 = S.bbox[2], by1 = S.bbox[3];
            // every 64-pixel output group: constant background outside the box, map samples inside
            const int wave = tid >> 6;
            for (int g = wave; g < n_grp; g += kMaskWaves) {
                const int p = g * 64 + lane;
                const int y = p / W, x = p - y * W;
                bool nzb = bg_nz, obb = bg_ob;
                if (y >= by0 && y <= by1 && x >= bx0 && x <= bx1) {
                    const int m = atomicExch(&map[p], 0);   // read + clear at the L2, never a stale L1 line
                    if (m != 0) {
                        nzb = (snz[m >> 5] >> (m & 31)) & 1u;
                        obb = (sob[m >> 5] >> (m & 31)) & 1u;
                    }
                }
                const unsigned long long b1 = __ballot(nzb), b2 = __ballot(obb);
                if (lane == 0) {
                    reinterpret_cast<uint2*>(dnz)[g] = make_uint2((uint32_t)b1, (uint32_t)(b1 >> 32));
                    reinterpret_cast<uint2*>(dob)[g] = make_uint2((uint32_t)b2, (uint32_t)(b2 >> 32));
                }
            }
        }
        __syncthreads();   // this frame's planes are the next frame's source (same workgroup)
    }
}

int launch_mask_chain(const EngineArrays& a, int frames_between, int flow_aided, hipStream_t s, hipEvent_t stop)
{
    const size_t n_grp = (size_t)a.cam.W * a.cam.H / 64;
    // Workgroups per object: the walks are latency-bound on one CU, so an object is spread over several (at most 8).  A
    // workgroup of 16 waves with 128 registers per thread fills the register file of its CU, and it stays for the whole
    // batch: a quarter of the device's CUs is left to the per-object chains of the other streams (the pose and velocity
    // filters need a nearly empty CU each) -- measured at 64 objects on 256 CUs: 3 workgroups per object +4 %
    // object-frames/s over 4, and the flow measurement's launch no longer waits for CUs.  All S * n_obj workgroups must be
    // resident together (the barrier of object_wait): S = 1, no barrier, when three quarters of the CUs cannot hold two per
    // object.  roft_config::mask_workgroups_per_object overrides the choice (clamped to what fits).
    const int cus = device_cu_count(), n_obj = a.n_obj > 0 ? a.n_obj : 1;
    const int fit = (cus - cus / 4) / n_obj;
    int S = a.mask_wgs > 0 ? std::min(a.mask_wgs, std::max(cus / n_obj, 1)) : fit;
    S = S < 1 ? 1 : (S > 8 ? 8 : S);
    const size_t lds_plane = (a.plane_words * 4 + 15) & ~(size_t)15;
    // list of a workgroup's groups next to its plane (4 B per group), their plane words behind it (8 B) if the CU's LDS
    // has the room; a share that does not fit even the list is walked in chunks
    const size_t lds_cap = 160 * 1024 - 4096;
    size_t list_cap = (n_grp + S - 1) / S;
    const int keep_words = lds_plane + ((list_cap * 4 + 15) & ~(size_t)15) + list_cap * 8 <= lds_cap ? 1 : 0;
    if (lds_plane + ((list_cap * 4 + 15) & ~(size_t)15) > lds_cap) list_cap = ((lds_cap - lds_plane) / 4) & ~(size_t)3;
    const size_t lds_step = lds_plane + ((list_cap * 4 + 15) & ~(size_t)15) + (keep_words ? list_cap * 8 : 0);
    const size_t lds_gen = (n_grp * 4 + 15) & ~(size_t)15;
    {
        const int cap = 160 * 1024 - 4096;   // (the kernel's static LDS -- control block, records, flow pointers -- is ~1.2 KB)
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(mask_chain_kernel<ROFT_FLOW_S16C2>), cap);
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(mask_chain_kernel<ROFT_FLOW_F32C2>), cap);
    }
    int launches = 0;
    const bool s16 = a.ffmt.type == ROFT_FLOW_S16C2;
    if (s16)
        hipLaunchKernelGGL(mask_chain_kernel<ROFT_FLOW_S16C2>, dim3(S, a.n_obj), dim3(kMaskThreads), (uint32_t)lds_step, s, a,
                           frames_between, flow_aided, (int)list_cap, keep_words);
    else
        hipLaunchKernelGGL(mask_chain_kernel<ROFT_FLOW_F32C2>, dim3(S, a.n_obj), dim3(kMaskThreads), (uint32_t)lds_step, s, a,
                           frames_between, flow_aided, (int)list_cap, keep_words);
    ++launches;
    if (s16)
        hipExtLaunchKernelGGL(mask_general_kernel<ROFT_FLOW_S16C2>, dim3(a.n_obj), dim3(kMaskThreads), (uint32_t)lds_gen, s, nullptr, stop, 0, a);
    else
        hipExtLaunchKernelGGL(mask_general_kernel<ROFT_FLOW_F32C2>, dim3(a.n_obj), dim3(kMaskThreads), (uint32_t)lds_gen, s, nullptr, stop, 0, a);
    return launches + 1;
}

// ---- plane -> u8 mask (operator-level output / roft_get_mask) --------------------------------
__global__ __launch_bounds__(256) void plane_to_mask_kernel(const uint32_t* nz, const uint32_t* ob, int npix,
                                                            uint8_t* mask)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const bool o = (ob[p >> 5] >> (p & 31)) & 1u;
    const bool n = nz ? ((nz[p >> 5] >> (p & 31)) & 1u) : false;
    mask[p] = o ? 255 : (n ? 1 : 0);
}

void launch_planes_to_mask(const uint32_t* nz, const uint32_t* ob, int npix, uint8_t* mask, hipStream_t s)
{
    hipLaunchKernelGGL(plane_to_mask_kernel, dim3((npix + 255) / 256), dim3(256), 0, s, nz, ob, npix, mask);
}

}  // namespace roft

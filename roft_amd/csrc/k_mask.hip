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
// batch: mask_ingest_kernel on the frames that deliver masks (u8 -> planes, counts), then ONE launch of mask_frame_kernel per
// frame:
//  * binary masks (no pixel of value 1, i.e. nz == obj -- decided on the device at ingest): every source pixel carries
//    the same value, so the reference's "later writer wins" map + remap is an order-free OR of the target bits.  Many small
//    workgroups per object (grid Q x n_obj, four waves each) walk a band of the source's 64-pixel groups, OR into an LDS
//    window around the band (LDS atomics) and flush its non-zero words with global atomicOr into the destination, which
//    the frame before left zeroed.  No map, no gather; only the obj plane of a binary mask is written and read.
//  * general masks ({0, 1, 255}): mask_general_kernel, one workgroup per object at the end of the batch,
//    handles the frames whose source is not binary: the winner among the sources of a target is the one with the LARGEST
//    linear index = an atomicMax on a W*H int32 map whose zero value doubles as "unmapped -> sample mask(0,0)" exactly
//    like the zero-initialised cv::Mat map (:237); the gather reads every entry inside the targets' bounding box with
//    an atomic exchange (read + clear), so the map is never memset and never read through a stale L1 line.
// The per-frame decisions (mode, source, flow count, binary or not) are made by the frame kernels and recorded in
// MaskRec rows that carry the state from frame to frame and from batch to batch.
#include <algorithm>
#include <cstdlib>

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
    ROFT_RESIDENT(a, RK_MASK_INGEST);
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

// Control blocks of a batch AND the ingest of the masks it delivers in ONE launch (round 6; bursts: on the mask stream the control
// block upload, the ingest and the first mask frame were three dependent launches, 27 - 35 us before the first frame could
// start).  Blocks [0, copy_blocks): the pinned staging block -> a.ctrl.  The other blocks: one 256-pixel-group chunk of one object
// of one delivering frame each; they read the two control fields they need (has_new_mask, new_mask) from the STAGING block, not
// from the device copy the first blocks are writing.  The counters the ingest adds to are zeroed by the mask chain that used
// their table last (mask_general_kernel, final launch; mask_reset_tables at allocation), not here: nothing in this kernel depends
// on anything else in it.
__global__ __launch_bounds__(256) void ctrl_ingest_kernel(const uint4* __restrict__ src, EngineArrays a, size_t n16, int copy_blocks,
                                                          int chunks, unsigned frames_packed)
{
    if ((int)blockIdx.x < copy_blocks) {
        uint4* dst = reinterpret_cast<uint4*>(a.ctrl);
        const size_t stride = (size_t)copy_blocks * blockDim.x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
        return;
    }
    ROFT_RESIDENT(a, RK_MASK_INGEST);
    const int w = (int)blockIdx.x - copy_blocks;
    const int chunk = w % chunks, rest = w / chunks, obj = rest % a.n_obj, fi = rest / a.n_obj;
    const int t = (int)((frames_packed >> (4 * fi)) & 15u);   // the fi-th delivering frame of the batch
    const FrameCtrl& c = reinterpret_cast<const FrameCtrl*>(src)[(size_t)t * a.n_obj + obj];
    if (!c.has_new_mask) return;
    const int n_grp = (a.cam.W * a.cam.H) >> 6;
    const int g = chunk * blockDim.x + threadIdx.x;
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

// returns false when the batch's delivering frames do not fit the packed argument (more than eight: never with T <= 8)
bool launch_ctrl_ingest(const void* staging, const EngineArrays& a, size_t n16, unsigned new_mask_frames, hipStream_t s, hipEvent_t stop)
{
    unsigned packed = 0;
    int n_frames = 0;
    for (int t = 0; t < a.T && t < 16; ++t)
        if (new_mask_frames & (1u << t)) {
            if (n_frames >= 8) return false;
            packed |= (unsigned)t << (4 * n_frames++);
        }
    const int n_grp = a.cam.W * a.cam.H / 64, chunks = (n_grp + 255) / 256;
    const int copy_blocks = (int)std::min<size_t>((n16 + 255) / 256, 64);
    const unsigned grid = (unsigned)copy_blocks + (unsigned)n_frames * (unsigned)a.n_obj * (unsigned)chunks;
    hipExtLaunchKernelGGL(ctrl_ingest_kernel, dim3(grid), dim3(256), 0, s, nullptr, stop, 0, reinterpret_cast<const uint4*>(staging), a, n16,
                          copy_blocks, chunks, packed);
    return true;
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

// ---- mask frames -----------------------------------------------------------------------------------
// One launch per frame of the batch (mask_frame_kernel), many small workgroups per object, NOTHING persistent: the mask
// of frame t is the source of frame t + 1, and the kernel boundary between two frames is the hand-over -- plain loads and
// stores, no barrier in memory among workgroups, no co-residency requirement, no watchdog.  (Rounds 2 - 3 walked the T
// frames of a batch in ONE persistent launch of 3 x n_obj workgroups of 16 waves that met at a barrier in memory after every
// frame: each of them filled the register file of its CU for the whole batch -- 192 of 256 CUs held by a latency-bound
// kernel, half of the chip's CU time, DESIGN.md section 5.  A workgroup here is four waves with a few KB of LDS: it fits
// next to the filters' workgroups and is gone after a few microseconds.)
#ifndef ROFT_FRAME_THREADS
#define ROFT_FRAME_THREADS 256
#endif
constexpr int kFrameThreads = ROFT_FRAME_THREADS;
constexpr int kFrameWaves = kFrameThreads / 64;
// (64-pixel groups whose walks through the flows are in flight together in one wave -- chase_groups' NCH: a pixel's
//  walk is a chain of dependent loads, the chains of different groups are independent)

struct MaskShared {
    int bbox[4];
    int n_list;
    unsigned long long w00[2];        // word 0 of the two candidate source planes (mask(0,0): hpp:214-215)
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

// Where the target bits of a workgroup go: an LDS window of plane words [off, off + words) -- the rows of the workgroup's
// source groups and a margin above and below --, and, for the few pixels that fly further, the destination plane itself.
struct OrTarget {
    ROFT_LDS uint32_t* win;     // LDS window (workgroup-uniform address)
    int off, words;             // first plane word of the window, its length
    uint32_t* dst;              // destination obj plane in HBM (zeroed by the frame before)
    __device__ __forceinline__ void hit(int tp) const
    {
        const int wi = (tp >> 5) - off;
        const uint32_t bit = 1u << (tp & 31);
        if ((unsigned)wi < (unsigned)words) (void)__hip_atomic_fetch_or(win + wi, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else (void)__hip_atomic_fetch_or(dst + (tp >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// Walks of the source pixels of one frame.  `list` (LDS) holds non-empty 64-pixel groups of the source plane as
// (row << 16 | column) of their first pixel -- the division by the image width is done once per group by the list
// pass, one group per thread, instead of by every wave that walks the group; wave w of the NW waves owns entries w, w + NW,
// ... (the object's rows spread over all waves), keeps the plane words of up to 64 of them in its lanes (lane i <-> the
// wave's i-th entry) and chases them NCH at a time: the flow reads of a wave are row-contiguous (64 x 8 B).  A surviving
// pixel is handed to `hit(target, x, y, source)`.
// The instruction count per pixel and flow matters as much as the load latency:
//  * per-group work (row / column of the group) is wave-uniform;
//  * a pixel that is not set, or left the image, carries t_x = NaN from then on -- NaN survives every flow addition and
//    converts to "out of the image", so there is no per-walk activity flag to keep (and a NaN flow drops the pixel the
//    same way, as cvttss2si does in the reference);
//  * the float -> int conversions are clamps; MODE 2: grid 1 and scale 1 (CV_32FC2), the flow element of a pixel is
//    the pixel itself; MODE 1: grid and scale are powers of two, the divisions are exact reciprocal multiplies;
//    MODE 0: true divisions.
template <int FT, int NCH, int MODE, int NW, class Hit>
__device__ __forceinline__ void chase_groups(const ChaseGeo g, const uint2* plane2_, const uint32_t* list, int n_list,
                                             int n_flows, bool clear00, const void* const* flows, Hit hit,
                                             const uint2* words = nullptr)
{
    const int W = g.W, H = g.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const ROFT_GLOBAL uint2* plane2 = (const ROFT_GLOBAL uint2*)plane2_;
    const float nan = __uint_as_float(0x7FC00000u);
    for (int e0 = wave; e0 < n_list; e0 += NW * 64) {
        const int my_e = e0 + lane * NW;
        uint32_t my_yx = 0u;
        uint2 mine = make_uint2(0u, 0u);
        if (my_e < n_list) {
            my_yx = list[my_e];
            if (words) {   // (LDS copy kept by the list pass: no second trip to memory)
                mine = words[my_e];
            } else {
                const int my_grp = (int)(((my_yx >> 16) * (uint32_t)W + (my_yx & 0xFFFFu)) >> 6);
                const unsigned long long w = *(const ROFT_GLOBAL unsigned long long*)(plane2 + my_grp);
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
// VGPRs per group -- and ALL groups of a wave (about a dozen) go out in one round: the frame pays one
// memory latency for its flow.  Position, bit and target are (re)computed when the data is back.  Same arithmetic as
// chase_groups with n_flows == 1, operation by operation.  Needs the plane words of the listed groups in LDS (`words`).
#ifndef ROFT_SINGLE_WALKS
#define ROFT_SINGLE_WALKS 12
#endif
constexpr int kSingleWalks = ROFT_SINGLE_WALKS;   // groups per wave whose flow loads are in flight together
template <int FT, int MODE, int NW>
__device__ __forceinline__ void walk_single(const ChaseGeo g, const uint32_t* list_, const uint2* words_, int n_list, bool clear00,
                                            const void* flow, const OrTarget tgt)
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
    for (int e0 = wave; e0 < n_list; e0 += NCH * NW) {
        uint2 raw[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int e = e0 + u * NW;
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
            const int e = e0 + u * NW;
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
                if (((bits >> lane) & 1ull) && (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H) tgt.hit(iy * W + ix);
            }
        }
    }
}

// The same walk for image widths that are multiples of 64 (a 64-pixel group never straddles two rows) with the bookkeeping of
// a group on the SCALAR unit.  walk_single spends ~100 vector instructions per group -- list entry and plane words fetched from
// LDS and broadcast group by group, the pixel's row and column, the element's offset and the 64-bit shift that isolates the
// lane's bit recomputed per lane -- and the walk of a frame is bound by exactly that: stamps inside the kernel put the
// processing of a round of twelve groups at 2.8 us against 1.8 us for the round trip that fetched their flow.  Here a wave
// reads its (up to 64) list entries and plane words ONCE, one entry per lane, and takes each group's values with v_readlane:
// row, column, element offset, the clearing of mask(0,0) and the row / column as floats are wave-uniform; what is left per lane
// is the load, two float additions, the two clamped conversions, the bounds and bit tests and the OR.  Same arithmetic on the same
// values, operation by operation: (float)(x0 + lane) = (float)x0 + (float)lane exactly (integers below 2^24).
template <int FT, int MODE, int NW>
__device__ __forceinline__ void walk_single_aligned(const ChaseGeo g, const uint32_t* list_, const uint2* words_, int n_list, bool clear00,
                                                    const void* flow, const OrTarget tgt)
{
    static_assert(MODE == 1 || MODE == 2, "power-of-two grid and scale");
    constexpr int NCH = kSingleWalks;
    const ROFT_LDS uint32_t* const list = (const ROFT_LDS uint32_t*)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(uint32_t)(uintptr_t)(const ROFT_LDS uint32_t*)list_);
    const ROFT_LDS uint32_t* const words = (const ROFT_LDS uint32_t*)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(uint32_t)(uintptr_t)(const ROFT_LDS uint32_t*)reinterpret_cast<const uint32_t*>(words_));
    const int W = g.W, H = g.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr uint32_t elem = (FT == ROFT_FLOW_S16C2) ? 4u : 8u;
    const int sh = (MODE == 2) ? 0 : __builtin_ctz((unsigned)g.grid);
    const unsigned long long fl_bits = (unsigned long long)flow;
    const ROFT_GLOBAL unsigned char* const fl = (const ROFT_GLOBAL unsigned char*)(
        ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(fl_bits >> 32)) << 32) |
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)fl_bits));
    // per-lane constants
    const uint32_t lane_off = (uint32_t)(lane >> sh) * elem;          // the lane's flow element relative to the group's first
    const uint32_t m_lo = lane < 32 ? 1u << lane : 0u, m_hi = lane < 32 ? 0u : 1u << (lane - 32);   // the lane's bit of a group's two words
    const float lane_f = (float)lane;
    // this wave's entries: w, w + NW, ... -- at most 64 of them (a chunk lists at most kFrameThreads groups)
    const int my_e = wave + lane * NW;
    uint32_t my_yx = 0u, my_lo = 0u, my_hi = 0u;
    if (my_e < n_list) { my_yx = list[my_e]; my_lo = words[2 * my_e]; my_hi = words[2 * my_e + 1]; }
    const int n_mine = n_list > wave ? min(64, (n_list - wave + NW - 1) / NW) : 0;   // (wave-uniform)
    for (int u0 = 0; u0 < n_mine; u0 += NCH) {
        uint2 raw[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            raw[u] = make_uint2(0u, 0u);
            if (u0 + u < n_mine) {   // (wave-uniform)
                const uint32_t yx = (uint32_t)__builtin_amdgcn_readlane((int)my_yx, u0 + u);
                const uint32_t y0 = yx >> 16, x0 = yx & 0xFFFFu;
                const uint32_t off = ((y0 >> sh) * (uint32_t)g.cols + (x0 >> sh)) * elem;   // (scalar unit)
                const ROFT_GLOBAL unsigned char* p = fl + off;
                if (FT == ROFT_FLOW_S16C2) {
                    raw[u] = make_uint2(*(const ROFT_GLOBAL uint32_t*)(p + lane_off), 0u);
                } else {
                    const unsigned long long w = *(const ROFT_GLOBAL unsigned long long*)(p + lane_off);
                    raw[u] = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            if (u0 + u < n_mine) {
                const uint32_t yx = (uint32_t)__builtin_amdgcn_readlane((int)my_yx, u0 + u);
                uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)my_lo, u0 + u);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)my_hi, u0 + u);
                if (clear00 && yx == 0u) lo &= ~1u;                      // mask_.at<uchar>(0,0) = 0
                const float x0f = (float)(int)(yx & 0xFFFFu), y0f = (float)(int)(yx >> 16);   // (wave-uniform)
                float dx, dy;
                if (FT == ROFT_FLOW_S16C2) { dx = (float)(short)(raw[u].x & 0xFFFFu); dy = (float)(short)(raw[u].x >> 16); }
                else { dx = __uint_as_float(raw[u].x); dy = __uint_as_float(raw[u].y); }
                if (MODE == 1) { dx *= g.inv_scale; dy *= g.inv_scale; }
                const float t_x = (x0f + lane_f) + dx, t_y = y0f + dy;
                const int ix = trunc_clamped(t_x), iy = trunc_clamped(t_y);
                if ((((lo & m_lo) | (hi & m_hi)) != 0u) && (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H) tgt.hit(iy * W + ix);
            }
        }
    }
}

// binary source: OR-scatter of the listed groups' pixels.  kBinaryWalks walks in flight per wave.
constexpr int kBinaryWalks = 8;
template <int FT, int NW>
__device__ __forceinline__ void propagate_binary(ChaseGeo g, const uint2* plane2, const uint32_t* list, int n_list, int n_flows,
                                                 bool clear00, const void* const* flows, const OrTarget tgt, const uint2* words)
{
    if (n_flows == 1 && words && g.mode != 0) {
        if ((g.W & 63) == 0 && n_list <= 64 * NW) {
            if (g.mode == 2) walk_single_aligned<FT, 2, NW>(g, list, words, n_list, clear00, flows[0], tgt);
            else walk_single_aligned<FT, 1, NW>(g, list, words, n_list, clear00, flows[0], tgt);
        } else if (g.mode == 2) walk_single<FT, 2, NW>(g, list, words, n_list, clear00, flows[0], tgt);
        else walk_single<FT, 1, NW>(g, list, words, n_list, clear00, flows[0], tgt);
        return;
    }
    auto hit = [tgt](int tp, int, int, int) { tgt.hit(tp); };
    if (g.mode == 2) chase_groups<FT, kBinaryWalks, 2, NW>(g, plane2, list, n_list, n_flows, clear00, flows, hit, words);
    else if (g.mode == 1) chase_groups<FT, kBinaryWalks, 1, NW>(g, plane2, list, n_list, n_flows, clear00, flows, hit, words);
    else chase_groups<FT, 4, 0, NW>(g, plane2, list, n_list, n_flows, clear00, flows, hit, words);
}

// general source: map of the winning (largest) source index per target + the targets' bounding box (LDS bbox[4])
template <int FT, int NW>
__device__ __noinline__ void propagate_general(ChaseGeo g, const uint2* plane2, const uint32_t* list, int n_list, int n_flows,
                                               bool clear00, const void* const* flows, int32_t* map, int* s_bbox)
{
    int bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = -1, by1 = -1;
    auto hit = [map, &bx0, &by0, &bx1, &by1](int tp, int ix, int iy, int p) {
        atomicMax(&map[tp], p);
        bx0 = min(bx0, ix); bx1 = max(bx1, ix);
        by0 = min(by0, iy); by1 = max(by1, iy);
    };
    if (g.mode == 2) chase_groups<FT, 4, 2, NW>(g, plane2, list, n_list, n_flows, clear00, flows, hit);
    else if (g.mode == 1) chase_groups<FT, 4, 1, NW>(g, plane2, list, n_list, n_flows, clear00, flows, hit);
    else chase_groups<FT, 4, 0, NW>(g, plane2, list, n_list, n_flows, clear00, flows, hit);
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

// Frame t of the batch, binary masks.  grid: (Q, n_obj).  Workgroup q of an object owns the 64-pixel groups
// [q, q + 1) * grp_per_wg of the source plane (a band of image rows) and the same share of the planes it copies / fills /
// zeroes.  It ORs the targets of its pixels into an LDS window -- the band's rows and `margin` rows above and below;
// the few pixels that fly further go to the destination plane directly -- and flushes the window's non-zero words with
// atomicOr into the destination, which the frame before left zeroed (bands overlap in their margins: an order-free OR).
// Every workgroup of an object makes the same decisions from the same two records; workgroup 0 leaves the record of the
// frame for the next one.  Frames whose source is three-valued are left to mask_general_kernel (bit t of mask_general).
// dynamic LDS: [win_cap words] window | [kFrameThreads] list | [kFrameThreads] plane words of the listed groups
#ifdef ROFT_MASK_PROFILE
// absolute 100 MHz stamps per object: dbg[8] = 2^62 - earliest workgroup start, dbg[i] = latest workgroup passing phase i (atomicMax both)
#define MTICK(i) do { __syncthreads(); if (threadIdx.x == 0) atomicMax((unsigned long long*)&a.state[blockIdx.y].dbg[(i)], (unsigned long long)wall_clock64()); } while (0)
#else
#define MTICK(i) do {} while (0)
#endif

#ifndef ROFT_MASK_WPE
#define ROFT_MASK_WPE 4   // (experiments: 7 forces 72 registers -- and 36 bytes of scratch in the multi-flow walk)
#endif
template <int FT, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(ROFT_MASK_WPE, 8)))
void mask_frame_kernel(EngineArrays a, int t, int frames_between, int flow_aided, int grp_per_wg, int margin, int win_cap)
{
#ifdef ROFT_MASK_PROFILE
    if (threadIdx.x == 0) atomicMax((unsigned long long*)&a.state[blockIdx.y].dbg[8], (unsigned long long)((1ll << 62) - wall_clock64()));
#endif
    ROFT_RESIDENT(a, THREADS == 128 ? RK_MASK_FRESH_EMPTY : RK_MASK_FRAME_EMPTY);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ MaskShared S;
    __shared__ FrameCtrl s_c;
    __shared__ MaskRec s_rec[2];   // [0] state after the frame before (frame 0: the carry), [1] this frame's counters
    static_assert(sizeof(MaskRec) == 32, "two 16-byte loads per record");
    uint32_t* s_win = reinterpret_cast<uint32_t*>(smem);
    uint32_t* s_list = s_win + win_cap;
    uint2* s_words = reinterpret_cast<uint2*>(s_list + THREADS);
    const int obj = blockIdx.y, q = blockIdx.x;
    const int W = a.cam.W, H = a.cam.H, wpr = a.cam.wpr, n_grp = (W * H) >> 6;
    const int tid = threadIdx.x, lane = tid & 63;
    const int g0 = q * grp_per_wg, g1 = min(n_grp, g0 + grp_per_wg);
    // the band's rows + margin = the LDS window
    const int r_lo = max(0, (g0 * 64) / W - margin), r_hi = min(H - 1, (g1 * 64 - 1) / W + margin);
    const int win_off = r_lo * wpr, win_words = min(win_cap, (r_hi - r_lo + 1) * wpr);
    // ---- one round trip: control block, the two records, and (speculatively) this thread's group word of BOTH planes the
    //      frame can read -- the last propagated mask (ring slot slot_prev0 + t inside the engine) and the mask delivered
    //      with the frame (slot_new + t; stale but valid memory when none was delivered)
    stage_ctrl(&s_c, frame_ctrl(a, t, obj));
    if (tid >= THREADS / 2 && tid < THREADS / 2 + 4) {
        const int k = tid - THREADS / 2;   // 0, 1: the state after the frame before; 2, 3: this frame's counters
        const MaskRec* prev = (t == 0) ? a.mrec_carry + obj : a.mrec + (size_t)t * a.n_obj + obj;
        reinterpret_cast<uint4*>(s_rec)[k] = (k >= 2) ? reinterpret_cast<const uint4*>(a.mrec + (size_t)(t + 1) * a.n_obj + obj)[k & 1]
                                                      : reinterpret_cast<const uint4*>(prev)[k & 1];
    }
    const int guess_prev = a.slot_prev0 >= 0 ? (a.slot_prev0 + t) % kPlaneSlots : -1, guess_new = a.slot_new + t;
    const ROFT_GLOBAL unsigned long long* pl_prev = guess_prev >= 0 ? (const ROFT_GLOBAL unsigned long long*)(a.planes + plane_offset(a, obj, guess_prev, 1)) : nullptr;
    const ROFT_GLOBAL unsigned long long* pl_new = (const ROFT_GLOBAL unsigned long long*)(a.planes + plane_offset(a, obj, guess_new, 1));
    unsigned long long w_prev = 0ull, w_new = 0ull;
    if (g0 + tid < g1) {
        if (pl_prev) w_prev = pl_prev[g0 + tid];
        w_new = pl_new[g0 + tid];
    }
    if (tid == THREADS * 3 / 4) { S.w00[0] = pl_prev ? pl_prev[0] : 0ull; S.w00[1] = pl_new[0]; }
    for (int i = tid; i < win_words; i += THREADS) s_win[i] = 0u;
    if (tid == 0) S.n_list = 0;
    __syncthreads();
    MTICK(4);
    const FrameCtrl& c = s_c;
    const MaskRec r = decide_frame(s_rec[0], s_rec[1], a.slot_new + t, c, frames_between, flow_aided);
    if (q == 0 && tid == 0) a.mrec[(size_t)(t + 1) * a.n_obj + obj] = r;   // (every workgroup computes the same record)
    if (tid < kMaxFlowHist) S.flows[tid] = (tid < r.n_flows) ? c.flow[tid] : nullptr;
    MTICK(0);
    // the obj plane of the NEXT frame's slot zeroed for that frame's OR flush (nobody reads that slot any more: its last
    // user is kPlaneSlots frames back)
    const size_t sh0 = (size_t)2 * g0, sh_n = (size_t)2 * (g1 - g0);   // this workgroup's share of a plane, in words
    plane_fill(a.planes + plane_offset(a, obj, (c.slot_cur + 1) % kPlaneSlots, 1) + sh0, 0u, sh_n);
    MTICK(5);
    const uint32_t* src = a.planes + plane_offset(a, obj, r.src_slot, 1);
    uint32_t* dst = a.planes + plane_offset(a, obj, c.slot_cur, 1);
    if (!r.src_binary) {
        if (q == 0 && tid == 0) atomicOr(&a.mask_general[obj], 1u << t);   // three-valued source: mask_general_kernel
        return;
    }
    if (r.mode == 0) {
        plane_copy(dst + sh0, src + sh0, sh_n);
        return;
    }
    const bool from_guess = r.src_slot == guess_prev || r.src_slot == guess_new;
    const unsigned long long w00 = from_guess ? S.w00[r.src_slot == guess_new ? 1 : 0] : *reinterpret_cast<const unsigned long long*>(src);
    if (r.mode == 2 && (w00 & 1ull)) {
        // mask(0,0) set in a new-mask frame: every target, mapped or not, samples a set pixel
        plane_fill(dst + sh0, ~0u, sh_n);
        return;
    }
    MTICK(6);
    const ChaseGeo geo = make_chase_geo(a.cam, a.ffmt);
    OrTarget tgt;
    tgt.win = (ROFT_LDS uint32_t*)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(ROFT_LDS uint32_t*)s_win);
    tgt.off = win_off;
    tgt.words = win_words;
    tgt.dst = dst;
    // this workgroup's non-empty groups -> list (any order: the scatter is order-free), kFrameThreads groups at a time, then
    // their walks
    const uint2* plane2 = reinterpret_cast<const uint2*>(src);
    for (int c0 = g0; c0 < g1; c0 += THREADS) {
        const int g = c0 + tid;
        unsigned long long ww = 0ull;
        if (g < g1) {
            if (c0 == g0 && from_guess) ww = (r.src_slot == guess_new) ? w_new : w_prev;
            else ww = *reinterpret_cast<const unsigned long long*>(plane2 + g);
        }
        if (c0 > g0) {
            __syncthreads();   // the walks of the chunk before have read the list
            if (tid == 0) S.n_list = 0;
            __syncthreads();
        }
        MTICK(7);
        const bool ne = ww != 0ull;
        const unsigned long long b = __ballot(ne);
        int base = 0;
        if (lane == 0 && b) base = atomicAdd(&S.n_list, __popcll(b));
        base = __shfl(base, 0, 64);
        if (ne) {
            const int e = base + __popcll(b & ((1ull << lane) - 1ull));
            const int p0 = g * 64, y0 = p0 / W;
            s_list[e] = ((uint32_t)y0 << 16) | (uint32_t)(p0 - y0 * W);
            s_words[e] = make_uint2((uint32_t)ww, (uint32_t)(ww >> 32));
        }
        __syncthreads();
        MTICK(1);
        if (S.n_list > 0) ROFT_RESIDENT_AS(a, THREADS == 128 ? RK_MASK_FRESH : RK_MASK_FRAME);
        propagate_binary<FT, THREADS / 64>(geo, plane2, s_list, S.n_list, r.n_flows, r.mode == 1, S.flows, tgt, s_words);
    }
    __syncthreads();
    MTICK(2);
    // flush: the non-zero words of the window into the (zeroed) destination
    for (int i = tid; i < win_words; i += THREADS) {
        const uint32_t v = s_win[i];
        if (v) atomicOr(&dst[win_off + i], v);
    }
    MTICK(3);
}

constexpr int kMaskThreads = 256;   // (idle on binary masks -- every mask the reference's sources deliver: four waves find room anywhere)
constexpr int kMaskWaves = kMaskThreads / 64;
constexpr int kGeneralList = 16384;   // 64-pixel groups listed at a time by mask_general_kernel (64 KB of LDS)
// One workgroup per object at the end of the batch's mask frames: the frames whose source is three-valued.
// dynamic LDS: list of the non-empty groups
template <int FT>
__global__ __launch_bounds__(kMaskThreads) void mask_general_kernel(EngineArrays a, int final_launch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ MaskShared S;
    uint32_t* s_list = reinterpret_cast<uint32_t*>(smem);
    const int obj = blockIdx.x;
    const int W = a.cam.W, H = a.cam.H, npix = W * H, n_grp = npix >> 6;
    const int tid = threadIdx.x, lane = tid & 63;
    ROFT_RESIDENT(a, RK_MASK_GENERAL);
    // The chain is through with this batch's ingest counters (rows 1 .. T of its table, read by the frame kernels' decisions): left
    // zeroed for the batch that uses the table next, whose ingest may then share a launch with its control block upload
    // (ctrl_ingest_kernel).  Only the LAST launch of a chain does this (the early one runs in front of the last frame).
    if (final_launch && tid < a.T) {
        MaskRec& r0 = a.mrec[(size_t)(tid + 1) * a.n_obj + obj];
        r0.new_count = 0;
        r0.new_ones = 0;
    }
    const unsigned todo = a.mask_general[obj];
    if (!todo) return;   // (almost always 0: every mask the reference's sources deliver is binary)
    __syncthreads();
    if (tid == 0) a.mask_general[obj] = 0u;   // (the next chain's frames set their bits behind this kernel)
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
            int32_t* map = a.map + (size_t)obj * npix;
            // the non-empty groups kGeneralList at a time (the list lives in LDS; an image of any size goes through in pieces,
            // the winner map and the targets' box accumulate over them)
            for (int c0 = 0; c0 < n_grp; c0 += kGeneralList) {
                const int c1 = min(n_grp, c0 + kGeneralList);
                for (int g0 = c0; g0 < c1; g0 += kMaskThreads) {
                    const int g = g0 + tid;
                    bool ne = false;
                    if (g < c1) { const uint2 w = plane2[g]; ne = (w.x | w.y) != 0u; }
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
                propagate_general<FT, kMaskWaves>(make_chase_geo(a.cam, a.ffmt), plane2, s_list, S.n_list, r.n_flows, r.mode == 1, S.flows, map, S.bbox);
                __syncthreads();
                if (tid == 0) S.n_list = 0;
                __syncthreads();
            }
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

int launch_mask_chain(const EngineArrays& a, int frames_between, int flow_aided, unsigned new_mask_frames, hipStream_t s, hipEvent_t stop,
                      hipEvent_t stop_early)
{
    const int n_grp = a.cam.W * a.cam.H / 64;
    // Groups per workgroup.  Automatic: ~3 image rows' worth of pixels more than a band of 16 rows at 640 pixels -- a workgroup
    // inside the object then lists a few dozen non-empty groups, a dozen per wave, whose flow reads go out in ONE round
    // (walk_single) -- and a third of that on frames that deliver a mask (the schedule tells the host which: the new mask is
    // chased through up to 30 flows, a chain of dependent reads per group).  roft_config::mask_workgroups_per_object = S > 0
    // splits the plane into exactly S bands instead (1: one workgroup walks the whole object, chunk by chunk).
    static const int rows_env = getenv("ROFT_MASK_ROWS") ? atoi(getenv("ROFT_MASK_ROWS")) : 0;           // (experiments)
    static const int rows_new_env = getenv("ROFT_MASK_ROWS_NEW") ? atoi(getenv("ROFT_MASK_ROWS_NEW")) : 0;
    const int rows_auto = rows_env > 0 ? rows_env : 20, rows_auto_new = rows_new_env > 0 ? rows_new_env : 6;
    auto per_for = [&](bool fresh) {
        if (a.mask_wgs > 0) return (n_grp + a.mask_wgs - 1) / a.mask_wgs;
        const int per = std::max(1, (fresh ? rows_auto_new : rows_auto) * a.cam.W / 64);
        return std::min(per, n_grp);
    };
    const size_t lds_cap = 160 * 1024 - 4096;   // (the kernel's static LDS -- control block, records, flow pointers -- is ~1.3 KB)
    {
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(mask_frame_kernel<ROFT_FLOW_S16C2, kFrameThreads>), (int)lds_cap);
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(mask_frame_kernel<ROFT_FLOW_F32C2, kFrameThreads>), (int)lds_cap);
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(mask_frame_kernel<ROFT_FLOW_S16C2, 128>), (int)lds_cap);
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(mask_frame_kernel<ROFT_FLOW_F32C2, 128>), (int)lds_cap);
    }
    int launches = 0;
    const bool s16 = a.ffmt.type == ROFT_FLOW_S16C2;
    // the frames of objects with three-valued masks (none, normally): one workgroup per object behind the frame kernels
    auto launch_general = [&](hipEvent_t ev, int final_launch) {
        const size_t lds_gen = ((size_t)std::min(n_grp, kGeneralList) * 4 + 15) & ~(size_t)15;
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(mask_general_kernel<ROFT_FLOW_S16C2>), kGeneralList * 4 + 16);
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(mask_general_kernel<ROFT_FLOW_F32C2>), kGeneralList * 4 + 16);
        if (s16)
            hipExtLaunchKernelGGL(mask_general_kernel<ROFT_FLOW_S16C2>, dim3(a.n_obj), dim3(kMaskThreads), (uint32_t)lds_gen, s, nullptr, ev, 0, a, final_launch);
        else
            hipExtLaunchKernelGGL(mask_general_kernel<ROFT_FLOW_F32C2>, dim3(a.n_obj), dim3(kMaskThreads), (uint32_t)lds_gen, s, nullptr, ev, 0, a, final_launch);
    };
    for (int t = 0; t < a.T; ++t) {
        const bool fresh = (new_mask_frames >> t) & 1u;
        const int per = per_for(fresh);
        // LDS window: the band's rows + a margin of rows above and below (pixels that fly further are ORed into the
        // destination plane directly): 16 rows for one flow step, 48 when a new mask is chased through several
        const int margin = fresh ? 48 : 16;
        // Workgroups of TWO waves on a frame that delivers a mask: its 80 bands x n_obj workgroups exceed what the device holds at once
        // (seven per CU by wave slots at four waves: the last one starts 33 us after the first), more than half of them find no
        // pixel and leave after one round trip, and the others chase their groups through six flows -- six dependent round trips
        // whatever the number of waves.  Half the waves per workgroup = twice the workgroups in flight: +1 - 3 % in runs of 60 steps
        // and more, nothing in a 20-frame burst.  (ROFT_MASK_FRESH_THREADS=256: four waves as on every other frame.)
        static const int fresh_threads_env = getenv("ROFT_MASK_FRESH_THREADS") ? atoi(getenv("ROFT_MASK_FRESH_THREADS")) : 128;
        const int threads = (fresh && fresh_threads_env == 128 && kFrameThreads > 128) ? 128 : kFrameThreads;
        const size_t fixed = (size_t)threads * 12;
        size_t win_cap = ((size_t)std::min(a.cam.H, (per * 64 + a.cam.W - 1) / a.cam.W + 1 + 2 * margin) * a.cam.wpr + 1) & ~(size_t)1;
        win_cap = std::min(win_cap, ((lds_cap - fixed) / 4) & ~(size_t)1);
        const size_t lds = win_cap * 4 + fixed;
        const dim3 grid((n_grp + per - 1) / per, a.n_obj);
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(threads), (uint32_t)lds, s, a, t, frames_between, flow_aided, per, margin, (int)win_cap); };
        if (threads == 128) { if (s16) go(mask_frame_kernel<ROFT_FLOW_S16C2, 128>); else go(mask_frame_kernel<ROFT_FLOW_F32C2, 128>); }
        else if (s16) go(mask_frame_kernel<ROFT_FLOW_S16C2, kFrameThreads>);
        else go(mask_frame_kernel<ROFT_FLOW_F32C2, kFrameThreads>);
        ++launches;
        // `stop_early`: the masks up to the batch's LAST BUT ONE frame are complete -- all that the flow measurements of the batch
        // read (frame t measures inside the mask of frame t - 1); the three-valued frames so far are brought up to date for it
        // (the kernel clears the bits it has served, the last frame sets its own again)
        if (stop_early && t == a.T - 2) { launch_general(stop_early, 0); ++launches; }
    }
    launch_general(stop, 1);
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

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
// One workgroup of 16 waves per object walks the frames of a batch (mask_chain_kernel): the mask of frame k is the
// source of frame k+1, so the recursion is sequential per object and a launch per frame only adds dispatch
// gaps.  Per frame: ingest of a newly delivered mask (u8 -> planes, counts), the mode decision of step_frame,
// and the propagation:
//  * binary masks (no pixel of value 1, i.e. nz == obj -- decided on the device at ingest): every source pixel
//    carries the same value, so the reference's "later writer wins" map + remap is an order-free OR of the target
//    bits.  The target plane lives in LDS (W*H/8 bytes), LDS atomicOr, one coalesced write-out; no map, no gather.
//  * general masks ({0, 1, 255}): the winner among the sources of a target is the one with the LARGEST linear
//    index = an atomicMax on a W*H int32 map whose zero value doubles as "unmapped -> sample mask(0,0)" exactly like
//    the zero-initialised cv::Mat map (:237); the gather reads every entry inside the targets' bounding box with an
//    atomic exchange (read + clear), so the map is never memset and never read through a stale L1 line.
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

// operator level (roft_flow_measurement, roft_depth_likelihood): frame 0's mask -> plane slot kSlotNew.
// grid: (ceil(W*H/64/256), n_obj)
__global__ __launch_bounds__(256) void mask_ingest_kernel(EngineArrays a)
{
    const int obj = blockIdx.y;
    const FrameCtrl& c = a.ctrl[obj];
    if (!c.has_new_mask) return;
    const int n_grp = (a.cam.W * a.cam.H) >> 6;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    int count = 0, ones = 0;
    if (g < n_grp)
        ingest_group(reinterpret_cast<const uint4*>(c.new_mask), g,
                     reinterpret_cast<uint2*>(a.planes + plane_offset(a, obj, kSlotNew, 0)),
                     reinterpret_cast<uint2*>(a.planes + plane_offset(a, obj, kSlotNew, 1)), count, ones);
}

void launch_mask_ingest(const EngineArrays& a, hipStream_t s)
{
    const int n_grp = a.cam.W * a.cam.H / 64;
    hipLaunchKernelGGL(mask_ingest_kernel, dim3((n_grp + 255) / 256, a.n_obj), dim3(256), 0, s, a);
}

// ---- mask chain ------------------------------------------------------------------------------------
constexpr int kMaskThreads = 1024;
constexpr int kMaskWaves = kMaskThreads / 64;
// (64-pixel groups whose walks through the flows are in flight together in one wave -- chase_groups' NCH: a pixel's
//  walk is a chain of dependent loads, the chains of different groups are independent)

struct MaskShared {
    int red[2][kMaskWaves];
    int bbox[4];
    int n_list;
    const void* flows[kMaxFlowHist];
};

__device__ __forceinline__ int2 block_sum2(int a0, int a1, MaskShared& S)
{
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { S.red[0][threadIdx.x >> 6] = a0; S.red[1][threadIdx.x >> 6] = a1; }
    __syncthreads();
    int t0 = 0, t1 = 0;
    for (int w = 0; w < kMaskWaves; ++w) { t0 += S.red[0][w]; t1 += S.red[1][w]; }
    return make_int2(t0, t1);
}

// Geometry the walks need (a handful of scalars instead of the whole EngineArrays block)
struct ChaseGeo {
    int W, H, cols;
    float grid_f, scale;
};

// Walks of the source pixels of one frame.  `list` (LDS) holds the indices of the non-empty 64-pixel groups of the
// source plane; wave w owns entries w, w + 16, ... (the object's rows spread over all waves), prefetches the plane
// words of up to 64 of them with one load (lane i <-> the wave's i-th entry) and chases them NCH at a time: the
// flow reads of a wave are row-contiguous (64 x 8 B).  A surviving pixel is handed to `hit(target, x, y, source)`.
template <int FT, int NCH, class Hit>
__device__ __forceinline__ void chase_groups(const ChaseGeo g, const uint2* plane2, const uint16_t* list, int n_list,
                                             int n_flows, bool clear00, const void* const* flows, Hit hit)
{
    const int W = g.W, H = g.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e0 = wave; e0 < n_list; e0 += kMaskWaves * 64) {
        const int my_e = e0 + lane * kMaskWaves;
        int my_grp = -1;
        uint2 mine = make_uint2(0u, 0u);
        if (my_e < n_list) { my_grp = list[my_e]; mine = plane2[my_grp]; }
        unsigned long long pending = __ballot(my_grp >= 0);
        while (pending) {
            float t_x[NCH], t_y[NCH];
            int grp[NCH];
            bool act[NCH];
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                act[u] = false;
                grp[u] = 0;
                t_x[u] = t_y[u] = 0.0f;
                if (pending) {
                    const int it = __builtin_ctzll(pending);
                    pending &= pending - 1;
                    grp[u] = __builtin_amdgcn_readlane(my_grp, it);
                    unsigned long long bits = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mine.y, it) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)mine.x, it);
                    if (clear00 && grp[u] == 0) bits &= ~1ull;                 // mask_.at<uchar>(0,0) = 0
                    act[u] = (bits >> lane) & 1ull;
                    const int p = grp[u] * 64 + lane;
                    const int py = p / W, px = p - py * W;
                    t_x[u] = (float)px;
                    t_y[u] = (float)py;
                }
            }
            // flows in chronological order: oldest buffered first (flows[n_flows-1]) ... current (flows[0])
            for (int j = n_flows - 1; j >= 0; --j) {
                const void* fl = flows[j];
                uint2 raw[NCH];
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int ix = trunc_int_x86(t_x[u]), iy = trunc_int_x86(t_y[u]);
                    if (ix < 0 || ix >= W || iy < 0 || iy >= H) act[u] = false;   // left the image: the pixel is dropped
                    // inactive lanes read element (0, 0): the loads stay unconditional and in flight together
                    const int fr = act[u] ? trunc_int_x86(t_y[u] / g.grid_f) : 0;
                    const int fc = act[u] ? trunc_int_x86(t_x[u] / g.grid_f) : 0;
                    raw[u] = flow_raw<FT>(fl, (size_t)(fr * g.cols + fc));
                }
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    float dx, dy;
                    flow_decode<FT>(raw[u], g.scale, dx, dy);
                    t_x[u] += dx;
                    t_y[u] += dy;
                }
            }
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                const int ix = trunc_int_x86(t_x[u]), iy = trunc_int_x86(t_y[u]);
                if (!act[u] || ix < 0 || ix >= W || iy < 0 || iy >= H) continue;
                hit(iy * W + ix, ix, iy, grp[u] * 64 + lane);
            }
        }
    }
}

// binary source: OR-scatter into the LDS plane (eight walks in flight per wave)
template <int FT>
__device__ __noinline__ void propagate_binary(ChaseGeo g, const uint2* plane2, const uint16_t* list, int n_list, int n_flows,
                                              bool clear00, const void* const* flows, uint32_t* s_tgt)
{
    chase_groups<FT, 8>(g, plane2, list, n_list, n_flows, clear00, flows,
                        [&](int tp, int, int, int) { atomicOr(&s_tgt[tp >> 5], 1u << (tp & 31)); });
}

// general source: map of the winning (largest) source index per target + the targets' bounding box (LDS bbox[4])
template <int FT>
__device__ __noinline__ void propagate_general(ChaseGeo g, const uint2* plane2, const uint16_t* list, int n_list, int n_flows,
                                               bool clear00, const void* const* flows, int32_t* map, int* s_bbox)
{
    int bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = -1, by1 = -1;
    chase_groups<FT, 4>(g, plane2, list, n_list, n_flows, clear00, flows, [&](int tp, int ix, int iy, int p) {
        atomicMax(&map[tp], p);
        bx0 = min(bx0, ix); bx1 = max(bx1, ix);
        by0 = min(by0, iy); by1 = max(by1, iy);
    });
    for (int off = 32; off > 0; off >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, off, 64)); by0 = min(by0, __shfl_xor(by0, off, 64));
        bx1 = max(bx1, __shfl_xor(bx1, off, 64)); by1 = max(by1, __shfl_xor(by1, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && bx1 >= 0) {
        atomicMin(&s_bbox[0], bx0); atomicMin(&s_bbox[1], by0);
        atomicMax(&s_bbox[2], bx1); atomicMax(&s_bbox[3], by1);
    }
}

// plane-wide copies / fills by the whole workgroup: 16-byte accesses when the planes are 16-byte aligned
// (plane_words % 4 == 0), 8-byte ones otherwise (plane_words is always even: W*H % 64 == 0)
__device__ __forceinline__ void plane_copy2(uint32_t* d0, uint32_t* d1, const uint32_t* s0, const uint32_t* s1, size_t n_words)
{
    if ((n_words & 3) == 0) {
        for (size_t i = threadIdx.x; i < n_words / 4; i += blockDim.x) {
            reinterpret_cast<uint4*>(d0)[i] = reinterpret_cast<const uint4*>(s0)[i];
            reinterpret_cast<uint4*>(d1)[i] = reinterpret_cast<const uint4*>(s1)[i];
        }
    } else {
        for (size_t i = threadIdx.x; i < n_words / 2; i += blockDim.x) {
            reinterpret_cast<uint2*>(d0)[i] = reinterpret_cast<const uint2*>(s0)[i];
            reinterpret_cast<uint2*>(d1)[i] = reinterpret_cast<const uint2*>(s1)[i];
        }
    }
}

__device__ __forceinline__ void plane_fill(uint32_t* d, uint32_t v, size_t n_words)
{
    if ((n_words & 3) == 0)
        for (size_t i = threadIdx.x; i < n_words / 4; i += blockDim.x) reinterpret_cast<uint4*>(d)[i] = make_uint4(v, v, v, v);
    else
        for (size_t i = threadIdx.x; i < n_words / 2; i += blockDim.x) reinterpret_cast<uint2*>(d)[i] = make_uint2(v, v);
}

// dynamic LDS: [plane_words] OR target | [W*H/64] uint16 list of non-empty groups
template <int FT>
__global__ __launch_bounds__(kMaskThreads) void mask_chain_kernel(EngineArrays a, int frames_between, int flow_aided)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ MaskShared S;
    uint32_t* s_tgt = reinterpret_cast<uint32_t*>(smem);
    uint16_t* s_list = reinterpret_cast<uint16_t*>(smem + ((a.plane_words * 4 + 15) & ~(size_t)15));
    const int obj = blockIdx.x;
    ObjState& st = a.state[obj];
    const int W = a.cam.W, H = a.cam.H, npix = W * H, n_grp = npix >> 6;
    const int tid = threadIdx.x, lane = tid & 63;
    int fbuf_n = st.fbuf_n, cur_binary = st.mask_binary, mode = 0;

    for (int t = 0; t < a.T; ++t) {
        const FrameCtrl& c = frame_ctrl(a, t, obj);
        const int has_new = c.has_new_mask;
        // ---- a newly delivered mask -> plane slot kSlotNew, its non-zero count and whether it is binary
        int new_count = 0, new_binary = 1;
        if (has_new) {
            const uint4* src = reinterpret_cast<const uint4*>(c.new_mask);
            uint2* nz = reinterpret_cast<uint2*>(a.planes + plane_offset(a, obj, kSlotNew, 0));
            uint2* ob = reinterpret_cast<uint2*>(a.planes + plane_offset(a, obj, kSlotNew, 1));
            int count = 0, ones = 0;
            for (int g = tid; g < n_grp; g += kMaskThreads) ingest_group(src, g, nz, ob, count, ones);
            const int2 tot = block_sum2(count, ones, S);   // (its barriers also order the plane stores before the reads below)
            new_count = tot.x;
            new_binary = tot.y == 0;
        }
        uint32_t* dnz = a.planes + plane_offset(a, obj, c.slot_cur, 0);
        uint32_t* dob = a.planes + plane_offset(a, obj, c.slot_cur, 1);
        int src_slot, n_flows;
        if (flow_aided) {
            mode = decide_mode(c, fbuf_n, new_count, frames_between, src_slot, n_flows);
        } else {  // no flow-aided segmentation: the delivered mask is used as is, otherwise the last one persists
            mode = 0;
            src_slot = has_new ? kSlotNew : c.slot_prev;
            n_flows = 0;
        }
        const int src_binary = (src_slot == kSlotNew) ? new_binary : cur_binary;
        const uint32_t* snz = a.planes + plane_offset(a, obj, src_slot, 0);
        const uint32_t* sob = a.planes + plane_offset(a, obj, src_slot, 1);

        if (mode == 0) {
            plane_copy2(dnz, dob, snz, sob, a.plane_words);
        } else {
            // background = mask(0,0) of the source: unmapped targets sample it; forced to 0 in mode 1
            const bool bg_nz = (mode == 2) && (snz[0] & 1u), bg_ob = (mode == 2) && (sob[0] & 1u);
            if (tid < kMaxFlowHist) S.flows[tid] = (tid < n_flows) ? c.flow[tid] : nullptr;
            if (tid == 0) { S.n_list = 0; S.bbox[0] = INT32_MAX; S.bbox[1] = INT32_MAX; S.bbox[2] = -1; S.bbox[3] = -1; }
            const bool fast = src_binary && !bg_nz;
            if (fast) plane_fill(s_tgt, 0u, a.plane_words);
            __syncthreads();
            // non-empty 64-pixel groups of the source -> list (any order: both scatters are order-free)
            const uint2* plane2 = reinterpret_cast<const uint2*>(snz);
            for (int g0 = 0; g0 < n_grp; g0 += kMaskThreads) {
                const int g = g0 + tid;
                bool ne = false;
                if (g < n_grp) { const uint2 w = plane2[g]; ne = (w.x | w.y) != 0u; }
                const unsigned long long b = __ballot(ne);
                int base = 0;
                if (lane == 0 && b) base = atomicAdd(&S.n_list, __popcll(b));
                base = __shfl(base, 0, 64);
                if (ne) s_list[base + __popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)g;
            }
            __syncthreads();
            const int n_list = S.n_list;
            const ChaseGeo geo{W, H, a.ffmt.cols, (float)a.ffmt.grid, a.ffmt.scale};
            if (fast) {
                // ---- binary source: OR-scatter into the LDS plane
                propagate_binary<FT>(geo, plane2, s_list, n_list, n_flows, mode == 1, S.flows, s_tgt);
                __syncthreads();
                plane_copy2(dnz, dob, s_tgt, s_tgt, a.plane_words);
            } else if (src_binary) {
                // binary source with mask(0,0) set in a new-mask frame: every target, mapped or not, samples a set pixel
                plane_fill(dnz, ~0u, a.plane_words);
                plane_fill(dob, ~0u, a.plane_words);
            } else {
                // ---- general source: map of the winning (largest) source index per target, then the gather
                int32_t* map = a.map + (size_t)obj * npix;
                propagate_general<FT>(geo, plane2, s_list, n_list, n_flows, mode == 1, S.flows, map, S.bbox);
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
        }
        fbuf_n = flow_aided ? next_fbuf(c, fbuf_n, new_count, mode, frames_between) : 0;
        cur_binary = src_binary;
        __syncthreads();   // this frame's planes are the next frame's source (same workgroup)
    }
    if (tid == 0) { st.fbuf_n = fbuf_n; st.mask_binary = cur_binary; st.mask_mode = mode; }
}

void launch_mask_chain(const EngineArrays& a, int frames_between, int flow_aided, hipStream_t s, hipEvent_t stop)
{
    const size_t lds = ((a.plane_words * 4 + 15) & ~(size_t)15) + (((size_t)a.cam.W * a.cam.H / 64) * 2 + 15 & ~(size_t)15);
    static bool attr_set = false;
    if (!attr_set) {
        const int cap = 160 * 1024 - 256 - (int)sizeof(MaskShared) - 128;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mask_chain_kernel<ROFT_FLOW_S16C2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mask_chain_kernel<ROFT_FLOW_F32C2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        attr_set = true;
    }
    if (a.ffmt.type == ROFT_FLOW_S16C2)
        hipExtLaunchKernelGGL(mask_chain_kernel<ROFT_FLOW_S16C2>, dim3(a.n_obj), dim3(kMaskThreads), (uint32_t)lds, s, nullptr, stop, 0,
                              a, frames_between, flow_aided);
    else
        hipExtLaunchKernelGGL(mask_chain_kernel<ROFT_FLOW_F32C2>, dim3(a.n_obj), dim3(kMaskThreads), (uint32_t)lds, s, nullptr, stop, 0,
                              a, frames_between, flow_aided);
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

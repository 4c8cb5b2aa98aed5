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
// The reference's "later writer wins" scatter in row-major source order is order-free here:
// the winner is the source with the LARGEST linear index, i.e. an atomicMax on a W*H int32 map
// whose zero value doubles as "unmapped -> sample mask(0,0)" exactly like the zero-initialised
// cv::Mat map (:237).  The gather kernel clears the entries it consumed, so the map is never
// memset.
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

// ---- ingest: raw u8 mask -> (nz, obj) bit planes + non-zero count -------------------------------
// One thread converts 64 consecutive pixels: four 16-byte loads, two 64-bit masks built in registers,
// two coalesced 8-byte stores.  grid: (ceil(W*H/64/256), n_obj).
__device__ __forceinline__ void bytes_to_bits(uint32_t w, int shift, unsigned long long& nz, unsigned long long& ob)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t v = (w >> (8 * k)) & 0xFFu;
        nz |= (unsigned long long)(v != 0u) << (shift + k);
        ob |= (unsigned long long)(v > 1u) << (shift + k);
    }
}

__global__ __launch_bounds__(256) void mask_ingest_kernel(EngineArrays a)
{
    const int obj = blockIdx.y;
    const FrameCtrl& c = a.ctrl[obj];
    if (!c.has_new_mask) return;
    const uint4* src = reinterpret_cast<const uint4*>(c.new_mask);
    uint2* nz = reinterpret_cast<uint2*>(a.planes + plane_offset(a, obj, kSlotNew, 0));
    uint2* ob = reinterpret_cast<uint2*>(a.planes + plane_offset(a, obj, kSlotNew, 1));
    const int n_grp = (a.cam.W * a.cam.H) >> 6;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    int count = 0;
    if (g < n_grp) {
        unsigned long long bnz = 0, bob = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = src[(size_t)g * 4 + q];
            bytes_to_bits(v.x, 16 * q, bnz, bob);
            bytes_to_bits(v.y, 16 * q + 4, bnz, bob);
            bytes_to_bits(v.z, 16 * q + 8, bnz, bob);
            bytes_to_bits(v.w, 16 * q + 12, bnz, bob);
        }
        nz[g] = make_uint2((uint32_t)bnz, (uint32_t)(bnz >> 32));
        ob[g] = make_uint2((uint32_t)bob, (uint32_t)(bob >> 32));
        count = __popcll(bnz);
    }
    for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off, 64);
    if ((threadIdx.x & 63) == 0 && count) atomicAdd(&a.state[obj].new_mask_count, count);
}


// ---- scatter (mode decision: decide_mode() in roft_device.h) --------------------------------------
// grid: (power of two >= n_grp/256, n_obj), block 256 = 4 waves; a wave owns 64 consecutive pixels per iteration and
// strides over the image (interleaved, so the object's rows spread over all waves); its lanes chase their
// pixel through the flows in parallel: the flow reads of a wave are row-contiguous (64 x 8 B), the map
// atomics land on neighbouring addresses.  Empty 64-pixel groups cost one wave-uniform 8-byte load.
template <int FT>
__global__ __launch_bounds__(256) void mask_scatter_kernel(EngineArrays a, int frames_between)
{
    const int obj = blockIdx.y;
    const FrameCtrl& c = a.ctrl[obj];
    const ObjState& st = a.state[obj];
    int src_slot, n_flows;
    const int mode = decide_mode(c, st, src_slot, n_flows);
    if (mode == 0) return;
    if (frames_between > 0 && n_flows > frames_between) n_flows = frames_between;

    __shared__ int s_bbox[4];
    if (threadIdx.x < 4) s_bbox[threadIdx.x] = (threadIdx.x < 2) ? INT32_MAX : -1;
    __syncthreads();

    const int W = a.cam.W, H = a.cam.H;
    const int npix = W * H;
    const int lane = threadIdx.x & 63;
    const uint2* plane2 = reinterpret_cast<const uint2*>(a.planes + plane_offset(a, obj, src_slot, 0));
    int32_t* map = a.map + (size_t)obj * npix;
    int bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = -1, by1 = -1;
    const int n_grp = npix / 64, wave_stride = gridDim.x * 4;
    const int wave_first = blockIdx.x * 4 + (threadIdx.x >> 6);
    // lane i prefetches the plane words of this wave's i-th group: one load instead of a chain of
    // dependent wave-uniform loads (needs <= 64 groups per wave)
    const int my_grp = wave_first + lane * wave_stride;
    uint2 mine = make_uint2(0u, 0u);
    if (my_grp < n_grp) mine = plane2[my_grp];
    // Non-empty groups of this wave (bit i <-> lane i's prefetched group).  They are chased kChase at a time: a
    // pixel's walk through the buffered flows is a chain of dependent loads, and the chains of different groups
    // are independent -- issuing them together divides the exposed memory latency by kChase.
    constexpr int kChase = 4;
    unsigned long long pending = __ballot((mine.x | mine.y) != 0u);
    const void* flows[kMaxFlowHist];   // fetched once: a pointer load per flow step would sit in every chain
#pragma unroll
    for (int j = 0; j < kMaxFlowHist; ++j) flows[j] = c.flow[j];
    const float grid_f = (float)a.ffmt.grid;
    while (pending) {
        float t_x[kChase], t_y[kChase];
        int p[kChase];
        bool act[kChase];
#pragma unroll
        for (int u = 0; u < kChase; ++u) {
            act[u] = false;
            p[u] = 0;
            t_x[u] = t_y[u] = 0.0f;
            if (pending) {
                const int it = __builtin_ctzll(pending);
                pending &= pending - 1;
                const int grp = wave_first + it * wave_stride;
                unsigned long long bits = ((unsigned long long)(uint32_t)__shfl((int)mine.y, it, 64) << 32) |
                                          (uint32_t)__shfl((int)mine.x, it, 64);
                if (mode == 1 && grp == 0) bits &= ~1ull;                 // mask_.at<uchar>(0,0) = 0
                act[u] = (bits >> lane) & 1ull;
                p[u] = grp * 64 + lane;
                const int py = p[u] / W, px = p[u] - py * W;
                t_x[u] = (float)px;
                t_y[u] = (float)py;
            }
        }
        // flows in chronological order: oldest buffered first (c.flow[n_flows-1]) ... current
#pragma unroll
        for (int j = kMaxFlowHist - 1; j >= 0; --j) {
            if (j >= n_flows) continue;
            const void* fl = flows[j];
            uint2 raw[kChase];
#pragma unroll
            for (int u = 0; u < kChase; ++u) {
                const int ix = trunc_int_x86(t_x[u]), iy = trunc_int_x86(t_y[u]);
                if (ix < 0 || ix >= W || iy < 0 || iy >= H) act[u] = false;   // left the image: the pixel is dropped
                // inactive lanes read element (0, 0): the loads stay unconditional and in flight together
                const int fr = act[u] ? trunc_int_x86(t_y[u] / grid_f) : 0;
                const int fc = act[u] ? trunc_int_x86(t_x[u] / grid_f) : 0;
                raw[u] = flow_raw<FT>(fl, (size_t)fr * (size_t)a.ffmt.cols + (size_t)fc);
            }
#pragma unroll
            for (int u = 0; u < kChase; ++u) {
                float dx, dy;
                flow_decode<FT>(raw[u], a.ffmt.scale, dx, dy);
                t_x[u] += dx;
                t_y[u] += dy;
            }
        }
#pragma unroll
        for (int u = 0; u < kChase; ++u) {
            const int ix = trunc_int_x86(t_x[u]), iy = trunc_int_x86(t_y[u]);
            if (!act[u] || ix < 0 || ix >= W || iy < 0 || iy >= H) continue;
            atomicMax(&map[iy * W + ix], p[u]);
            bx0 = min(bx0, ix); bx1 = max(bx1, ix);
            by0 = min(by0, iy); by1 = max(by1, iy);
        }
    }
    // bounding box of the targets: wave shuffle -> LDS -> one set of global atomics per block
    for (int off = 32; off > 0; off >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, off, 64)); by0 = min(by0, __shfl_xor(by0, off, 64));
        bx1 = max(bx1, __shfl_xor(bx1, off, 64)); by1 = max(by1, __shfl_xor(by1, off, 64));
    }
    if (lane == 0 && bx1 >= 0) {
        atomicMin(&s_bbox[0], bx0); atomicMin(&s_bbox[1], by0);
        atomicMax(&s_bbox[2], bx1); atomicMax(&s_bbox[3], by1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_bbox[2] >= 0) {
        int* bb = a.state[obj].bbox;
        atomicMin(&bb[0], s_bbox[0]); atomicMin(&bb[1], s_bbox[1]);
        atomicMax(&bb[2], s_bbox[2]); atomicMax(&bb[3], s_bbox[3]);
    }
}

// grid: (ceil(W*H/64/4), n_obj); each wave produces 64 output pixels (two plane words) per step
__device__ void mask_gather_body(const EngineArrays& a, int frames_between)
{
    const int obj = blockIdx.y;
    const FrameCtrl& c = a.ctrl[obj];
    ObjState& st = a.state[obj];
    int src_slot, n_flows;
    const int mode = decide_mode(c, st, src_slot, n_flows);
    const uint32_t* snz = a.planes + plane_offset(a, obj, src_slot, 0);
    const uint32_t* sob = a.planes + plane_offset(a, obj, src_slot, 1);
    uint32_t* dnz = a.planes + plane_offset(a, obj, c.slot_cur, 0);
    uint32_t* dob = a.planes + plane_offset(a, obj, c.slot_cur, 1);
    const int W = a.cam.W;
    const int npix = W * a.cam.H;
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;

    // background = mask(0,0) of the source: forced to 0 in mode 1
    bool bg_nz = false, bg_ob = false;
    if (mode == 2) { bg_nz = snz[0] & 1u; bg_ob = sob[0] & 1u; }
    const int bx0 = st.bbox[0], by0 = st.bbox[1], bx1 = st.bbox[2], by1 = st.bbox[3];
    int32_t* map = a.map + (size_t)obj * npix;

    const uint32_t f1 = bg_nz ? 0xFFFFFFFFu : 0u, f2 = bg_ob ? 0xFFFFFFFFu : 0u;
    if (mode == 0 || (W & 63) == 0) {
        // ---- phase A: copy (mode 0) or constant fill of every 64-pixel group outside the target bounding
        // box; no map traffic.  Groups inside the box are left to phase B.
        // (one 64-pixel group per LANE: the 8-byte words of a wave's 64 groups are stored coalesced)
        const int n_grp = npix >> 6;
        const uint2* snz2 = reinterpret_cast<const uint2*>(snz);
        const uint2* sob2 = reinterpret_cast<const uint2*>(sob);
        uint2* dnz2 = reinterpret_cast<uint2*>(dnz);
        uint2* dob2 = reinterpret_cast<uint2*>(dob);
        for (int g = wave_global * 64 + lane; g < n_grp; g += nwaves * 64) {
            if (mode == 0) {
                dnz2[g] = snz2[g];
                dob2[g] = sob2[g];
                continue;
            }
            const int base = g << 6;
            const int yw = base / W, xw = base - yw * W;
            if (yw < by0 || yw > by1 || xw > bx1 || xw + 63 < bx0) {
                dnz2[g] = make_uint2(f1, f1);
                dob2[g] = make_uint2(f2, f2);
            }
        }
        if (mode == 0 || bx1 < 0) return;
        // ---- phase B: the groups inside the box, spread evenly over all waves and processed four at a
        // time so that the map loads (and then the source-plane loads) of a batch are in flight together
        const int gx0 = bx0 >> 6, cols = (bx1 >> 6) - gx0 + 1, count = (by1 - by0 + 1) * cols;
        for (int q0 = wave_global * 4; q0 < count; q0 += nwaves * 4) {
            int pp[4], mm[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = q0 + k;
                pp[k] = -1;
                mm[k] = 0;
                if (q < count) {
                    const int r = q / cols;
                    const int p = (by0 + r) * W + (gx0 + (q - r * cols)) * 64 + lane;
                    pp[k] = p;
                    const int x = p - (by0 + r) * W;
                    if (x >= bx0 && x <= bx1) mm[k] = map[p];
                }
            }
            uint32_t wn[4], wo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wn[k] = 0; wo[k] = 0;
                if (mm[k] != 0) {
                    map[pp[k]] = 0;
                    wn[k] = snz[mm[k] >> 5];
                    wo[k] = sob[mm[k] >> 5];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (pp[k] < 0) continue;   // wave-uniform
                const bool nzb = mm[k] ? ((wn[k] >> (mm[k] & 31)) & 1u) : bg_nz;
                const bool obb = mm[k] ? ((wo[k] >> (mm[k] & 31)) & 1u) : bg_ob;
                const unsigned long long b1 = __ballot(nzb);
                const unsigned long long b2 = __ballot(obb);
                if (lane == 0) {
                    const int word2 = pp[k] >> 6;
                    reinterpret_cast<uint2*>(dnz)[word2] = make_uint2((uint32_t)b1, (uint32_t)(b1 >> 32));
                    reinterpret_cast<uint2*>(dob)[word2] = make_uint2((uint32_t)b2, (uint32_t)(b2 >> 32));
                }
            }
        }
        return;
    }
    // ---- generic path (W not a multiple of 64: a group may straddle rows)
    for (int base = wave_global * 64; base < npix; base += nwaves * 64) {
        const int word2 = base >> 6;
        const int p = base + lane;
        const int y = p / W, x = p - y * W;
        bool nzb = bg_nz, obb = bg_ob;
        if (y >= by0 && y <= by1 && x >= bx0 && x <= bx1) {
            const int m = map[p];
            if (m != 0) {
                map[p] = 0;
                nzb = (snz[m >> 5] >> (m & 31)) & 1u;
                obb = (sob[m >> 5] >> (m & 31)) & 1u;
            }
        }
        unsigned long long b1 = __ballot(nzb);
        unsigned long long b2 = __ballot(obb);
        if (lane == 0) {
            reinterpret_cast<uint2*>(dnz)[word2] = make_uint2((uint32_t)b1, (uint32_t)(b1 >> 32));
            reinterpret_cast<uint2*>(dob)[word2] = make_uint2((uint32_t)b2, (uint32_t)(b2 >> 32));
        }
    }
}

// (Folding the per-object bookkeeping into this kernel -- "the last workgroup of an object does it" -- was measured
// and dropped: the device-scope fence each of the ~4000 workgroups needs before its atomic counter increment
// writes back the XCD's L2 and made this kernel 157 us instead of 18, slowing every concurrent kernel too.)
__global__ __launch_bounds__(256) void mask_gather_kernel(EngineArrays a, int frames_between)
{
    mask_gather_body(a, frames_between);
}

// bookkeeping after the gather: flow buffer count, reset per-frame scratch state
__global__ void mask_finish_kernel(EngineArrays a, int frames_between)
{
    const int obj = blockIdx.x * blockDim.x + threadIdx.x;
    if (obj >= a.n_obj) return;
    mask_bookkeeping(a.ctrl[obj], a.state[obj]);
}

// no flow-aided segmentation: the delivered mask is used as is, otherwise the last one persists
__global__ __launch_bounds__(256) void mask_plain_kernel(EngineArrays a)
{
    const int obj = blockIdx.y;
    const FrameCtrl& c = a.ctrl[obj];
    const int src_slot = c.has_new_mask ? kSlotNew : c.slot_prev;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.plane_words) return;
    for (int which = 0; which < 2; ++which)
        a.planes[plane_offset(a, obj, c.slot_cur, which) + i] = a.planes[plane_offset(a, obj, src_slot, which) + i];
}

void launch_mask_ingest(const EngineArrays& a, hipStream_t s)
{
    const int n_grp = a.cam.W * a.cam.H / 64;
    hipLaunchKernelGGL(mask_ingest_kernel, dim3((n_grp + 255) / 256, a.n_obj), dim3(256), 0, s, a);
}

void launch_mask_propagate(const EngineArrays& a, int frames_between, int flow_aided, bool finish, hipStream_t s)
{
    if (!flow_aided) {
        hipLaunchKernelGGL(mask_plain_kernel, dim3((unsigned)((a.plane_words + 255) / 256), a.n_obj), dim3(256), 0, s, a);
        if (finish) hipLaunchKernelGGL(mask_finish_kernel, dim3((a.n_obj + 63) / 64), dim3(64), 0, s, a, frames_between);
        return;
    }
    const int n_grp = a.cam.W * a.cam.H / 64;
    // Few, long-lived workgroups: every wave prefetches its (at most 64) plane groups with one load and chases the
    // non-empty ones kChase at a time, so a grid that is resident all at once (64 objects x 32 blocks at 640x480)
    // beats one that needs several rounds of workgroup launches (96 blocks: 0.106 vs 0.097 ms per frame).  A power
    // of two keeps the wave stride from being a multiple of the groups per row -- with 30 blocks (stride 120 = 12
    // rows of 640 pixels) each wave stays in one image column and a few waves get all of the object.
    int sblocks = 8;
    while (sblocks * 4 * 64 < n_grp) sblocks *= 2;   // each wave prefetches at most 64 groups
    if (a.ffmt.type == ROFT_FLOW_S16C2)
        hipLaunchKernelGGL(mask_scatter_kernel<ROFT_FLOW_S16C2>, dim3(sblocks, a.n_obj), dim3(256), 0, s, a, frames_between);
    else
        hipLaunchKernelGGL(mask_scatter_kernel<ROFT_FLOW_F32C2>, dim3(sblocks, a.n_obj), dim3(256), 0, s, a, frames_between);
    const size_t waves = ((size_t)a.cam.W * a.cam.H + 63) / 64;
    int gx = (int)((waves + 3) / 4);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(mask_gather_kernel, dim3(gx, a.n_obj), dim3(256), 0, s, a, frames_between);
    if (finish) hipLaunchKernelGGL(mask_finish_kernel, dim3((a.n_obj + 63) / 64), dim3(64), 0, s, a, frames_between);
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

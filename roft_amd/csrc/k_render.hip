// k_render.hip -- depth render of the object mesh + masked depth likelihood + outlier decision
// (gfx950).
//
// Reference:
//   ROFTFilter::correct_outlier_rejection / pick_best_alternative / buffer_outlier_rejection_features
//                                               src/roft-lib/src/ROFTFilter.cpp:467-676
//   SICAD::superimpose (OpenGL) contract         src/roft-lib/src/SICAD.cpp:924-1066,1601-1656,
//                                               src/roft-lib/shader/shader_model.frag:30-52
//
// MI355X design.
//  * No OpenGL: a compute rasteriser, one thread per triangle, min-depth resolved with atomicMin on
//    the IEEE bits of the (positive) eye-space Z -- order independent, so the tile is bit-identical
//    from run to run.  Per-pixel arithmetic is IEEE float with the operation order of the render
//    contract in oracle/ro_render.c (this file is built with -ffp-contract=off).
//  * The reference copies the whole depth frame and mask when a pose arrives and scans them six
//    frames later (findNonZero, every second pixel).  Here the "features" are extracted once into
//    a compact (pixel, depth) list indexed by rank/2, so the likelihood is a dense reduction over
//    ~N_mask/2 samples and the frame itself need not be retained.
#include <algorithm>
#include <atomic>

#include "plane_rank.h"

namespace roft {

constexpr int kFeatThreads = 1024;

// The reference buffers depth + mask when a pose arrives (ROFTFilter.cpp:313-322, :353) and tests the NEXT pose against
// them; here the buffered sets live in a small ring (FrameCtrl::feat_write / feat_read name the slots), so buffering
// is part of the mask chain of the frame and never waits for the pose chain that reads an older set.
// Feature slot s holds the pixel of row-major rank 2s of the current obj plane (`k += 2` over the
// findNonZero list, ROFTFilter.cpp:556) and its depth.  Plane words are in row-major order, so one block
// scan over the popcounts of contiguous word chunks gives every chunk its starting rank; the expansion of the
// words into (pixel, depth) slots is described at the loop below.
#ifdef ROFT_FEAT_PROFILE
#define FTICK(i) do { __syncthreads(); if (threadIdx.x == 0) { long long _t = wall_clock64(); a.state[obj].dbg[i] = _t - f_t0; f_t0 = _t; } } while (0)
#else
#define FTICK(i) do {} while (0)
#endif

// LDS = false: planes that do not fit the LDS (beyond ~1.1 Mpixel) are read from memory by both passes.
// grid: (n_obj, frames listed in `frames_packed`: four bits per frame index, the batch's frames that buffer features for some object)
template <bool LDS>
__global__ __launch_bounds__(kFeatThreads) void features_kernel(EngineArrays a, unsigned frames_packed)
{
    ROFT_RESIDENT(a, RK_FEATURES);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_wave[16];
    __shared__ int s_chunk[kFeatThreads];
    const int obj = blockIdx.x;
    const FrameCtrl& c = frame_ctrl(a, (int)((frames_packed >> (4 * blockIdx.y)) & 15u), obj);
#ifdef ROFT_FEAT_PROFILE
    long long f_t0 = wall_clock64();
#endif
    int feat_write = c.feat_write, slot_cur = c.slot_cur;
    const float* depth = c.depth_cur;
    asm volatile("" : "+v"(feat_write), "+v"(slot_cur), "+v"(depth));   // all three fetched before the first barrier
    if (feat_write < 0) return;
    const int W = a.cam.W, wpr = a.cam.wpr;
    const uint32_t* gplane = a.planes + plane_offset(a, obj, slot_cur, 1);
    uint32_t* s_stage = reinterpret_cast<uint32_t*>(smem);   // staged with coalesced 16-byte loads
    if (LDS) {
        const size_t n4 = a.plane_words / 4;
        for (size_t i = threadIdx.x; i < n4; i += blockDim.x)
            reinterpret_cast<uint4*>(s_stage)[i] = reinterpret_cast<const uint4*>(gplane)[i];
        for (size_t i = n4 * 4 + threadIdx.x; i < a.plane_words; i += blockDim.x) s_stage[i] = gplane[i];
        __syncthreads();
    }
    const uint32_t* s_plane = LDS ? s_stage : gplane;
    FTICK(0);
    uint32_t* fpix = a.feat_pix + ((size_t)obj * kFeatRing + feat_write) * a.feat_cap;
    float* fdep = a.feat_depth + ((size_t)obj * kFeatRing + feat_write) * a.feat_cap;
    const int n_words = (int)a.plane_words;
    const int per = (n_words + blockDim.x - 1) / blockDim.x;
    const int w0 = min(n_words, (int)threadIdx.x * per), w1 = min(n_words, w0 + per);
    int cnt = 0;
    for (int w = w0; w < w1; ++w) cnt += __popc(s_plane[w]);
    int total;
    s_chunk[threadIdx.x] = block_exclusive_scan(cnt, s_wave, &total);   // starting rank of the thread's chunk
    __syncthreads();
    FTICK(1);
    // Expansion with the words dealt out round-robin -- the dense words of the mask are neighbours, so a contiguous
    // split would leave the whole expansion to a few threads.  A word's starting rank = its chunk's + the popcounts
    // of the chunk's earlier words; the bits whose rank is even are kept (parity alternates along the set bits).
    for (int w = threadIdx.x; w < n_words; w += blockDim.x) {
        uint32_t bits = s_plane[w];
        if (!bits) continue;
        const int chunk = w / per;
        int r0 = s_chunk[chunk];
        for (int v = chunk * per; v < w; ++v) r0 += __popc(s_plane[v]);
        if (r0 & 1) bits &= bits - 1;                 // first set bit has an odd rank: skip it
        int slot = (r0 + 1) >> 1;
        const int row = w / wpr;
        const uint32_t pix0 = ((uint32_t)row << 16) | (uint32_t)((w - row * wpr) * 32);   // (v << 16 | u): no division by W downstream
        while (bits) {
            if (slot < a.feat_cap) fpix[slot] = pix0 + (uint32_t)__builtin_ctz(bits);
            ++slot;
            bits &= bits - 1;                         // the kept bit ...
            bits &= bits - 1;                         // ... and its odd-ranked successor (no-op on 0)
        }
    }
    __syncthreads();  // fpix is written and read by this workgroup only
    FTICK(2);
    // depth gathers, kBatch per thread in flight together (a loop that consumes each index right after loading it
    // pays two memory latencies per element)
    constexpr int kBatch = 16;
    const int n = min((total + 1) / 2, a.feat_cap);
    for (int sb = threadIdx.x; sb < n; sb += kBatch * blockDim.x) {
        uint32_t px[kBatch];
        float d[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int sl = sb + u * (int)blockDim.x;
            px[u] = (sl < n) ? fpix[sl] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) d[u] = depth[(size_t)(px[u] >> 16) * W + (px[u] & 0xFFFFu)];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int sl = sb + u * (int)blockDim.x;
            if (sl < n) fdep[sl] = d[u];
        }
    }
    FTICK(3);
    if (threadIdx.x == 0) a.state[obj].n_feat[feat_write] = n;
}

// feat_frames: bit t = some object buffers features in frame t of the batch (0: every frame -- the operator level).  Only those
// frames get workgroups (round 6: a 1024-thread workgroup with the plane's 38 KB of LDS needs a place on a CU even to find out
// that its frame has nothing to do, and five of a batch's six frames have not).
void launch_features(const EngineArrays& a, hipStream_t s, hipEvent_t stop, unsigned feat_frames)
{
    unsigned packed = 0;
    int n_frames = 0;
    for (int t = 0; t < a.T && t < 8; ++t)
        if (feat_frames == 0 || (feat_frames & (1u << t))) packed |= (unsigned)t << (4 * n_frames++);
    if (n_frames == 0) n_frames = 1;   // (a.T == 0 cannot happen; the launch carries the caller's stop event: never skipped)
    const size_t lds = (a.plane_words * 4 + 15) & ~(size_t)15;
    const size_t lds_cap = 160 * 1024 - 256 - kFeatThreads * sizeof(int) - 128;
    if (lds <= lds_cap) {
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(features_kernel<true>), (int)lds_cap);
        hipExtLaunchKernelGGL(features_kernel<true>, dim3(a.n_obj, n_frames), dim3(kFeatThreads), (uint32_t)lds, s, nullptr, stop, 0, a, packed);
    } else {
        hipExtLaunchKernelGGL(features_kernel<false>, dim3(a.n_obj, n_frames), dim3(kFeatThreads), 0, s, nullptr, stop, 0, a, packed);
    }
}

// ---- rasteriser ---------------------------------------------------------------------------------
struct RenderPose {
    float R[9];
    float t[3];
};

__device__ __forceinline__ RenderPose make_pose(const double* x, const double* q)
{
    RenderPose p;
    const double w = q[0], qx = q[1], qy = q[2], qz = q[3];
    p.R[0] = (float)(1.0 - 2.0 * (qy * qy + qz * qz)); p.R[1] = (float)(2.0 * (qx * qy - w * qz)); p.R[2] = (float)(2.0 * (qx * qz + w * qy));
    p.R[3] = (float)(2.0 * (qx * qy + w * qz)); p.R[4] = (float)(1.0 - 2.0 * (qx * qx + qz * qz)); p.R[5] = (float)(2.0 * (qy * qz - w * qx));
    p.R[6] = (float)(2.0 * (qx * qz - w * qy)); p.R[7] = (float)(2.0 * (qy * qz + w * qx)); p.R[8] = (float)(1.0 - 2.0 * (qx * qx + qy * qy));
    for (int i = 0; i < 3; ++i) p.t[i] = (float)x[i];
    return p;
}

__device__ __forceinline__ void project_vertex(const float* v, const RenderPose& P, float fx, float fy, float cx,
                                               float cy, float& sx, float& sy, float& z)
{
    const float X = ((P.R[0] * v[0] + P.R[1] * v[1]) + P.R[2] * v[2]) + P.t[0];
    const float Y = ((P.R[3] * v[0] + P.R[4] * v[1]) + P.R[5] * v[2]) + P.t[1];
    z = ((P.R[6] * v[0] + P.R[7] * v[1]) + P.R[8] * v[2]) + P.t[2];
    if (z > 0.001f) {
        const float iz = 1.0f / z;   // (one reciprocal per vertex: the contract of oracle/ro_render.c)
        sx = (fx * X) * iz + cx;
        sy = (fy * Y) * iz + cy;
    } else {
        sx = sy = 0.0f;
    }
}

// Scan conversion of one projected triangle on a w x h target: `store(i, j, z)` receives every covered pixel with its
// eye-space depth (the render contract of oracle/ro_render.c, operation by operation); rows outside [j_lo, j_hi] are
// skipped (a strip of the target).  cull: 0 = draw; 1 / 2 = the mesh is a closed surface (mesh_class.h) and this triangle is
// wound counter-clockwise (1) / clockwise (2) seen from outside: it is drawn only if it faces the camera -- a counter-clockwise
// triangle that does has NEGATIVE screen area under the contract's projection (x right, y down, z forward).
// (The candidate pixels as ONE loop of columns x rows iterations instead of two nested ones -- rounds of a wave = its largest box
// instead of tallest x widest -- was measured in rounds 5 and 6, with and without the back-face rule: 15.6 against 15.0 us for the
// triangle phase, the compiler hoists the row terms of the edge functions out of the inner loop.  Nested it stays.)
template <class Store>
__device__ __forceinline__ void raster_projected(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2,
                                                 float z2, int w, int h, int j_lo, int j_hi, int cull, Store store)
{
    if (!(z0 > 0.001f && z1 > 0.001f && z2 > 0.001f)) return;
    const float area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    if (area == 0.0f || !(area == area)) return;
    if (cull && ((area < 0.0f) == (cull == 2))) return;   // faces away
    const float minx = fminf(x0, fminf(x1, x2)), maxx = fmaxf(x0, fmaxf(x1, x2));
    const float miny = fminf(y0, fminf(y1, y2)), maxy = fmaxf(y0, fmaxf(y1, y2));
    float fi0 = ceilf(minx - 0.5f), fi1 = floorf(maxx - 0.5f);
    float fj0 = ceilf(miny - 0.5f), fj1 = floorf(maxy - 0.5f);
    if (fi0 < 0.0f) fi0 = 0.0f;
    if (fj0 < 0.0f) fj0 = 0.0f;
    if (fi1 > (float)(w - 1)) fi1 = (float)(w - 1);
    if (fj1 > (float)(h - 1)) fj1 = (float)(h - 1);
    if (!(fi0 <= fi1) || !(fj0 <= fj1)) return;
    const int ia = (int)fi0, ib = (int)fi1, ja = max((int)fj0, j_lo), jb = min((int)fj1, j_hi);
    if (jb < ja) return;
    // perspective-correct depth of a covered pixel as ONE quotient (oracle/ro_render.c):
    //   z = area z0 z1 z2 / (w0 z1 z2 + w1 z0 z2 + w2 z0 z1)
    // -- five multiplications per triangle, three multiply-adds and a division per pixel; no reciprocal of a vertex depth, no
    // normalised barycentric weights (an IEEE division is ~10 instructions and the kernel is bound by instruction issue)
    const float p12 = z1 * z2, p02 = z0 * z2, p01 = z0 * z1;
    const float num = area * (z0 * p12);
    for (int j = ja; j <= jb; ++j) {
        const float py = (float)j + 0.5f;
        for (int i = ia; i <= ib; ++i) {
            const float px = (float)i + 0.5f;
            const float w0 = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1);
            const float w1 = (x0 - x2) * (py - y2) - (y0 - y2) * (px - x2);
            const float w2 = (x1 - x0) * (py - y0) - (y1 - y0) * (px - x0);
            const bool inside = (area > 0.0f) ? (w0 >= 0.0f && w1 >= 0.0f && w2 >= 0.0f)
                                              : (w0 <= 0.0f && w1 <= 0.0f && w2 <= 0.0f);
            if (!inside) continue;
            const float den = (w0 * p12 + w1 * p02) + w2 * p01;
            const float z = num / den;
            if (!(z > 0.0f)) continue;
            store(i, j, z);
        }
    }
}

// Operator level (roft_depth_likelihood): likelihood of both alternatives over the buffered features from z-buffers in
// HBM (EngineArrays::zbuf), decision, and the selected belief becomes the corrected belief (ROFTFilter.cpp:581-583,
// 670-675).  One workgroup per object.  The engine's own test is outlier_fused_kernel below.
constexpr int kOutlierThreads = 1024;

__global__ __launch_bounds__(kOutlierThreads) void outlier_kernel(EngineArrays a, int lin)
{
    __shared__ double s_err[2][kOutlierThreads / 64];
    __shared__ double s_cnt[2][kOutlierThreads / 64];
    __shared__ int s_sel;
    const int obj = blockIdx.x;
    ObjState& st = a.state[obj];
    PoseLane& pl = st.lane[lin];
    if (pl.pending_frame < 0) return;
    const FrameCtrl& c = frame_ctrl(a, pl.pending_frame, obj);
    const int d = a.cam.divider, tw = a.tile_w;
    const int fslot = (c.feat_read >= 0) ? c.feat_read : 0;
    const uint32_t* fpix = a.feat_pix + ((size_t)obj * kFeatRing + fslot) * a.feat_cap;
    const float* fdep = a.feat_depth + ((size_t)obj * kFeatRing + fslot) * a.feat_cap;
    const uint32_t* z0 = a.zbuf + (((size_t)lin * a.n_obj + obj) * 2) * a.tile_w * a.tile_h;
    const uint32_t* z1 = z0 + (size_t)a.tile_w * a.tile_h;
    double err[2] = {0.0, 0.0}, cnt[2] = {0.0, 0.0};
    const int n = st.n_feat[fslot];
    // kBatch feature slots per thread and iteration: their loads are independent and go out together (one iteration
    // covers 16 k features; per-thread accumulation order is the slot order either way)
    constexpr int kBatch = 16;
    for (int i0 = threadIdx.x; i0 < n; i0 += kBatch * blockDim.x) {
        float dep[kBatch];
        uint32_t pix[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int i = i0 + k * blockDim.x;
            dep[k] = (i < n) ? fdep[i] : 0.0f;
            pix[k] = (i < n) ? fpix[i] : 0u;
        }
        uint32_t b0[kBatch], b1[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int v = (int)(pix[k] >> 16), u = (int)(pix[k] & 0xFFFFu);
            const size_t ti = (size_t)(v / d) * tw + (u / d);
            b0[k] = z0[ti];
            b1[k] = z1[ti];
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (!((dep[k] > 0) && ((double)dep[k] < 2.0))) continue;  // hard-coded 2.0 (ROFTFilter.cpp:561)
            if (b0[k] != 0x7F800000u) { err[0] += (double)fabsf(dep[k] - __uint_as_float(b0[k])); cnt[0] += 1.0; }
            if (b1[k] != 0x7F800000u) { err[1] += (double)fabsf(dep[k] - __uint_as_float(b1[k])); cnt[1] += 1.0; }
        }
    }
    for (int k = 0; k < 2; ++k) {
        double e = err[k], n2 = cnt[k];
        for (int off = 32; off > 0; off >>= 1) { e += __shfl_down(e, off, 64); n2 += __shfl_down(n2, off, 64); }
        if ((threadIdx.x & 63) == 0) { s_err[k][threadIdx.x >> 6] = e; s_cnt[k][threadIdx.x >> 6] = n2; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double L[2];
        for (int k = 0; k < 2; ++k) {
            double e = 0.0, n2 = 0.0;
            for (int w = 0; w < kOutlierThreads / 64; ++w) { e += s_err[k][w]; n2 += s_cnt[k][w]; }
            L[k] = (n2 == 0.0) ? 1.7976931348623157e308 : (e / n2) / 1.0;  // gain is a bool -> 1.0 (ROFTFilter.h:64)
        }
        const int sel = (L[0] > 2.0 * L[1]) ? 1 : 0;
        pl.outlier_selected = sel;
        pl.outlier_L[0] = L[0];
        pl.outlier_L[1] = L[1];
        for (int k = 0; k < 2; ++k) {
            double n2 = 0.0;
            for (int w = 0; w < kOutlierThreads / 64; ++w) n2 += s_cnt[k][w];
            pl.outlier_cnt[k] = n2;
        }
        s_sel = sel;
    }
    __syncthreads();
    const PoseBelief& src = st.belief[b_alt(lin, s_sel ? 1 : 0)];
    PoseBelief& dst = st.belief[c.cur_slot];
    for (int i = threadIdx.x; i < 144; i += blockDim.x) dst.cov[i] = src.cov[i];
    if (threadIdx.x < 13) dst.mean[threadIdx.x] = src.mean[threadIdx.x];
    if (roft_object_output* row = log_row(a, c, obj)) {
        if (threadIdx.x < 13) row->pose[threadIdx.x] = src.mean[threadIdx.x];
        if (threadIdx.x == 0) {
            row->outlier_selected = s_sel;
            row->outlier_L[0] = pl.outlier_L[0];
            row->outlier_L[1] = pl.outlier_L[1];
        }
    }
}

// ---- the engine's outlier test: depth render + likelihood of ONE alternative per workgroup, all in LDS ------------------
// The render is only ever sampled at the buffered feature pixels, and the object covers a small window of the render
// target (about 70 x 90 of 320 x 240 pixels at the metric shape).  So: the mesh vertices are projected once per
// alternative into LDS (a vertex is shared by six triangles), their pixel bounding box is the z window -- also in LDS,
// resolved with LDS atomicMin, +inf = empty --, and the likelihood reads it in place.  No z-buffer in HBM, no clear
// pass, no global atomics, one launch instead of three.  A window that does not fit the LDS next to the vertices is
// rendered in horizontal strips (every strip walks all triangles and all features); a mesh whose vertices do not fit is
// projected per triangle.  The per-pixel arithmetic is raster_projected's, whatever the strips, bands and the vertex cache: the depths are bit-identical
// to oracle/ro_render.c in every configuration (tests/test_parity_gpu.py drives this kernel through roft_render_depth and
// roft_outlier_test).
// With CUs to spare an alternative is shared by `parts` workgroups, as R bands of the window's rows x G groups of its
// triangles (round 4; R G = parts, R = the fewest bands that fit the LDS in one piece -- 1 for most objects).  The G workgroups of
// a band each draw THEIR triangles (runs of 64, dealt out in turn) into a window of their own, write it through to a slab in
// memory (agent-coherent stores, EngineArrays::zmerge), count themselves in; the one that arrives last reads the other slabs
// back (agent-coherent loads), keeps the nearest depth per pixel -- the same minimum, so the same bits, as one window would
// hold -- and scores the band.  Every band leaves its partial sums; the pose chain segment that follows adds them up in band
// order and decides.  (G = 1: the row split of rounds 2 - 3, every workgroup walks all triangles; the launch falls back to it
// when the slabs do not cover objects x parts, roft_debug_outlier_split(0) forces it.)  The launch lasts as long as its slowest
// workgroup, which in the row split is the middle band of the largest object (49 - 56 us of triangles at 8 and 16 objects against
// 25 - 30 with the triangles dealt out: 60 - 66 -> 45 - 46 us per launch).
// grid: (2 alternatives x parts, n_obj).  dynamic LDS: [3 * vcache_cap floats] | [win_cap z values]
constexpr int kFusedThreads = 1024;
// phase stamps (-DROFT_FUSED_PROFILE; PHASES=fused tools/k1_phase_profile.py): 100 MHz ticks -> ObjState::dbg[alt * 8 + phase]
#ifdef ROFT_FUSED_PROFILE
#define UTICK(i) do { __syncthreads(); if (threadIdx.x == 0 && bx < 4) { long long _t = wall_clock64(); st.dbg[bx * 8 + (i)] += _t - u_t0; u_t0 = _t; } } while (0)
#else
#define UTICK(i) do {} while (0)
#endif

__global__ __launch_bounds__(kFusedThreads) void outlier_fused_kernel(EngineArrays a, int lin, int vcache_cap, int win_cap, int parts,
                                                                     int split_tris, float* tile_dump)
{
    ROFT_RESIDENT(a, RK_OUTLIER);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ long long s_hi[kFusedThreads / 64], s_lo[kFusedThreads / 64];
    __shared__ int s_cnt[kFusedThreads / 64];
    __shared__ int s_box[4];
    __shared__ int s_behind;   // some vertex is not in front of the near plane: a closed mesh is then drawn whole too
    __shared__ int s_last;
    // Workgroups are handed to the XCDs round robin by their linear index; the 2 x parts workgroups of an object read the
    // same mesh (186 KB of indices + 98 KB of vertices at the bench's 15.5 k triangles): with the grid laid out as
    // [group of 8 objects][workgroup of the object][object within the group] they share one XCD and one L2.
    const int per = 2 * parts, lin_id = (int)blockIdx.x;
    const int obj = (lin_id / (8 * per)) * 8 + (lin_id & 7), bx = (lin_id >> 3) % per;
    if (obj >= a.n_obj) return;
    const int alt = bx / parts, part = bx % parts, tid = threadIdx.x;
    ObjState& st = a.state[obj];
    PoseLane& pl = st.lane[lin];
    if (pl.pending_frame < 0) return;   // no outlier test pending between the pose chain segments
    const FrameCtrl& c = frame_ctrl(a, pl.pending_frame, obj);
    const ObjParams& prm = a.params[obj];
    const PoseBelief& bl = st.belief[b_alt(lin, alt)];
    const RenderPose P = make_pose(bl.mean + 6, bl.mean + 9);
    const int d = a.cam.divider, tw = a.tile_w, th = a.tile_h;
    const float fx = (float)(a.cam.fx / d), fy = (float)(a.cam.fy / d), cx = (float)(a.cam.cx / d), cy = (float)(a.cam.cy / d);
    const int nv = prm.n_verts, nt = prm.n_tris;
    const bool cached = nv <= vcache_cap;
    float* s_v = reinterpret_cast<float*>(smem);
    uint32_t* s_z = reinterpret_cast<uint32_t*>(smem + (((size_t)vcache_cap * 12 + 15) & ~(size_t)15));
    if (tid < 4) s_box[tid] = (tid < 2) ? INT32_MAX : -1;
    if (tid == 0) s_behind = 0;
#ifdef ROFT_FUSED_PROFILE
    long long u_t0 = wall_clock64();
    if (tid < 8 && bx < 4) st.dbg[bx * 8 + tid] = 0;
#endif
    __syncthreads();
    // vertices -> screen; pixel bounding box of the triangles that can be drawn (pixel ranges as raster_projected clips
    // them: ceil(min - 0.5) .. floor(max - 0.5) are monotone, so the box of the vertices covers every triangle)
    {
        int bi0 = INT32_MAX, bj0 = INT32_MAX, bi1 = -1, bj1 = -1;
        bool behind = false;
        constexpr int kVB = 4;   // vertices per thread whose coordinates are fetched together
        for (int vb = tid; vb < nv; vb += kVB * kFusedThreads) {
          float vc[kVB][3];
#pragma unroll
          for (int k = 0; k < kVB; ++k) {
              const int v = min(vb + k * kFusedThreads, nv - 1);
#pragma unroll
              for (int q = 0; q < 3; ++q) vc[k][q] = prm.verts[(size_t)3 * v + q];
          }
#pragma unroll
          for (int k = 0; k < kVB; ++k) {
            const int v = vb + k * kFusedThreads;
            if (v >= nv) break;
            float sx, sy, z;
            project_vertex(vc[k], P, fx, fy, cx, cy, sx, sy, z);
            if (cached) { s_v[3 * v] = sx; s_v[3 * v + 1] = sy; s_v[3 * v + 2] = z; }
            if (z > 0.001f) {
                // (a vertex far outside the target clamps to an empty or full range; float -> int saturates)
                const float lo_i = fminf(fmaxf(ceilf(sx - 0.5f), 0.0f), (float)tw), hi_i = fminf(fmaxf(floorf(sx - 0.5f), -1.0f), (float)(tw - 1));
                const float lo_j = fminf(fmaxf(ceilf(sy - 0.5f), 0.0f), (float)th), hi_j = fminf(fmaxf(floorf(sy - 0.5f), -1.0f), (float)(th - 1));
                bi0 = min(bi0, (int)lo_i); bi1 = max(bi1, (int)hi_i);
                bj0 = min(bj0, (int)lo_j); bj1 = max(bj1, (int)hi_j);
            } else {
                behind = true;
            }
          }
        }
        if (__any(behind) && (tid & 63) == 0) atomicOr(&s_behind, 1);
        for (int off = 32; off > 0; off >>= 1) {
            bi0 = min(bi0, __shfl_xor(bi0, off, 64)); bj0 = min(bj0, __shfl_xor(bj0, off, 64));
            bi1 = max(bi1, __shfl_xor(bi1, off, 64)); bj1 = max(bj1, __shfl_xor(bj1, off, 64));
        }
        if ((tid & 63) == 0) {
            atomicMin(&s_box[0], bi0); atomicMin(&s_box[1], bj0);
            atomicMax(&s_box[2], bi1); atomicMax(&s_box[3], bj1);
        }
    }
    __syncthreads();
    // (a triangle's pixels lie between the smallest lower bound and the largest upper bound of its vertices; vertices
    //  left or above the target contribute lower bound 0, vertices right or below it the upper bound w - 1 / h - 1)
    UTICK(0);
    const int i0 = min(s_box[0], tw - 1), i1 = s_box[2];
    // The `parts` workgroups of the alternative share its work as R bands of the window's rows x G groups of its triangles
    // (R G = parts).  Rows only (G = 1): every workgroup walks all triangles and draws the rows of its band.  With triangle
    // groups, the G workgroups of a band each draw their share of the triangles into a window of their own, the windows are
    // merged in memory and the workgroup that arrives last scores the band: R is the smallest divisor of `parts` whose bands
    // fit the LDS in one piece (1 for most objects: set-up and walk are divided by `parts`, nothing is done twice but the
    // projection).  Every workgroup of the alternative sees the same box and decides alike.
    int j0 = min(s_box[1], th - 1), j1 = s_box[3];
    int R = parts;
    if (split_tris && parts > 1 && i1 >= i0 && j1 >= j0 && i0 >= 0 && j0 >= 0) {
        const long long w_all = i1 - i0 + 1, rows_all = j1 - j0 + 1;
        for (R = 1; R < parts; ++R)
            if (parts % R == 0 && ((rows_all + R - 1) / R) * w_all <= (long long)win_cap) break;
    }
    const int G = parts / R, band = part / G, grp = part % G;
    const bool split = G > 1;
    if (j1 >= j0 && R > 1) {
        const int rows_all = j1 - j0 + 1, b0 = j0 + (int)((long long)rows_all * band / R), b1 = j0 + (int)((long long)rows_all * (band + 1) / R) - 1;
        j0 = b0;
        j1 = b1;
    }
    const int win_w = i1 - i0 + 1;
    const int fslot = (c.feat_read >= 0) ? c.feat_read : 0;
    const uint32_t* fpix = a.feat_pix + ((size_t)obj * kFeatRing + fslot) * a.feat_cap;
    const float* fdep = a.feat_depth + ((size_t)obj * kFeatRing + fslot) * a.feat_cap;
    const int n = st.n_feat[fslot];
    LikelihoodSum err;
    int cnt = 0;
    if (win_w > 0 && j1 >= j0 && i0 >= 0 && j0 >= 0) {
        const int rows = max(1, win_cap / win_w);   // (win_cap >= the target's width: a strip holds at least one row)
        for (int js = j0; js <= j1; js += rows) {
            const int je = min(j1, js + rows - 1), npx = win_w * (je - js + 1);
            for (int i = tid; i < npx; i += kFusedThreads) s_z[i] = 0x7F800000u;
            __syncthreads();
            UTICK(1);
            ROFT_LDS uint32_t* const zw = pin_lds(s_z);
            auto store = [zw, i0, js, win_w](int i, int j, float z) {
                (void)__hip_atomic_fetch_min(zw + ((j - js) * win_w + (i - i0)), __float_as_uint(z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            constexpr int kTB = 8;   // triangles per thread whose vertex indices are fetched together
            // closed mesh, everything in front of the near plane: triangles that face away are left out (the render contract)
            const uint8_t* const flips = (prm.tri_flip && !s_behind && nv < (1 << 30)) ? prm.tri_flip : nullptr;   // (the cull code rides in an index's top bits)
            // (split: the triangles are dealt out to the workgroups of the alternative in runs of 64 -- one run per wave and
            //  fetch, so the index loads stay coalesced and neighbouring runs, which cost alike, go to different workgroups)
            const int stride = G, first = grp;
            auto tri_of = [=](int slot) { return ((slot >> 6) * stride + first) * 64 + (slot & 63); };   // slot: this workgroup's own numbering
            const int n_slots = split ? ((nt + 63) / 64 + stride - 1 - first) / stride * 64 : nt;        // (its runs; the last one may be short)
            for (int tb = tid; tb < n_slots; tb += kTB * kFusedThreads) {
              int idx[kTB][3];   // (idx[k][0] carries the triangle's cull code in its two top bits: no registers of its own)
#pragma unroll
              for (int k = 0; k < kTB; ++k) {
                  const int t = min(tri_of(min(tb + k * kFusedThreads, n_slots - 1)), nt - 1);
                  const int32_t* tri = prm.tris + (size_t)3 * t;
                  idx[k][0] = tri[0]; idx[k][1] = tri[1]; idx[k][2] = tri[2];
                  if (flips) idx[k][0] |= (1 + (int)flips[t]) << 30;
              }
#pragma unroll
              for (int k = 0; k < kTB; ++k) {
                if (tb + k * kFusedThreads >= n_slots || tri_of(tb + k * kFusedThreads) >= nt) break;
                const int cull_k = (int)((unsigned)idx[k][0] >> 30);
                const int v0 = idx[k][0] & 0x3FFFFFFF, v1 = idx[k][1], v2 = idx[k][2];
                float x0, y0, z0, x1, y1, z1, x2, y2, z2;
                if (cached) {
                    x0 = s_v[3 * v0]; y0 = s_v[3 * v0 + 1]; z0 = s_v[3 * v0 + 2];
                    x1 = s_v[3 * v1]; y1 = s_v[3 * v1 + 1]; z1 = s_v[3 * v1 + 2];
                    x2 = s_v[3 * v2]; y2 = s_v[3 * v2 + 1]; z2 = s_v[3 * v2 + 2];
                } else {
                    project_vertex(prm.verts + (size_t)3 * v0, P, fx, fy, cx, cy, x0, y0, z0);
                    project_vertex(prm.verts + (size_t)3 * v1, P, fx, fy, cx, cy, x1, y1, z1);
                    project_vertex(prm.verts + (size_t)3 * v2, P, fx, fy, cx, cy, x2, y2, z2);
                }
                raster_projected(x0, y0, z0, x1, y1, z1, x2, y2, z2, tw, th, js, je, cull_k, store);
              }
            }
            __syncthreads();
            UTICK(2);
            if (split) {
                // This workgroup's window -> its slab in memory, written through (agent-coherent stores: the workgroups of an
                // alternative may sit on different XCDs, each behind an L2 of its own); the workgroup that arrives last reads all
                // slabs back with agent-coherent loads, keeps the nearest depth of every pixel and goes on alone: dump, samples,
                // sums.  No atomics on the pixels, nothing to clear afterwards: every slab is rewritten whole by the next test.
                const size_t slab0 = (((size_t)lin * a.zmerge_slabs + (size_t)obj * parts + (size_t)band * G) * 2 + alt) * a.zmerge_stride;   // the band's first slab
                uint32_t* mine = a.zmerge + slab0 + (size_t)grp * 2 * a.zmerge_stride;
                int* arrivals = a.zcount + (((size_t)lin * a.n_obj + obj) * 2 + alt) * kMaxOutlierParts + band;
                for (int i = tid; i < npx; i += kFusedThreads)
                    __hip_atomic_store(&mine[i], s_z[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                __builtin_amdgcn_s_waitcnt(0);   // every store of this wave acknowledged
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                __syncthreads();
                if (tid == 0) {
                    const int before = __hip_atomic_fetch_add(arrivals, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_last = (before == G - 1) ? 1 : 0;
                    if (before == G - 1) __hip_atomic_store(arrivals, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                if (!s_last) return;
                // (agent-coherent loads, a pixel's G - 1 of them in flight together: as atomic loads the compiler waits for each one)
                static_assert(kMaxOutlierParts <= 8, "eight slabs per pixel");
                const uint32_t* others = a.zmerge + slab0;
                const size_t slab_step = 2 * a.zmerge_stride;
                // the other groups' slabs, compacted (G - 1 pointers); the loads and the wait that makes them valid are ONE asm
                // statement per pixel with early-clobber outputs (round 5, ADVICE: as separate statements the compiler was free to
                // move, spill or reuse the destination registers between the load and the s_waitcnt)
                const uint32_t* op[7];
                {
                    int k = 0;
#pragma unroll
                    for (int p = 0; p < 8; ++p)
                        if (p < G && p != grp && k < 7) op[k++] = others + (size_t)p * slab_step;
                    for (; k < 7; ++k) op[k] = others;
                }
                for (int i = tid; i < npx; i += kFusedThreads) {
                    uint32_t m = s_z[i];
                    uint32_t v0, v1, v2, v3, v4, v5, v6;
                    switch (G - 1) {   // (uniform)
                    case 1:
                        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v0) : "v"(op[0] + i) : "memory");
                        m = min(m, v0);
                        break;
                    case 2:
                        asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(v0), "=&v"(v1) : "v"(op[0] + i), "v"(op[1] + i) : "memory");
                        m = min(m, min(v0, v1));
                        break;
                    case 3:
                        asm volatile("global_load_dword %0, %3, off sc1\n\tglobal_load_dword %1, %4, off sc1\n\tglobal_load_dword %2, %5, off sc1\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(op[0] + i), "v"(op[1] + i), "v"(op[2] + i) : "memory");
                        m = min(min(m, v0), min(v1, v2));
                        break;
                    case 4:
                        asm volatile("global_load_dword %0, %4, off sc1\n\tglobal_load_dword %1, %5, off sc1\n\tglobal_load_dword %2, %6, off sc1\n\t"
                                     "global_load_dword %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(op[0] + i), "v"(op[1] + i), "v"(op[2] + i), "v"(op[3] + i) : "memory");
                        m = min(m, min(min(v0, v1), min(v2, v3)));
                        break;
                    default:   // 5 .. 7 others: seven loads (the spare pointers repeat the first slab: harmless under min)
                        asm volatile("global_load_dword %0, %7, off sc1\n\tglobal_load_dword %1, %8, off sc1\n\tglobal_load_dword %2, %9, off sc1\n\t"
                                     "global_load_dword %3, %10, off sc1\n\tglobal_load_dword %4, %11, off sc1\n\tglobal_load_dword %5, %12, off sc1\n\t"
                                     "global_load_dword %6, %13, off sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6)
                                     : "v"(op[0] + i), "v"(op[1] + i), "v"(op[2] + i), "v"(op[3] + i), "v"(op[4] + i), "v"(op[5] + i), "v"(op[6] + i) : "memory");
                        m = min(min(m, v0), min(min(v1, v2), min(min(v3, v4), min(v5, v6))));
                        break;
                    case 0: break;
                    }
                    s_z[i] = m;
                }
                __syncthreads();
            }
            // operator level (roft_render_depth / roft_outlier_test): the strip of the window as this workgroup drew it ->
            // the caller's (zero-filled) render tile of the alternative, 0 = background as the reference reads it back
            if (tile_dump) {
                float* tile = tile_dump + (size_t)alt * tw * th;
                for (int i = tid; i < npx; i += kFusedThreads) {
                    const uint32_t b = s_z[i];
                    tile[(size_t)(js + i / win_w) * tw + (i0 + i % win_w)] = (b == 0x7F800000u) ? 0.0f : __uint_as_float(b);
                }
            }
            // likelihood samples of this strip (feature slots in ascending order per thread, as outlier_kernel adds them)
            constexpr int kBatch = 16;
            for (int f0 = tid; f0 < n; f0 += kBatch * kFusedThreads) {
                float dep[kBatch];
                uint32_t pix[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    const int i = f0 + k * kFusedThreads;
                    dep[k] = (i < n) ? fdep[i] : 0.0f;
                    pix[k] = (i < n) ? fpix[i] : 0u;
                }
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    if (!((dep[k] > 0) && ((double)dep[k] < 2.0))) continue;  // hard-coded 2.0 (ROFTFilter.cpp:561)
                    const int v = (int)(pix[k] >> 16), u = (int)(pix[k] & 0xFFFFu);
                    const int tj = v / d, ti = u / d;
                    if (tj < js || tj > je || ti < i0 || ti > i1) continue;
                    const uint32_t b = s_z[(tj - js) * win_w + (ti - i0)];
                    if (b != 0x7F800000u) { err.add(fabsf(dep[k] - __uint_as_float(b))); cnt += 1; }
                }
            }
            __syncthreads();   // the next strip clears the window
            UTICK(3);
#ifdef ROFT_FUSED_PROFILE
            if (tid == 0 && bx < 4) st.dbg[bx * 8 + 4] += 1;
            if (tid == 0 && bx < 4) st.dbg[bx * 8 + 5] = (long long)win_w * 100 ;
            if (tid == 0 && bx < 4) st.dbg[bx * 8 + 6] = (long long)(j1 - j0 + 1) * 100;
#endif
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        err.hi += __shfl_down(err.hi, off, 64);
        err.lo += __shfl_down(err.lo, off, 64);
        cnt += __shfl_down(cnt, off, 64);
    }
    if ((tid & 63) == 0) { s_hi[tid >> 6] = err.hi; s_lo[tid >> 6] = err.lo; s_cnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        long long hi = 0, lo = 0;
        int n2 = 0;
        for (int w = 0; w < kFusedThreads / 64; ++w) { hi += s_hi[w]; lo += s_lo[w]; n2 += s_cnt[w]; }
        // (one workgroup per band gets here -- with triangle groups the one that merged the band -- with the sums of its rows)
        pl.part_hi[alt][band] = hi;
        pl.part_lo[alt][band] = lo;
        pl.part_cnt[alt][band] = (double)n2;
        if (band == 0) pl.n_parts[alt] = R;
    }
}

static std::atomic<int> g_outlier_split{-1};   // roft_debug_outlier_split
void set_outlier_split(int mode) { g_outlier_split.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }

void launch_outlier(const EngineArrays& a, int lin, hipStream_t s, hipEvent_t stop, const OutlierLaunchOpts* opts)
{
    // objects that do not test this frame return immediately; the decision (ROFTFilter.cpp:581-583) is taken by the
    // pose chain segment that follows (ukf_chain_kernel)
    const size_t lds_total = 160 * 1024 - 4096;
    const size_t vbytes = ((size_t)a.max_verts * 12 + 15) & ~(size_t)15;
    // cache the projected vertices when they leave room for a window of 8 k pixels (a window that large or larger is
    // rendered in strips) and for the widest row of the target
    const size_t min_win = (size_t)4 * std::max(8192, a.tile_w);
    const bool cache = vbytes + min_win <= lds_total && !(opts && opts->no_vertex_cache);
    const int vcache_cap = cache ? a.max_verts : 0;
    int win_cap = (int)((lds_total - (cache ? vbytes : 0)) / 4);
    (void)set_max_dynamic_lds(reinterpret_cast<const void*>(outlier_fused_kernel), (int)lds_total);
    // bands per alternative: as many workgroups as the chip has CUs to spare
    int parts = std::max(1, std::min(kMaxOutlierParts, device_cu_count() / (2 * std::max(a.n_obj, 1))));
    if (a.outlier_parts > 0) parts = std::min(a.outlier_parts, kMaxOutlierParts);
    if (opts && opts->parts > 0) parts = std::min(opts->parts, kMaxOutlierParts);
    if (opts && opts->parts < 0) parts = std::max(1, parts / -opts->parts);
    // Several workgroups per alternative share its TRIANGLES (R bands x G triangle groups, decided per alternative inside the
    // kernel from the size of its window: see there) unless the caller asks for rows only.
    static const int split_env = getenv("ROFT_OUTLIER_SPLIT") ? atoi(getenv("ROFT_OUTLIER_SPLIT")) : -1;   // (experiments)
    const int forced = g_outlier_split.load(std::memory_order_relaxed) >= 0 ? g_outlier_split.load(std::memory_order_relaxed) : split_env;
    int split = (opts && opts->split >= 0) ? opts->split : (forced >= 0 ? forced : 1);
    if (parts <= 1 || !a.zmerge || (size_t)a.n_obj * parts > a.zmerge_slabs) split = 0;
    // (a band is a fraction of the window: request only the LDS it can need, so that other chains' workgroups fit next to it)
    const int lds_parts = split ? 1 : parts;
    const size_t win_need = (size_t)4 * std::max((size_t)a.tile_w, ((size_t)a.tile_w * a.tile_h + lds_parts - 1) / lds_parts + (size_t)a.tile_w);
    const size_t lds = std::min(lds_total, (((cache ? vbytes : 0) + win_need + 15) & ~(size_t)15));
    win_cap = std::min(win_cap, (int)((lds - (cache ? vbytes : 0)) / 4));
    // (operator level: a smaller window forces the strip path)
    if (opts && opts->window_pixels > 0) win_cap = std::max(a.tile_w, std::min(win_cap, opts->window_pixels));
    hipExtLaunchKernelGGL(outlier_fused_kernel, dim3(2 * parts * ((a.n_obj + 7) / 8) * 8), dim3(kFusedThreads), (uint32_t)lds, s, nullptr, stop, 0, a, lin,
                          vcache_cap, win_cap, parts, split, opts ? opts->tile_dump : nullptr);
}

void launch_outlier_only(const EngineArrays& a, hipStream_t s)
{
    hipLaunchKernelGGL(outlier_kernel, dim3(a.n_obj), dim3(kOutlierThreads), 0, s, a, 0);
}

}  // namespace roft

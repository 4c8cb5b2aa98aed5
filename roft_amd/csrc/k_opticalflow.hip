// k_opticalflow.hip -- dense optical-flow PRODUCER: pyramidal Lucas-Kanade on gfx950.
//
// The reference gets its flow frames from NVIDIA's fixed-function optical-flow engine
// (src/roft-lib/src/ImageOpticalFlowNVOF.cpp:100-159, tools/nvof/dumper/src/main.cpp:40-146); MI355X has no such
// unit, so the flow that feeds the filter is computed here.  The contract kept from the reference is the product:
// forward flow of frame k-1 pixels towards frame k as CV_32FC2 at grid 1 (NVOF 2.0 shape) or CV_16SC2 S10.5 at
// grid 4 (NVOF 1.0 shape, ImageOpticalFlowNVOF.cpp:19-80).  The algorithm is specified in oracle/ro_opticalflow.c
// (edge-extended images, central-difference gradients of the previous frame, (2r+1)^2 window, coarse-to-fine with
// nearest-neighbour x2 initialisation) and this file follows it tap for tap.
//
// MI355X mapping: one workgroup = a 32 x 8 pixel tile of one pyramid level of one image pair; the previous image
// tile with an (r + 1)-pixel halo and its gradient tile (Ix, Iy interleaved) are staged in LDS once and serve the G-matrix and the
// residual taps of all 256 pixels; only the pixels of the current image under the warped window (whose position
// depends on the evolving flow) go to the vector cache.  Batched over image pairs in blockIdx.z.
#include <algorithm>

#include "opticalflow.h"
#include "roft_device.h"

namespace roft {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// level 0: u8 -> float, four pixels per thread
__global__ __launch_bounds__(256) void of_convert_kernel(OfArgs a)
{
    const int pair = blockIdx.z, which = blockIdx.y;
    const uint8_t* src = which ? a.cur[pair] : a.prev[pair];
    float* dst = a.pyr + ((size_t)pair * 2 + which) * a.pyr_stride;
    const int n4 = a.lv[0].w * a.lv[0].h / 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const uchar4 v = reinterpret_cast<const uchar4*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
    }
}

__global__ __launch_bounds__(256) void of_down_kernel(OfArgs a, int l)
{
    const int pair = blockIdx.z, which = blockIdx.y;
    float* base = a.pyr + ((size_t)pair * 2 + which) * a.pyr_stride;
    const float* src = base + a.lv[l - 1].off;
    float* dst = base + a.lv[l].off;
    const int w = a.lv[l].w, h = a.lv[l].h, ws = a.lv[l - 1].w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        const size_t i0 = (size_t)(2 * y) * ws + 2 * x, i1 = i0 + ws;
        dst[i] = 0.25f * (((src[i0] + src[i0 + 1]) + src[i1]) + src[i1 + 1]);
    }
}

__device__ __forceinline__ float at_g(const float* I, int w, int h, int x, int y)
{
    return I[(size_t)clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)];
}

constexpr int kOfTx = 32, kOfTy = 8;

// One pyramid level.  RT = compile-time window radius (1..4), or 0 for the generic runtime-radius version.
// LDS: the previous-image tile with an (r + 1) halo, and from it the gradient tiles Ix, Iy with an r halo, so that a
// tap costs three LDS reads.  The current image is warped with ONE pair of bilinear weights per pixel and iteration
// (the displacement is constant over the window): the (2r+2)^2 pixels under the window are loaded once, combined
// row by row (horizontal pass in registers, vertical pass against the previous row) -- 64 cached loads per iteration
// at r = 3 instead of 4 per tap.
template <int RT>
__global__ __launch_bounds__(kOfTx * kOfTy) void of_lk_kernel(OfArgs a, int l)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = RT ? RT : a.radius, R = r + 1;
    const int tw = kOfTx + 2 * R, th = kOfTy + 2 * R;      // I0 tile
    const int gw = kOfTx + 2 * r, gh = kOfTy + 2 * r;      // gradient tiles
    float* tile = reinterpret_cast<float*>(smem);
    float2* grad = reinterpret_cast<float2*>(tile + ((tw * th + 1) & ~1));   // (Ix, Iy) interleaved: one 8-byte LDS read per tap
    const int pair = blockIdx.z;
    const int w = a.lv[l].w, h = a.lv[l].h;
    const float* I0 = a.pyr + ((size_t)pair * 2 + 0) * a.pyr_stride + a.lv[l].off;
    const float* I1 = a.pyr + ((size_t)pair * 2 + 1) * a.pyr_stride + a.lv[l].off;
    const int x0 = blockIdx.x * kOfTx, y0 = blockIdx.y * kOfTy;
    const int tid = threadIdx.y * kOfTx + threadIdx.x;
    for (int i = tid; i < tw * th; i += kOfTx * kOfTy) {
        const int ty = i / tw, tx = i - ty * tw;
        tile[i] = at_g(I0, w, h, x0 - R + tx, y0 - R + ty);   // edge-extended previous image
    }
    __syncthreads();
    for (int i = tid; i < gw * gh; i += kOfTx * kOfTy) {
        const int ty = i / gw, tx = i - ty * gw;
        const float* t = tile + (ty + 1) * tw + (tx + 1);
        grad[i] = make_float2(0.5f * (t[1] - t[-1]), 0.5f * (t[tw] - t[-tw]));
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= w || y >= h) return;

    float dx = 0.0f, dy = 0.0f;
    if (l < a.levels - 1) {
        const float* c = a.coarse + (size_t)pair * a.flow_stride + a.flow_off[l + 1] +
                         2 * ((size_t)(y >> 1) * a.lv[l + 1].w + (x >> 1));
        dx = 2.0f * c[0];
        dy = 2.0f * c[1];
    }
    // window origin inside the tiles: tap (ox, oy) -> gradient tile (threadIdx + r + o), I0 tile (threadIdx + R + o)
    const float2* g0 = grad + threadIdx.y * gw + threadIdx.x;
    const float* t0 = tile + (threadIdx.y + 1) * tw + (threadIdx.x + 1);
    const int n = 2 * r + 1;
    float g11 = 0.0f, g12 = 0.0f, g22 = 0.0f;
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j) {
            const float2 g = g0[k * gw + j];
            g11 = fmaf(g.x, g.x, g11); g12 = fmaf(g.x, g.y, g12); g22 = fmaf(g.y, g.y, g22);
        }
    const float det = g11 * g22 - g12 * g12;
    if (det > a.det_min) {
        const float inv = 1.0f / det;
        constexpr int NMAX = RT ? 2 * RT + 1 : 15;
        for (int it = 0; it < a.iterations; ++it) {
            const float xf = (float)x + dx, yf = (float)y + dy;
            const float fx = floorf(xf), fy = floorf(yf);
            const float ax = xf - fx, ay = yf - fy;
            const int qx = (int)fminf(fmaxf(fx, -64.0f), (float)w + 64.0f) - r;
            const int qy = (int)fminf(fmaxf(fy, -64.0f), (float)h + 64.0f) - r;
            float b1 = 0.0f, b2 = 0.0f;
            float hp[NMAX];   // horizontal pass of the previous row
#pragma unroll 1
            for (int k = 0; k <= n; ++k) {
                const float* row = I1 + (size_t)clampi(qy + k, 0, h - 1) * w;
                float hc[NMAX];
                float left = row[clampi(qx, 0, w - 1)];
#pragma unroll
                for (int j = 0; j < NMAX; ++j) {
                    if (j >= n) break;
                    const float right = row[clampi(qx + j + 1, 0, w - 1)];
                    hc[j] = fmaf(ax, right, (1.0f - ax) * left);
                    left = right;
                }
                if (k > 0) {
#pragma unroll
                    for (int j = 0; j < NMAX; ++j) {
                        if (j >= n) break;
                        const int o = (k - 1) * gw + j;
                        const float dt = fmaf(ay, hc[j], (1.0f - ay) * hp[j]) - t0[(k - 1) * tw + j];
                        const float2 g = g0[o];
                        b1 = fmaf(g.x, dt, b1); b2 = fmaf(g.y, dt, b2);
                    }
                }
#pragma unroll
                for (int j = 0; j < NMAX; ++j) hp[j] = hc[j];
            }
            dx -= (g22 * b1 - g12 * b2) * inv;
            dy -= (g11 * b2 - g12 * b1) * inv;
        }
    }
    float* dst = (l == 0) ? a.out_f32[pair] : a.coarse + (size_t)pair * a.flow_stride + a.flow_off[l];
    reinterpret_cast<float2*>(dst)[(size_t)y * w + x] = make_float2(dx, dy);
}

// CV_16SC2 grid 4: block-centre sample, S10.5, saturated
__global__ __launch_bounds__(256) void of_quantise_kernel(const float* const* field, int16_t* const* out, int W, int H)
{
    const int pair = blockIdx.y;
    const int gw = W / 4, gh = H / 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= gw * gh) return;
    const int gy = i / gw, gx = i - gy * gw;
    const float2 v = reinterpret_cast<const float2*>(field[pair])[(size_t)(4 * gy + 2) * W + (4 * gx + 2)];
    const float qx = fminf(fmaxf(rintf(v.x * 32.0f), -32768.0f), 32767.0f);
    const float qy = fminf(fmaxf(rintf(v.y * 32.0f), -32768.0f), 32767.0f);
    reinterpret_cast<short2*>(out[pair])[i] = make_short2((short)qx, (short)qy);
}

void launch_optical_flow(const OfArgs& a, hipStream_t s)
{
    const int n4 = a.lv[0].w * a.lv[0].h / 4;
    hipLaunchKernelGGL(of_convert_kernel, dim3(std::min((n4 + 255) / 256, 256), 2, a.n), dim3(256), 0, s, a);
    for (int l = 1; l < a.levels; ++l) {
        const int npx = a.lv[l].w * a.lv[l].h;
        hipLaunchKernelGGL(of_down_kernel, dim3(std::min((npx + 255) / 256, 256), 2, a.n), dim3(256), 0, s, a, l);
    }
    const int r = a.radius, R = r + 1;
    const size_t lds = ((((size_t)(kOfTx + 2 * R) * (kOfTy + 2 * R) + 1) & ~(size_t)1) + 2 * (size_t)(kOfTx + 2 * r) * (kOfTy + 2 * r)) * sizeof(float);
    for (int l = a.levels - 1; l >= 0; --l) {
        const dim3 grid((a.lv[l].w + kOfTx - 1) / kOfTx, (a.lv[l].h + kOfTy - 1) / kOfTy, a.n), block(kOfTx, kOfTy);
        switch (r) {
            case 1: hipLaunchKernelGGL(of_lk_kernel<1>, grid, block, lds, s, a, l); break;
            case 2: hipLaunchKernelGGL(of_lk_kernel<2>, grid, block, lds, s, a, l); break;
            case 3: hipLaunchKernelGGL(of_lk_kernel<3>, grid, block, lds, s, a, l); break;
            case 4: hipLaunchKernelGGL(of_lk_kernel<4>, grid, block, lds, s, a, l); break;
            default: hipLaunchKernelGGL(of_lk_kernel<0>, grid, block, lds, s, a, l); break;
        }
    }
}

void launch_flow_quantise(const float* const* field, int16_t* const* out, int n, int W, int H, hipStream_t s)
{
    hipLaunchKernelGGL(of_quantise_kernel, dim3(((W / 4) * (H / 4) + 255) / 256, n), dim3(256), 0, s, field, out, W, H);
}

}  // namespace roft

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
// tile with an (r + 1)-pixel halo is staged in LDS once and serves the gradient and the G-matrix taps of all 256
// pixels; only the bilinear samples of the current image (whose position depends on the evolving flow) go to the
// vector cache.  Batched over image pairs in blockIdx.z.
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

__device__ __forceinline__ float bilinear_g(const float* I, int w, int h, float xf, float yf)
{
    const float fx = floorf(xf), fy = floorf(yf);
    const float ax = xf - fx, ay = yf - fy;
    const int x0 = (int)fminf(fmaxf(fx, -4.0f), (float)w + 4.0f), y0 = (int)fminf(fmaxf(fy, -4.0f), (float)h + 4.0f);
    const float i00 = at_g(I, w, h, x0, y0), i01 = at_g(I, w, h, x0 + 1, y0);
    const float i10 = at_g(I, w, h, x0, y0 + 1), i11 = at_g(I, w, h, x0 + 1, y0 + 1);
    const float top = (1.0f - ax) * i00 + ax * i01, bot = (1.0f - ax) * i10 + ax * i11;
    return (1.0f - ay) * top + ay * bot;
}

constexpr int kOfTx = 32, kOfTy = 8;

// dynamic LDS: (kOfTx + 2R) x (kOfTy + 2R) floats, R = radius + 1
__global__ __launch_bounds__(kOfTx * kOfTy) void of_lk_kernel(OfArgs a, int l)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tile = reinterpret_cast<float*>(smem);
    const int pair = blockIdx.z;
    const int w = a.lv[l].w, h = a.lv[l].h, r = a.radius, R = r + 1;
    const float* I0 = a.pyr + ((size_t)pair * 2 + 0) * a.pyr_stride + a.lv[l].off;
    const float* I1 = a.pyr + ((size_t)pair * 2 + 1) * a.pyr_stride + a.lv[l].off;
    const int x0 = blockIdx.x * kOfTx, y0 = blockIdx.y * kOfTy;
    const int tw = kOfTx + 2 * R, th = kOfTy + 2 * R;
    const int tid = threadIdx.y * kOfTx + threadIdx.x;
    for (int i = tid; i < tw * th; i += kOfTx * kOfTy) {
        const int ty = i / tw, tx = i - ty * tw;
        tile[i] = at_g(I0, w, h, x0 - R + tx, y0 - R + ty);   // edge-extended previous image
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= w || y >= h) return;
    const int cx = threadIdx.x + R, cy = threadIdx.y + R;   // this pixel inside the tile

    float dx = 0.0f, dy = 0.0f;
    if (l < a.levels - 1) {
        const float* c = a.coarse + (size_t)pair * a.flow_stride + a.flow_off[l + 1] +
                         2 * ((size_t)(y >> 1) * a.lv[l + 1].w + (x >> 1));
        dx = 2.0f * c[0];
        dy = 2.0f * c[1];
    }
    float g11 = 0.0f, g12 = 0.0f, g22 = 0.0f;
    for (int oy = -r; oy <= r; ++oy)
        for (int ox = -r; ox <= r; ++ox) {
            const float* t = tile + (cy + oy) * tw + (cx + ox);
            const float ix = 0.5f * (t[1] - t[-1]);
            const float iy = 0.5f * (t[tw] - t[-tw]);
            g11 += ix * ix; g12 += ix * iy; g22 += iy * iy;
        }
    const float det = g11 * g22 - g12 * g12;
    if (det > a.det_min) {
        const float inv = 1.0f / det;
        for (int it = 0; it < a.iterations; ++it) {
            float b1 = 0.0f, b2 = 0.0f;
            for (int oy = -r; oy <= r; ++oy)
                for (int ox = -r; ox <= r; ++ox) {
                    const float* t = tile + (cy + oy) * tw + (cx + ox);
                    const float ix = 0.5f * (t[1] - t[-1]);
                    const float iy = 0.5f * (t[tw] - t[-tw]);
                    const float dt = bilinear_g(I1, w, h, (float)(x + ox) + dx, (float)(y + oy) + dy) - t[0];
                    b1 += ix * dt; b2 += iy * dt;
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
    const int R = a.radius + 1;
    const size_t lds = (size_t)(kOfTx + 2 * R) * (kOfTy + 2 * R) * sizeof(float);
    for (int l = a.levels - 1; l >= 0; --l)
        hipLaunchKernelGGL(of_lk_kernel, dim3((a.lv[l].w + kOfTx - 1) / kOfTx, (a.lv[l].h + kOfTy - 1) / kOfTy, a.n),
                           dim3(kOfTx, kOfTy), lds, s, a, l);
}

void launch_flow_quantise(const float* const* field, int16_t* const* out, int n, int W, int H, hipStream_t s)
{
    hipLaunchKernelGGL(of_quantise_kernel, dim3(((W / 4) * (H / 4) + 255) / 256, n), dim3(256), 0, s, field, out, W, H);
}

}  // namespace roft

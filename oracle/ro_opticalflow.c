/*
 * ro_opticalflow.c -- CPU oracle (test infrastructure) for the dense optical-flow PRODUCER.
 *
 * There is NO reference algorithm to restate here: the reference obtains flow from NVIDIA's fixed-function
 * optical-flow engine through OpenCV (src/roft-lib/src/ImageOpticalFlowNVOF.cpp:100-159, tools/nvof/dumper), which
 * has no AMD counterpart and no published arithmetic.  What the reference fixes is the CONTRACT of the product:
 * a forward flow field of frame k-1 pixels towards frame k, stored as CV_32FC2 at grid 1 or CV_16SC2 (S10.5, scale
 * 32) at grid 4 in the `.float` format (ImageOpticalFlowNVOF.cpp:19-80, OpticalFlowUtilities.cpp:77-136).
 * This file is the plain-C statement of the algorithm this project uses to fill that contract -- dense pyramidal
 * Lucas-Kanade -- and serves as the checker of the HIP kernels (k_opticalflow.hip).  "parity unpinned" by
 * construction; accuracy is tested against the analytic flow of the synthetic streams.
 *
 * Specification (all arithmetic in float):
 *   pyramid   level 0 = gray as float; level l+1 (y,x) = 0.25 * sum of the 2x2 block of level l
 *   Ic(z)     = I(clamp(z)) (edge-extended image)
 *   gradient  Ix(z) = 0.5 (I0c(z + (1,0)) - I0c(z - (1,0))), Iy likewise (of the edge-extended previous image)
 *   per pixel p of level l, window offsets o in [-r, r]^2 in row-major order:
 *     G = sum [Ix^2, Ix Iy; Ix Iy, Iy^2](p + o);  d = 2 * d_{l+1}(p >> 1) (0 at the coarsest level)
 *     if det G > det_min, `iterations` times:  b = sum grad(p+o) * (W(p + o) - I0c(p + o));  d -= G^-1 b
 *     W = the current image warped by d with ONE pair of bilinear weights for the whole window (d is constant over
 *     it): q = p + d, q0 = floor(q), a = q - q0;  h(z) = fma(ax, I1c(z + (1,0)), (1-ax) I1c(z))  (horizontal pass),
 *     W(p + o) = fma(ay, h(q0 + o + (0,1)), (1-ay) h(q0 + o))  (vertical pass)
 *     every accumulation of the window sums is ONE fused multiply-add per tap, s <- fma(a, b, s) (single rounding):
 *     the form the GPU's (packed) FMA units execute; everything else is separate IEEE operations (-ffp-contract=off)
 *   output    level-0 field; CV_16SC2: sampled at the centre (4i+2, 4j+2) of each 4x4 block, round(32 d) saturated
 */
#include "roft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float at(const float* I, int w, int h, int x, int y) { return I[(size_t)clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)]; }

int ro_optical_flow(const uint8_t* prev, const uint8_t* cur, int W, int H, int levels, int radius, int iterations,
                    float det_min, float* flow /* H x W x 2 */)
{
    if (levels < 1 || levels > 6 || (W % (1 << (levels - 1))) || (H % (1 << (levels - 1)))) return -1;
    float* P0[6];
    float* P1[6];
    float* D[6];
    int w[6], h[6];
    for (int l = 0; l < levels; l++) {
        w[l] = W >> l; h[l] = H >> l;
        P0[l] = (float*)malloc(sizeof(float) * w[l] * h[l]);
        P1[l] = (float*)malloc(sizeof(float) * w[l] * h[l]);
        D[l] = (l == 0) ? flow : (float*)malloc(sizeof(float) * 2 * w[l] * h[l]);
    }
    for (size_t i = 0; i < (size_t)W * H; i++) { P0[0][i] = (float)prev[i]; P1[0][i] = (float)cur[i]; }
    for (int l = 1; l < levels; l++)
        for (int y = 0; y < h[l]; y++)
            for (int x = 0; x < w[l]; x++) {
                const float* a = P0[l - 1]; const float* b = P1[l - 1];
                const size_t i0 = (size_t)(2 * y) * w[l - 1] + 2 * x, i1 = i0 + w[l - 1];
                P0[l][(size_t)y * w[l] + x] = 0.25f * (((a[i0] + a[i0 + 1]) + a[i1]) + a[i1 + 1]);
                P1[l][(size_t)y * w[l] + x] = 0.25f * (((b[i0] + b[i0 + 1]) + b[i1]) + b[i1 + 1]);
            }
    for (int l = levels - 1; l >= 0; l--) {
        const int wl = w[l], hl = h[l];
        const float* I0 = P0[l]; const float* I1 = P1[l];
        for (int y = 0; y < hl; y++)
            for (int x = 0; x < wl; x++) {
                float dx = 0.0f, dy = 0.0f;
                if (l < levels - 1) {
                    const float* c = D[l + 1] + 2 * ((size_t)(y >> 1) * w[l + 1] + (x >> 1));
                    dx = 2.0f * c[0]; dy = 2.0f * c[1];
                }
                float g11 = 0.0f, g12 = 0.0f, g22 = 0.0f;
                for (int oy = -radius; oy <= radius; oy++)
                    for (int ox = -radius; ox <= radius; ox++) {
                        const float ix = 0.5f * (at(I0, wl, hl, x + ox + 1, y + oy) - at(I0, wl, hl, x + ox - 1, y + oy));
                        const float iy = 0.5f * (at(I0, wl, hl, x + ox, y + oy + 1) - at(I0, wl, hl, x + ox, y + oy - 1));
                        g11 = fmaf(ix, ix, g11); g12 = fmaf(ix, iy, g12); g22 = fmaf(iy, iy, g22);
                    }
                const float det = g11 * g22 - g12 * g12;
                if (det > det_min) {
                    const float inv = 1.0f / det;
                    for (int it = 0; it < iterations; it++) {
                        const float xf = (float)x + dx, yf = (float)y + dy;
                        const float fx = floorf(xf), fy = floorf(yf);
                        const float ax = xf - fx, ay = yf - fy;
                        /* keep the integer conversion in range for wild displacements */
                        const int x0 = (int)fminf(fmaxf(fx, -64.0f), (float)wl + 64.0f);
                        const int y0 = (int)fminf(fmaxf(fy, -64.0f), (float)hl + 64.0f);
                        float b1 = 0.0f, b2 = 0.0f;
                        for (int oy = -radius; oy <= radius; oy++)
                            for (int ox = -radius; ox <= radius; ox++) {
                                const float ix = 0.5f * (at(I0, wl, hl, x + ox + 1, y + oy) - at(I0, wl, hl, x + ox - 1, y + oy));
                                const float iy = 0.5f * (at(I0, wl, hl, x + ox, y + oy + 1) - at(I0, wl, hl, x + ox, y + oy - 1));
                                const float top = fmaf(ax, at(I1, wl, hl, x0 + ox + 1, y0 + oy), (1.0f - ax) * at(I1, wl, hl, x0 + ox, y0 + oy));
                                const float bot = fmaf(ax, at(I1, wl, hl, x0 + ox + 1, y0 + oy + 1), (1.0f - ax) * at(I1, wl, hl, x0 + ox, y0 + oy + 1));
                                const float it_ = fmaf(ay, bot, (1.0f - ay) * top) - at(I0, wl, hl, x + ox, y + oy);
                                b1 = fmaf(ix, it_, b1); b2 = fmaf(iy, it_, b2);
                            }
                        dx -= (g22 * b1 - g12 * b2) * inv;
                        dy -= (g11 * b2 - g12 * b1) * inv;
                    }
                }
                D[l][2 * ((size_t)y * wl + x)] = dx;
                D[l][2 * ((size_t)y * wl + x) + 1] = dy;
            }
    }
    for (int l = 0; l < levels; l++) {
        free(P0[l]); free(P1[l]);
        if (l) free(D[l]);
    }
    return 0;
}

/* CV_16SC2 grid-4 product of a dense field (S10.5 fixed point, ImageOpticalFlowNVOF.cpp:19-80) */
void ro_flow_to_s16_grid4(const float* flow, int W, int H, int16_t* out /* H/4 x W/4 x 2 */)
{
    for (int j = 0; j < H / 4; j++)
        for (int i = 0; i < W / 4; i++)
            for (int c = 0; c < 2; c++) {
                float v = rintf(flow[2 * ((size_t)(4 * j + 2) * W + (4 * i + 2)) + c] * 32.0f);
                v = fminf(fmaxf(v, -32768.0f), 32767.0f);
                out[2 * ((size_t)j * (W / 4) + i) + c] = (int16_t)v;
            }
}

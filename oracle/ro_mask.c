/*
 * ro_mask.c -- CPU oracle (test infrastructure) for optical-flow-aided mask propagation.
 *
 * Follows ImageSegmentationOFAidedSource<T>::map  include/ROFT/ImageSegmentationOFAidedSource.hpp:234-281
 * and the cv::remap(mask_, mask_, map, INTER_LINEAR, BORDER_CONSTANT) calls at :215 and :225.
 * The map holds integer-valued source coordinates, so bilinear interpolation degenerates to a
 * plain gather out(y,x) = mask(map(y,x)); untouched map entries are (0,0) and therefore sample
 * mask(0,0).  "Later writers overwrite" in row-major source order (:277).
 */
#include "roft_oracle.h"

#include <string.h>
#include <stdlib.h>

static inline void flow_at(const ro_flow* f, int row, int col, float* dx, float* dy)
{
    size_t idx = ((size_t)row * (size_t)f->cols + (size_t)col) * 2;
    if (f->type == RO_FLOW_S16C2) {
        const int16_t* p = (const int16_t*)f->data;
        *dx = (float)p[idx] / f->scale;
        *dy = (float)p[idx + 1] / f->scale;
    } else {
        const float* p = (const float*)f->data;
        *dx = (float)p[idx] / f->scale;
        *dy = (float)p[idx + 1] / f->scale;
    }
}

/* (int)float as the reference's x86-64 build evaluates it (cvttss2si): truncation toward zero,
 * and the "integer indefinite" value INT_MIN for NaN / out-of-range inputs (flow entries may be
 * NaN or 1e10, OpticalFlowUtilities.h:19-22) -- which then fails the `< 0` bounds test. */
static inline int trunc_int(float x)
{
    if (!(x > -2147483904.0f && x < 2147483648.0f)) return INT32_MIN;
    return (int)x;
}

void ro_mask_propagate(uint8_t* mask, int W, int H, const ro_flow* flows, int n_flows,
                       int frames_between, int32_t* scratch)
{
    int32_t* map = scratch; /* linear source index, 0 == (0,0) == untouched */
    memset(map, 0, sizeof(int32_t) * (size_t)W * H);

    int start = 0;
    if (frames_between > 0) {
        start = n_flows - frames_between;
        if (start < 0) start = 0;
    }

    for (int py = 0; py < H; py++) {
        for (int px = 0; px < W; px++) {
            if (mask[(size_t)py * W + px] == 0) continue;
            float t_x = (float)px;
            float t_y = (float)py;
            int error = 0;
            for (int j = start; j < n_flows; j++) {
                if ((trunc_int(t_x) < 0) || (trunc_int(t_x) >= W) || (trunc_int(t_y) < 0) || (trunc_int(t_y) >= H)) {
                    error = 1;
                    break;
                }
                /* flow_grid_size_ is a size_t: float / size_t -> float division (:268) */
                float dx, dy;
                flow_at(&flows[j], trunc_int(t_y / (float)flows[j].grid), trunc_int(t_x / (float)flows[j].grid),
                        &dx, &dy);
                t_x += dx;
                t_y += dy;
            }
            if (error || (trunc_int(t_x) < 0) || (trunc_int(t_x) >= W) || (trunc_int(t_y) < 0) || (trunc_int(t_y) >= H))
                continue;
            map[(size_t)trunc_int(t_y) * W + (size_t)trunc_int(t_x)] = py * W + px;
        }
    }

    uint8_t* src = (uint8_t*)malloc((size_t)W * H);
    memcpy(src, mask, (size_t)W * H);
    for (size_t i = 0; i < (size_t)W * H; i++) mask[i] = src[map[i]];
    free(src);
}

void ro_mask_binarise(const uint8_t* src, uint8_t* dst, size_t n)
{
    for (size_t i = 0; i < n; i++) dst[i] = (src[i] > 1) ? 255 : 0;
}

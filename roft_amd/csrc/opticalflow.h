// opticalflow.h -- argument block of the optical-flow producer kernels (k_opticalflow.hip)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace roft {

struct OfLevel {
    int w, h;
    size_t off;   // offset (floats) of this level inside one image's pyramid
};

struct OfArgs {
    int n;                 // image pairs
    int levels, radius, iterations;
    float det_min;
    OfLevel lv[6];
    size_t pyr_stride;     // floats per image pyramid
    size_t flow_off[6];    // offset (floats) of level l's flow field inside one pair's coarse-flow workspace (l >= 1)
    size_t flow_stride;    // floats of coarse-flow workspace per pair
    const uint8_t* const* prev;   // [n] device pointers (array itself in device memory)
    const uint8_t* const* cur;
    float* pyr;            // [n][2][pyr_stride]
    float* coarse;         // [n][flow_stride]
    float* const* out_f32; // [n] level-0 field (H x W x 2), always written
};

void launch_optical_flow(const OfArgs& a, hipStream_t s);
void launch_flow_quantise(const float* const* field, int16_t* const* out, int n, int W, int H, hipStream_t s);

}  // namespace roft

// plane_rank.h -- row-major rank bookkeeping on a mask bit plane staged in LDS (gfx950).
//
// cv::findNonZero order is row-major; the reference walks that list with a stride (every 35th pixel
// for the flow measurement, hpp:237; every 2nd for the depth likelihood, ROFTFilter.cpp:556).
// Plane words are in row-major order too, so a block scan over the popcounts of contiguous word chunks
// gives every word its starting rank and the pixel of rank r is found without materialising the list.
#pragma once

#include "roft_device.h"

namespace roft {

// inclusive prefix sum over the 64 lanes of a wave on the DPP network (no LDS traffic): Kogge-Stone inside the four
// 16-lane rows (row_shr 1, 2, 4, 8; lanes without a source add 0), then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// block-wide exclusive scan of one int per thread (blockDim.x multiple of 64, <= 1024): wave scans on the DPP network,
// the wave totals are scanned redundantly by every wave (no serial section), two barriers
__device__ inline int block_exclusive_scan(int v, int* s_wave /*>= 16 ints*/, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int inc = wave_inclusive_scan(v);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    const int wv = lane < nw ? s_wave[lane] : 0;
    const int winc = wave_inclusive_scan(wv);
    *total = __builtin_amdgcn_readlane(winc, 63);
    const int res = __builtin_amdgcn_readlane(winc - wv, __builtin_amdgcn_readfirstlane(wave)) + inc - v;
    __syncthreads();   // s_wave may be rewritten by the next scan
    return res;
}

}  // namespace roft

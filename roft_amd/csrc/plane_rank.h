// plane_rank.h -- row-major rank queries on a mask bit plane staged in LDS (gfx950).
//
// cv::findNonZero order is row-major; the reference walks that list with a stride (every 35th pixel
// for the flow measurement, hpp:237; every 2nd for the depth likelihood, ROFTFilter.cpp:556).
// With the plane (W*H/8 bytes) in LDS, the pixel of rank r is found without materialising the list:
// binary search over the exclusive row prefix, then a popcount walk over the row's words.
#pragma once

#include "roft_device.h"

namespace roft {

// block-wide exclusive scan of one int per thread (blockDim.x multiple of 64, <= 1024)
__device__ inline int block_exclusive_scan(int v, int* s_wave /*>= 17 ints*/, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < nw; ++w) { int t = s_wave[w]; s_wave[w] = run; run += t; }
        s_wave[16] = run;
    }
    __syncthreads();
    const int res = s_wave[wave] + inc - v;
    *total = s_wave[16];
    __syncthreads();
    return res;
}

__host__ __device__ inline size_t plane_lds_bytes(size_t plane_words, int H)
{
    return ((plane_words * 4 + 15) & ~(size_t)15) + (size_t)(H + 1) * 4;
}

// Stage `plane` (global) into s_plane and build s_rowpref[0..H]; returns the number of set bits.
__device__ inline int stage_plane(const uint32_t* plane, size_t plane_words, int H, int wpr, uint32_t* s_plane,
                                  int* s_rowpref, int* s_wave)
{
    const size_t n4 = plane_words / 4;
    for (size_t i = threadIdx.x; i < n4; i += blockDim.x)
        reinterpret_cast<uint4*>(s_plane)[i] = reinterpret_cast<const uint4*>(plane)[i];
    for (size_t i = n4 * 4 + threadIdx.x; i < plane_words; i += blockDim.x) s_plane[i] = plane[i];
    __syncthreads();
    int carry = 0;
    for (int r0 = 0; r0 < H; r0 += blockDim.x) {
        const int r = r0 + threadIdx.x;
        int cnt = 0;
        if (r < H)
            for (int w = 0; w < wpr; ++w) cnt += __popc(s_plane[(size_t)r * wpr + w]);
        int total;
        const int ex = block_exclusive_scan(cnt, s_wave, &total);
        if (r < H) s_rowpref[r] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) s_rowpref[H] = carry;
    __syncthreads();
    return s_rowpref[H];
}

// pixel (u, v) of row-major rank `rank` (0 <= rank < set bits)
__device__ inline void select_rank(const uint32_t* s_plane, const int* s_rowpref, int H, int wpr, int rank, int& u,
                                   int& v)
{
    int lo = 0, hi = H;  // largest row with rowpref[row] <= rank
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_rowpref[mid] <= rank) lo = mid; else hi = mid;
    }
    int k = rank - s_rowpref[lo];
    int w = 0;
    uint32_t bits = s_plane[(size_t)lo * wpr];
    int pc = __popc(bits);
    while (k >= pc) { k -= pc; ++w; bits = s_plane[(size_t)lo * wpr + w]; pc = __popc(bits); }
    for (int i = 0; i < k; ++i) bits &= bits - 1;
    u = w * 32 + __builtin_ctz(bits);
    v = lo;
}

}  // namespace roft

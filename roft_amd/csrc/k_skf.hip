// k_skf.hip -- velocity Kalman filter: prediction + sequential correction in information form (gfx950).
//
// Reference:
//   bfl::KFPrediction over SpatialVelocityModel (F = I, P += Q)   src/roft-lib/src/SpatialVelocityModel.cpp:15-27
//   SKFCorrection::correctStep                                    src/roft-lib/src/SKFCorrection.cpp:37-153
//   observability rule (N < 3 -> keep the belief from before the prediction)  src/roft-lib/src/ROFTFilter.cpp:294-301
//
// The reference applies the N two-row measurements one after the other (cpp:129-149).  All of
// them are linearised at fixed H_j and their weights l_j are fixed before the loop, so the
// recursion is algebraically the batch update
//     Lambda = (P^-)^-1 + sum_j l_j H_j' R^-1 H_j,   P^+ = Lambda^-1,
//     s^+    = s^- + P^+ sum_j l_j H_j' R^-1 (y_j - H_j s^-)
// which is a pure reduction over the measurements: one workgroup per object, fp64 accumulators,
// wave shuffle + LDS reduction, 6x6 Cholesky solves on one lane.  Differs from the sequential
// form only by rounding (tolerance stated in tests/test_parity_gpu.py).
//
// Laplacian re-weighting, including the reference's pairing quirk: the 2N innovation vector is
// viewed as an N x 2 COLUMN-major matrix (cpp:93), so the median / scale are fitted to
// sqrt(e[k]^2 + e[N+k]^2) while the likelihoods use the true per-point norm (cpp:111).
// The median is an exact order statistic found by an 8-pass radix select over the IEEE bit
// patterns (non-negative doubles order like unsigned integers) -- no sort.
#include "plane_rank.h"

namespace roft {

__device__ __forceinline__ void h_rows(const FlowRec& r, const DevCamera& cam, double dt, double h[12])
{
    const double z = (double)r.z;
    const double uu = (r.u - cam.cx);
    const double vv = (r.v - cam.cy);
    h[0] = (cam.fx / z) * dt;
    h[1] = 0.0;
    h[2] = (-uu / z) * dt;
    h[3] = (-uu * vv / cam.fy) * dt;
    h[4] = (cam.fx + uu * uu / cam.fx) * dt;
    h[5] = (-vv * cam.fx / cam.fy) * dt;
    h[6] = 0.0;
    h[7] = (cam.fy / z) * dt;
    h[8] = (-vv / z) * dt;
    h[9] = (-(cam.fy + vv * vv / cam.fy)) * dt;
    h[10] = (vv * uu / cam.fx) * dt;
    h[11] = (uu * cam.fy / cam.fx) * dt;
}

// Measurement accessors: the engine feeds compact flow records (H rows rebuilt on the fly, no
// 96-byte-per-point matrix in HBM); the operator-level entry point feeds explicit (y, H) arrays.
struct RecAccessor {
    const FlowRec* recs;
    DevCamera cam;
    double dt;
    // H rows with reciprocal multiplies instead of the ten IEEE divisions of the literal expression
    // (hpp:279-280): agrees with h_rows() to an ulp or two; the bit-exact assembly is expand_yh_kernel.
    double ifx, ify;   // 1 / fx, 1 / fy: two of the three divisions per point are the same for every point
    __device__ RecAccessor(const FlowRec* recs_, DevCamera cam_, double dt_) : recs(recs_), cam(cam_), dt(dt_), ifx(1.0 / cam_.fx), ify(1.0 / cam_.fy) {}
    __device__ __forceinline__ void get(int j, double h[12], double y[2]) const
    {
        const FlowRec r = recs[j];
        const double iz = 1.0 / (double)r.z;
        const double uu = (r.u - cam.cx), vv = (r.v - cam.cy);
        h[0] = (cam.fx * iz) * dt;
        h[1] = 0.0;
        h[2] = (-uu * iz) * dt;
        h[3] = (-uu * vv * ify) * dt;
        h[4] = (cam.fx + uu * uu * ifx) * dt;
        h[5] = (-vv * cam.fx * ify) * dt;
        h[6] = 0.0;
        h[7] = (cam.fy * iz) * dt;
        h[8] = (-vv * iz) * dt;
        h[9] = (-(cam.fy + vv * vv * ify)) * dt;
        h[10] = (vv * uu * ifx) * dt;
        h[11] = (uu * cam.fy * ifx) * dt;
        y[0] = (double)r.dx;
        y[1] = (double)r.dy;
    }
};

struct ArrayAccessor {
    const double* y;
    const double* H;
    __device__ __forceinline__ void get(int j, double h[12], double yy[2]) const
    {
        for (int i = 0; i < 12; ++i) h[i] = H[(size_t)12 * j + i];
        yy[0] = y[2 * j];
        yy[1] = y[2 * j + 1];
    }
};

// innovation component k (flat index into the 2N vector) at the predicted mean
template <class Acc>
__device__ __forceinline__ double innov_at(const Acc& acc, int k, const double* x)
{
    double h[12], y[2];
    acc.get(k >> 1, h, y);
    const double* hr = h + 6 * (k & 1);
    double pred = 0.0;
    for (int i = 0; i < 6; ++i) pred += hr[i] * x[i];
    return -(pred - y[k & 1]);
}

// One workgroup of eight waves per object.  A frame has 700 - 3 000 flow points (640x480 - 1280x720): more waves
// shorten the passes over the points (a handful of dependent operations per point and thread) but lengthen every
// reduction and barrier between them.  Measured per frame, chains overlapping: 256 threads 23 / 68 us
// (640x480 / 1280x720), 512 threads 22 / 49 us, 1024 threads 32 / 52 us.
#ifndef ROFT_SKF_THREADS
#define ROFT_SKF_THREADS 512
#endif
constexpr int kSkfThreads = ROFT_SKF_THREADS;

// phase stamps (build with -DROFT_SKF_PROFILE): SKFTICK(i) stores the 100 MHz wall clock ticks since the previous stamp
#ifdef ROFT_SKF_PROFILE
#define SKFTICK(i) do { __syncthreads(); if (threadIdx.x == 0) { long long _t = wall_clock64(); S.dbg[i] = _t - S.t0; S.t0 = _t; } } while (0)
#else
#define SKFTICK(i) do {} while (0)
#endif

// Wave reductions of doubles on the DPP network (no LDS traffic): both halves of the double travel as 32-bit DPP
// moves -- row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 / row_bcast:31 across rows -- and lane 63
// ends up with the reduction over all 64 lanes.  Lanes without a source receive `fill`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double v, double fill)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_to_lane63(double v)
{
    v += dpp_move_f64<0x111, 0xf>(v, 0.0);
    v += dpp_move_f64<0x112, 0xf>(v, 0.0);
    v += dpp_move_f64<0x114, 0xf>(v, 0.0);
    v += dpp_move_f64<0x118, 0xf>(v, 0.0);
    v += dpp_move_f64<0x142, 0xa>(v, 0.0);
    v += dpp_move_f64<0x143, 0xc>(v, 0.0);
    return v;
}

__device__ __forceinline__ double wave_max_to_lane63(double v)
{
    const double ninf = -INFINITY;
    v = fmax(v, dpp_move_f64<0x111, 0xf>(v, ninf));
    v = fmax(v, dpp_move_f64<0x112, 0xf>(v, ninf));
    v = fmax(v, dpp_move_f64<0x114, 0xf>(v, ninf));
    v = fmax(v, dpp_move_f64<0x118, 0xf>(v, ninf));
    v = fmax(v, dpp_move_f64<0x142, 0xa>(v, ninf));
    v = fmax(v, dpp_move_f64<0x143, 0xc>(v, ninf));
    return v;
}

// Wave reduction of up to 32 accumulators at once ("transpose-reduce"): in the step with lane distance `off` every lane
// keeps one half of its values -- the lower half if its bit `off` is clear, the upper half otherwise -- and adds to each
// the partner's copy of it, so the number of values per lane halves while the number of lanes summed doubles: 16 + 8 +
// 4 + 2 + 1 + 1 = 32 exchanges instead of 6 per accumulator.  On return v[0] of lane l is the sum over all 64 lanes of the
// accumulator with index  bit5(l) 16 + bit4(l) 8 + bit3(l) 4 + bit2(l) 2 + bit1(l)  (both lanes of a pair hold it).
template <int H>
__device__ __forceinline__ void transpose_reduce_step(double* v, int off)
{
    const bool up = (threadIdx.x & off) != 0;
#pragma unroll
    for (int k = 0; k < H; ++k) {
        const double keep = up ? v[k + H] : v[k];
        const double send = up ? v[k] : v[k + H];
        v[k] = keep + __shfl_xor(send, off, 64);
    }
}

__device__ __forceinline__ void wave_transpose_reduce32(double v[32])
{
    transpose_reduce_step<16>(v, 32);
    transpose_reduce_step<8>(v, 16);
    transpose_reduce_step<4>(v, 8);
    transpose_reduce_step<2>(v, 4);
    transpose_reduce_step<1>(v, 2);
    v[0] += __shfl_xor(v[0], 1, 64);
}

__device__ double block_sum(double v, double* s_red)
{
    v = wave_sum_to_lane63(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
    return t;
}

__device__ double block_max(double v, double* s_red)
{
    v = wave_max_to_lane63(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = s_red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmax(t, s_red[w]);
    return t;
}

// exact order statistic `rank` (0-based) of N non-negative doubles
__device__ unsigned long long radix_select(const double* vals, int N, int rank, int* s_hist, int* s_sel)
{
    unsigned long long prefix = 0, mask = 0;
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const unsigned long long k = (unsigned long long)__double_as_longlong(vals[i]);
            if ((k & mask) == prefix) atomicAdd(&s_hist[(int)((k >> shift) & 255ull)], 1);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            const int h0 = s_hist[4 * l], h1 = s_hist[4 * l + 1], h2 = s_hist[4 * l + 2], h3 = s_hist[4 * l + 3];
            const int loc = h0 + h1 + h2 + h3;
            int inc = loc;
            for (int off = 1; off < 64; off <<= 1) {
                int t = __shfl_up(inc, off, 64);
                if (l >= off) inc += t;
            }
            const int exc = inc - loc;
            // the lane whose [exc, inc) interval contains rank
            if (rank >= exc && rank < inc) {
                int r = rank - exc, b = 4 * l;
                if (r >= h0) { r -= h0; ++b; if (r >= h1) { r -= h1; ++b; if (r >= h2) { r -= h2; ++b; } } }
                s_sel[0] = b;
                s_sel[1] = r;
            }
        }
        __syncthreads();
        prefix |= ((unsigned long long)s_sel[0]) << shift;
        mask |= 255ull << shift;
        rank = s_sel[1];
        __syncthreads();
    }
    return prefix;
}

// In-place inverse of a 6x6 SPD matrix held in LDS (row-major M[36]) by ONE wave, lane (i, j) = 6 i + j owning one
// entry: six sweeps (Gauss-Jordan without pivoting -- the pivots of an SPD matrix are its positive Schur
// complements), then the lower triangle is overwritten by the upper one so that the result is exactly symmetric.
// Every lane of the wave must call; returns (wave-uniform) false if a pivot is not positive.
// 1 / d from the hardware seed (v_rcp_f64, ~24 bits) and two Newton steps: full double accuracy in a third of the dependent
// instructions of an IEEE division -- these reciprocals sit on the one-wave chain of the frame (pivots, scales)
__device__ __forceinline__ double skf_rcp(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
}

__device__ bool spd_inverse6_wave(double* M)
{
    const int l = threadIdx.x & 63;
    const bool on = l < 36;
    const int i = on ? l / 6 : 0, j = on ? l % 6 : 0;
    double a = M[on ? l : 0];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double d = M[k * 7], rk = M[k * 6 + j], ck = M[i * 6 + k];
        if (!(d > 0.0)) ok = false;
        const double inv = skf_rcp(d);       // one reciprocal per sweep; the four cases are selects, not branches
        const double scaled = a * inv;
        const double swept = a - ck * (rk * inv);
        a = (i == k) ? ((j == k) ? inv : scaled) : ((j == k) ? -scaled : swept);
        __builtin_amdgcn_wave_barrier();   // LDS operations of a wave execute in order: all reads above precede the writes
        if (on) M[l] = a;
        __builtin_amdgcn_wave_barrier();
    }
    const double up = M[i <= j ? i * 6 + j : j * 6 + i];
    __builtin_amdgcn_wave_barrier();
    if (on) M[l] = up;
    __builtin_amdgcn_wave_barrier();
    return ok;
}

constexpr int kSkfLdsN = 4096;  // measurement counts up to this keep innovations + norms in LDS (96 KB)
constexpr int kBins = 1024;
constexpr int kBucketCap = 256;

struct SkfShared {
    double red[kSkfThreads / 64];
    double acc[27][kSkfThreads / 64];
    alignas(16) int hist[kBins];
    int sel[2];
    int cnt[2];
    double list[2][kBucketCap];
    double med[2];
    double Lm[36], Ppi[36], eta[6], xo[6];   // information matrix -> posterior covariance, prior information, ...
    double ein[2 * kSkfLdsN];
    double qn[kSkfLdsN];
#ifdef ROFT_SKF_PROFILE
    long long t0, dbg[16];
#endif
};

// Exact order statistics of ranks ra <= rb (rb - ra <= 1) of N non-negative doubles in three barrier phases:
// a monotone 1024-bin histogram over [0, 8 x mean] locates the bucket of each rank, the (few) members of that
// bucket are collected and ranked by counting.  Returns false if a bucket holds more than kBucketCap values.
//
// The bins cover [0, 8 x mean], what lies above goes into the last one (the mapping stays monotone, so the prefix counts
// stay exact): the median of non-negative values is at most twice their mean, and a few gross outliers -- flow vectors
// that lost their pixel -- do not stretch the bins until the median's bucket holds hundreds of values, as bins over
// [min, max] did (1280x720, N ~ 3 000: the select took 11.6 of the frame's 28 us and some objects fell back to the
// 8-pass radix select).  No pass for the extremes either: the sum comes with the values.
//
// bucket_select_prepare: called by every thread with its share of the sum of the values, BEFORE the barrier that
// publishes the values.
__device__ void bucket_select_prepare(SkfShared& S, double my_sum)
{
    const double ws = wave_sum_to_lane63(my_sum);
    if ((threadIdx.x & 63) == 63) S.red[threadIdx.x >> 6] = ws;
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) S.hist[i] = 0;
    if (threadIdx.x < 2) S.cnt[threadIdx.x] = 0;
}

__device__ bool bucket_select2(const double* vals, int N, int ra, int rb, SkfShared& S, double& va, double& vb)
{
    double sum = S.red[0];
    for (int w = 1; w < kSkfThreads / 64; ++w) sum += S.red[w];
    const double top = 8.0 * (sum / (double)N);
    if (!(top > 0.0)) { va = vb = 0.0; __syncthreads(); return true; }   // all values are zero (barrier: S.red is the caller's next)
    const double scale = (double)(kBins - 1) / top;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        int b = (int)(vals[k] * scale);   // monotone in vals[k]
        b = b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
        atomicAdd(&S.hist[b], 1);
    }
    __syncthreads();
    // Every wave scans the whole histogram by itself (16 consecutive bins per lane, a shuffle scan over the lane sums)
    // and keeps the two buckets in registers: no barrier between the histogram and the collection.
    int bin[2], base[2];
    {
        constexpr int kPerLane = kBins / 64;
        static_assert(kPerLane == 16, "four int4 reads per lane");
        const int l = threadIdx.x & 63;
        int h[kPerLane], tot = 0;
#pragma unroll
        for (int i = 0; i < kPerLane / 4; ++i) {
            const int4 q = *reinterpret_cast<const int4*>(&S.hist[kPerLane * l + 4 * i]);
            h[4 * i] = q.x; h[4 * i + 1] = q.y; h[4 * i + 2] = q.z; h[4 * i + 3] = q.w;
            tot += q.x + q.y + q.z + q.w;
        }
        int incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (l >= off) incl += o; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = q ? rb : ra;
            int pre = incl - tot, mybin = 0, mybase = 0;
            const bool mine = r >= pre && r < incl;
#pragma unroll
            for (int i = 0; i < kPerLane; ++i) {
                if (r >= pre && r < pre + h[i]) { mybin = kPerLane * l + i; mybase = pre; }
                pre += h[i];
            }
            const int src = __ffsll((long long)__ballot(mine)) - 1;   // exactly one lane: 0 <= r < N = sum of the bins
            bin[q] = __shfl(mybin, src, 64);
            base[q] = __shfl(mybase, src, 64);
        }
    }
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const double v = vals[k];
        int b = (int)(v * scale);
        b = b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
        if (b == bin[0]) { const int p = atomicAdd(&S.cnt[0], 1); if (p < kBucketCap) S.list[0][p] = v; }
        if (b == bin[1]) { const int p = atomicAdd(&S.cnt[1], 1); if (p < kBucketCap) S.list[1][p] = v; }
    }
    __syncthreads();
    if (S.cnt[0] > kBucketCap || S.cnt[1] > kBucketCap) return false;
    for (int q = 0; q < 2; ++q) {
        const int m = S.cnt[q], r = (q ? rb : ra) - base[q];
        if ((int)threadIdx.x < m) {
            const double v = S.list[q][threadIdx.x];
            int less = 0, eq = 0;
            for (int j = 0; j < m; ++j) { const double u = S.list[q][j]; less += (u < v) ? 1 : 0; eq += (u == v) ? 1 : 0; }
            if (r >= less && r < less + eq) S.med[q] = v;   // all writers hold the same value
        }
    }
    __syncthreads();
    va = S.med[0];   // next written after the barriers of the next call
    vb = S.med[1];
    return true;
}

// Correction of (x, P_pred) with N measurements; the result is left in S.xo / S.Lm (valid if 0 is returned).
// Returns (to every thread) 0 = corrected, 3 = numerically singular.
// scratch: 3 * N doubles of global memory, used only when N > kSkfLdsN.
template <class Acc>
__device__ int skf_core(const Acc& acc_in, int N, const double x[6], const double* P_pred, const double r_flow[2],
                        int reweight, double* scratch, SkfShared& S)
{
    double mi = 0.0, b = 0.0, lmax = 1.0;
    bool weighted = false;
    const bool in_lds = N <= kSkfLdsN;
    double* ein = in_lds ? S.ein : scratch;            // 2N innovations at the predicted mean
    double* qn = in_lds ? S.qn : scratch + 2 * (size_t)N;
    if (reweight) {
        for (int j = threadIdx.x; j < N; j += blockDim.x) {
            double h[12], y[2];
            acc_in.get(j, h, y);
            double p0 = 0.0, p1 = 0.0;
            for (int i = 0; i < 6; ++i) { p0 += h[i] * x[i]; p1 += h[6 + i] * x[i]; }
            ein[2 * j] = -(p0 - y[0]);
            ein[2 * j + 1] = -(p1 - y[1]);
        }
        __syncthreads();  // (scratch is written and read by this workgroup only)
        SKFTICK(1);
        double qsum = 0.0;
        for (int k = threadIdx.x; k < N; k += blockDim.x) {
            const double e0 = ein[k], e1 = ein[N + k];   // column-major pairing of the reference (cpp:93)
            const double nrm = sqrt(e0 * e0 + e1 * e1);
            qn[k] = nrm;
            qsum += nrm;
        }
        bucket_select_prepare(S, qsum);
        __syncthreads();
        SKFTICK(2);
        const int ra = (N % 2 == 0) ? N / 2 - 1 : N / 2, rb = N / 2;
        double va, vb;
        if (!bucket_select2(qn, N, ra, rb, S, va, vb)) {
            // a bucket overflowed (heavily clustered values): exact 8-pass radix select instead
            va = __longlong_as_double((long long)radix_select(qn, N, ra, S.hist, S.sel));
            vb = va;
            if (rb != ra) {
                // element of rank N/2: va again if duplicates reach that rank, else the smallest value > va
                double cnt_le = 0.0, min_gt = INFINITY;
                for (int k = threadIdx.x; k < N; k += blockDim.x) {
                    const double vq = qn[k];
                    if (vq <= va) cnt_le += 1.0; else min_gt = fmin(min_gt, vq);
                }
                cnt_le = block_sum(cnt_le, S.red);
                min_gt = -block_max(-min_gt, S.red);
                vb = (cnt_le > (double)(N / 2)) ? va : min_gt;
            }
        }
        SKFTICK(3);
        mi = (N % 2 == 0) ? 0.5 * (va + vb) : vb;
        double sabs = 0.0;
        for (int k = threadIdx.x; k < N; k += blockDim.x) sabs += fabs(qn[k] - mi);
        b = block_sum(sabs, S.red) / N;
        SKFTICK(4);
        if (b > 1e-4) {
            weighted = true;
            double m = 0.0;
            __syncthreads();   // every thread has read the norms (the sum above): they make room for the likelihoods
            const double ib = skf_rcp(b), half_ib = 0.5 * ib;
            for (int j = threadIdx.x; j < N; j += blockDim.x) {
                const double e0 = ein[2 * j], e1 = ein[2 * j + 1];
                const double nj = sqrt(e0 * e0 + e1 * e1);
                double l = half_ib * exp(-fabs(nj - mi) * ib);
                if (l < 1e-6) l = 1e-6;
                qn[j] = l;     // (read back by the same thread in the accumulation below: same j -> same thread)
                m = fmax(m, l);
            }
            lmax = block_max(m, S.red);
        }
    }

    SKFTICK(5);
    // information accumulation: 21 unique entries of sum l H'R^-1 H and 6 of sum l H'R^-1 e
    double acc[32];   // 27 used; padded for the transpose-reduce below
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    const double ir0 = 1.0 / r_flow[0], ir1 = 1.0 / r_flow[1];
    const double ilmax = weighted ? skf_rcp(lmax) : 1.0;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        double h[12], y[2];
        acc_in.get(j, h, y);
        double e0, e1;
        if (reweight) {   // innovations and likelihoods of the passes above
            e0 = ein[2 * j];
            e1 = ein[2 * j + 1];
        } else {
            double p0 = 0.0, p1 = 0.0;
            for (int i = 0; i < 6; ++i) { p0 += h[i] * x[i]; p1 += h[6 + i] * x[i]; }
            e0 = -(p0 - y[0]);
            e1 = -(p1 - y[1]);
        }
        double l = 1.0;
        if (weighted) l = qn[j] * ilmax;
        const double w0 = l * ir0, w1 = l * ir1;
        int t = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int k = i; k < 6; ++k) acc[t++] += w0 * h[i] * h[k] + w1 * h[6 + i] * h[6 + k];
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[21 + i] += w0 * h[i] * e0 + w1 * h[6 + i] * e1;
    }
    SKFTICK(6);
    __syncthreads();
    wave_transpose_reduce32(acc);
    {
        const int l = threadIdx.x & 63, idx = ((l >> 5) & 1) * 16 + ((l >> 4) & 1) * 8 + ((l >> 3) & 1) * 4 + ((l >> 2) & 1) * 2 + ((l >> 1) & 1);
        if (!(l & 1) && idx < 27) S.acc[idx][threadIdx.x >> 6] = acc[0];
    }
    __syncthreads();
    SKFTICK(7);

    // posterior: P = (P_pred^-1 + sum l H'R^-1 H)^-1, x = x_pred + P eta.  Wave 0 assembles the sums while wave 1
    // inverts the prior covariance; wave 0 then inverts the information matrix.
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (wave == 0 && lane < 27) {
            double v = 0.0;
            for (int w = 0; w < kSkfThreads / 64; ++w) v += S.acc[lane][w];
            if (lane < 21) {
                int i = 0, k = lane;          // lane -> (i, k), k >= i, of the upper triangle in row-major order
                while (k >= 6 - i) { k -= 6 - i; ++i; }
                k += i;
                S.Lm[i * 6 + k] = v;
                S.Lm[k * 6 + i] = v;
            } else {
                S.eta[lane - 21] = v;
            }
        }
        if (wave == 1) {
            if (lane < 36) S.Ppi[lane] = P_pred[lane];
            __builtin_amdgcn_wave_barrier();
            const bool ok = spd_inverse6_wave(S.Ppi);
            if (lane == 0) S.sel[1] = ok ? 1 : 0;
        }
        __syncthreads();
        if (wave == 0) {
            if (lane < 36) S.Lm[lane] += S.Ppi[lane];
            __builtin_amdgcn_wave_barrier();
            const bool ok = spd_inverse6_wave(S.Lm) && S.sel[1] != 0;
            if (lane < 6) {
                double sum = 0.0;
                for (int k = 0; k < 6; ++k) sum += S.Lm[lane * 6 + k] * S.eta[k];
                S.xo[lane] = x[lane] + sum;
            }
            if (lane == 0) S.sel[0] = ok ? 0 : 3;
        }
    }
    __syncthreads();
    SKFTICK(8);
    return S.sel[0];
}

// The twist of a frame -> the ring slot a pose lane reads it from.  The lane may be running NEXT to this kernel (frame-granular
// hand-over, EngineArrays::handoff), on a CU of another XCD with an L2 of its own: the six values go out as agent-coherent
// stores (threads 0 .. 5 of wave 0), the wave waits until they are acknowledged, then thread 0 publishes the tag -- no cache
// write-back, no fence.
__device__ __forceinline__ void publish_twist_value(ObjState& st, int slot, int i, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&st.twist_hist[slot][i]), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void publish_twist_tag(ObjState& st, int slot, int frame_idx)
{
    if (threadIdx.x < 64) {   // wave 0: the wave that stored the values
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);   // every store of this wave acknowledged
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (threadIdx.x == 0) __hip_atomic_store(&st.twist_tag[slot], frame_idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One workgroup per object walks the frames of the batch: the velocity belief of frame k is the prior of frame k+1,
// so the recursion is sequential per object; the flow records of all frames are already there (flow_measure_kernel).
#ifndef PRIO_SKF
#define PRIO_SKF 2
#endif
__global__ __launch_bounds__(kSkfThreads) void skf_chain_kernel(EngineArrays a, int reweight)
{
    ROFT_RESIDENT(a, RK_SKF_CHAIN);
    __shared__ SkfShared S;
    __shared__ double s_x[6];
    __shared__ double s_P[36];
    __builtin_amdgcn_s_setprio(PRIO_SKF);   // (a per-object chain of barrier phases next to wide kernels: see ukf_chain_kernel)

    __shared__ FrameCtrl s_c;
    __shared__ int s_npts;
    const int obj = blockIdx.x;
    ObjState& st = a.state[obj];
    const ObjParams& prm = a.params[obj];
    // resident: the pose lanes of the batch may be released next to this kernel (they only ever wait for workgroups that run)
    if (threadIdx.x == 0 && a.skf_started) (void)__hip_atomic_fetch_add(a.skf_started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int t = 0; t < a.T; ++t) {
        const int slot = t * a.n_obj + obj;
        stage_ctrl(&s_c, a.ctrl[slot]);   // (the barrier at the end of the previous frame precedes this overwrite)
        if (threadIdx.x == kSkfThreads - 1) s_npts = a.npts[slot];
        __syncthreads();
        const FrameCtrl& c = s_c;
        const int n_pts = c.vel_stage ? s_npts : -1;
        const int N = n_pts;
        roft_object_output* row = log_row(a, c, obj);
        if (threadIdx.x == 0) st.n_flow_points = n_pts;

        // unobservable / no data: the belief is left exactly as it was (ROFTFilter.cpp:291-301)
        if (N < 3) {
            if (threadIdx.x < 6) {
                const double v = st.v_mean[threadIdx.x];
                publish_twist_value(st, c.twist_slot, threadIdx.x, v);
                if (row) row->twist[threadIdx.x] = v;
            }
            if (threadIdx.x == 0) {
                st.skf_status = (N <= 0) ? 1 : 2;
                if (row) row->n_flow_points = n_pts;
            }
            publish_twist_tag(st, c.twist_slot, c.frame_idx);
            __syncthreads();
            continue;
        }
#ifdef ROFT_SKF_PROFILE
        if (threadIdx.x == 0) S.t0 = wall_clock64();
#endif
        if (threadIdx.x < 6) s_x[threadIdx.x] = st.v_mean[threadIdx.x];  // s^- = s (F = I)
        if (threadIdx.x < 36) {
            const int i = threadIdx.x;
            s_P[i] = st.v_cov[i] + ((i / 6 == i % 6) ? prm.v_q[i / 6] : 0.0);  // P^- = P + Q
        }
        __syncthreads();
        double x[6];
        for (int i = 0; i < 6; ++i) x[i] = s_x[i];
        SKFTICK(0);

        RecAccessor acc(a.recs + (size_t)slot * a.cand_cap, a.cam, c.dt);
        const int rc = skf_core(acc, N, x, s_P, prm.r_flow, reweight, a.norms + (size_t)obj * 3 * a.cand_cap, S);
        // rc 3: numerically singular, belief left unchanged
        if (rc == 0 && threadIdx.x < 36) st.v_cov[threadIdx.x] = S.Lm[threadIdx.x];
        if (threadIdx.x < 6) {
            const double v = (rc == 0) ? S.xo[threadIdx.x] : s_x[threadIdx.x];
            if (rc == 0) st.v_mean[threadIdx.x] = v;
            publish_twist_value(st, c.twist_slot, threadIdx.x, v);
            if (row) row->twist[threadIdx.x] = v;
        }
        if (threadIdx.x == 0) {
            st.skf_status = rc;
#ifdef ROFT_SKF_PROFILE
            for (int i = 0; i < 9; ++i) st.dbg[i] = S.dbg[i];
#endif
            if (row) row->n_flow_points = n_pts;   // (row->outlier_selected belongs to the pose lane that walks the frame)
        }
        publish_twist_tag(st, c.twist_slot, c.frame_idx);
        __syncthreads();   // the belief written here is the next frame's prior (same workgroup)
    }
}

void launch_skf_chain(const EngineArrays& a, int reweight, hipStream_t s, hipEvent_t stop)
{
    hipExtLaunchKernelGGL(skf_chain_kernel, dim3(a.n_obj), dim3(kSkfThreads), 0, s, nullptr, stop, 0, a, reweight);
}

// ---- operator level ---------------------------------------------------------------------------
__global__ __launch_bounds__(kSkfThreads) void skf_arrays_kernel(const double* x_pred, const double* P_pred, int N,
                                                                 const double* y, const double* H, const double* Rdiag,
                                                                 int reweight, double* norms, double* x_out,
                                                                 double* P_out, int* status)
{
    __shared__ SkfShared S;
    if (N <= 0) {  // SKFCorrection.cpp:61-69
        if (threadIdx.x < 6) x_out[threadIdx.x] = x_pred[threadIdx.x];
        if (threadIdx.x < 36) P_out[threadIdx.x] = P_pred[threadIdx.x];
        if (threadIdx.x == 0) *status = 1;
        return;
    }
    double x[6], r[2] = {Rdiag[0], Rdiag[1]};
    for (int i = 0; i < 6; ++i) x[i] = x_pred[i];
    ArrayAccessor acc{y, H};
    const int rc = skf_core(acc, N, x, P_pred, r, reweight, norms, S);
    if (threadIdx.x < 6) x_out[threadIdx.x] = (rc == 0) ? S.xo[threadIdx.x] : x_pred[threadIdx.x];
    if (threadIdx.x < 36) P_out[threadIdx.x] = (rc == 0) ? S.Lm[threadIdx.x] : P_pred[threadIdx.x];
    if (threadIdx.x == 0) *status = rc;
}

// operator level through the accessor the engine uses: compact flow records, H rows rebuilt on the fly
__global__ __launch_bounds__(kSkfThreads) void skf_records_kernel(const double* x_pred, const double* P_pred, int N,
                                                                  const FlowRec* recs, DevCamera cam, double dt,
                                                                  const double* Rdiag, int reweight, double* norms,
                                                                  double* x_out, double* P_out, int* status)
{
    __shared__ SkfShared S;
    if (N <= 0) {
        if (threadIdx.x < 6) x_out[threadIdx.x] = x_pred[threadIdx.x];
        if (threadIdx.x < 36) P_out[threadIdx.x] = P_pred[threadIdx.x];
        if (threadIdx.x == 0) *status = 1;
        return;
    }
    double x[6], r[2] = {Rdiag[0], Rdiag[1]};
    for (int i = 0; i < 6; ++i) x[i] = x_pred[i];
    RecAccessor acc(recs, cam, dt);
    const int rc = skf_core(acc, N, x, P_pred, r, reweight, norms, S);
    if (threadIdx.x < 6) x_out[threadIdx.x] = (rc == 0) ? S.xo[threadIdx.x] : x_pred[threadIdx.x];
    if (threadIdx.x < 36) P_out[threadIdx.x] = (rc == 0) ? S.Lm[threadIdx.x] : P_pred[threadIdx.x];
    if (threadIdx.x == 0) *status = rc;
}

void launch_skf_records(const double* x_pred, const double* P_pred, int N, const FlowRec* recs, DevCamera cam, double dt,
                        const double* Rdiag, int reweight, double* norms, double* x_out, double* P_out, int* status,
                        hipStream_t s)
{
    hipLaunchKernelGGL(skf_records_kernel, dim3(1), dim3(kSkfThreads), 0, s, x_pred, P_pred, N, recs, cam, dt, Rdiag,
                       reweight, norms, x_out, P_out, status);
}

void launch_skf_arrays(const double* x_pred, const double* P_pred, int N, const double* y, const double* H,
                       const double* Rdiag, int reweight, double* norms, double* x_out, double* P_out, int* status,
                       hipStream_t s)
{
    hipLaunchKernelGGL(skf_arrays_kernel, dim3(1), dim3(kSkfThreads), 0, s, x_pred, P_pred, N, y, H, Rdiag, reweight,
                       norms, x_out, P_out, status);
}

__global__ void kf_predict_kernel(const double* x, const double* P, const double* qdiag, double* xo, double* Po)
{
    const int i = threadIdx.x;
    if (i < 6) xo[i] = x[i];
    if (i < 36) Po[i] = P[i] + ((i / 6 == i % 6) ? qdiag[i / 6] : 0.0);
}

void launch_kf_predict(const double* x, const double* P, const double* qdiag, double* xo, double* Po, hipStream_t s)
{
    hipLaunchKernelGGL(kf_predict_kernel, dim3(1), dim3(64), 0, s, x, P, qdiag, xo, Po);
}

}  // namespace roft

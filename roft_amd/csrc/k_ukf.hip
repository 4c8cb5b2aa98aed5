// k_ukf.hip -- pose UKF: sigma-point fan-out, prediction and correction, one workgroup of four wavefronts per
// object (gfx950).
//
// Reference:
//   bfl::UKFPrediction (generic state model) over CartesianQuaternionModel::motion / Q(T)
//                                              src/roft-lib/src/CartesianQuaternionModel.cpp:86-141
//   ROFT::UKFCorrection::correctStep           src/roft-lib/src/UKFCorrection.cpp:54-133
//   CartesianQuaternionMeasurement::{predictedMeasure, innovation}
//                                              src/roft-lib/src/CartesianQuaternionMeasurement.cpp:357-487
//   bfl sigma_point / unscented_transform / UTWeight / quaternion utils: third party
//   (robotology/bayes-filters-lib, un-pinned); algorithm as restated in oracle/ro_ukf.c.
//
// MI355X design.  One workgroup of four wavefronts per object, matrices in LDS.  A UKF step is a chain of
// small dependent phases, so its speed is the instruction latency of the critical path, not throughput:
// thread = matrix entry for the 12x12 products (one pass of 144 threads), thread = sigma point for the fan-out
// and the model evaluations (<= 49 threads of wave 0), and inside the Jacobi rounds wave 0 rotates the matrix
// while wave 1 rotates the eigenvectors.  The matrix square root is U sqrt(S) from a Jacobi
// eigen-decomposition with the round-robin parallel ordering (n/2 disjoint rotations per round, so a 12x12
// sweep is 11 rounds instead of 66 sequential rotations).
// The augmented covariance is block diagonal -- blkdiag(P, Q) or blkdiag(P, R) with R diagonal --
// so only the 12x12 state block (and the 9x9 process-noise block) is ever decomposed, and when an
// outlier-rejection step needs two corrections of the same prediction they share one
// decomposition.  Products this small (12x12x49) do not fill an MFMA tile batch; plain fp64 FMA.
#include "roft_device.h"

// Nothing in this file is compared bit for bit with the oracle (the UKF agrees with it to rounding, not exactly),
// so fused multiply-adds are welcome here although the build default is -ffp-contract=off.
#pragma clang fp contract(fast)

namespace roft {

// phase cycle counters (build with -DROFT_UKF_PROFILE): TICK(L, i) adds the shader cycles since the
// previous tick to slot i
#ifdef ROFT_UKF_PROFILE
#define TICK(L, i) do { __syncthreads(); if (threadIdx.x == 0) { long long _t = clock64(); (L).dbg[i] += _t - (L).t0; (L).t0 = _t; } } while (0)
#else
#define TICK(L, i) do {} while (0)
#endif

constexpr int kUkfThreads = 256;   // four waves, one per SIMD of the CU

// ordering point for code executed by ONE wave of the workgroup (LDS traffic between its lanes)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- quaternion helpers (same conventions as oracle/ro_la.c) -----------------------------------
__device__ __forceinline__ void quat_mul(const double a[4], const double b[4], double o[4])
{
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    const double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}

// sin(n/2)/n and cos(n/2).  The sigma-point rotations are small (a few hundredths of a radian): below a half angle of
// 0.25 rad the Taylor polynomials are exact to the last bit or two (first dropped terms 3e-21 / 1e-23) and replace two
// libm calls, a square root's worth of argument handling and a division; larger angles take the library path.
__device__ __forceinline__ void half_angle(double n, double& sin_over_n, double& c)
{
    const double h = 0.5 * n;
    if (h < 0.25) {
        const double x = h * h;
        const double sinc = 1.0 + x * (-1.0 / 6.0 + x * (1.0 / 120.0 + x * (-1.0 / 5040.0 + x * (1.0 / 362880.0 +
                            x * (-1.0 / 39916800.0 + x * (1.0 / 6227020800.0))))));
        c = 1.0 + x * (-0.5 + x * (1.0 / 24.0 + x * (-1.0 / 720.0 + x * (1.0 / 40320.0 + x * (-1.0 / 3628800.0 +
            x * (1.0 / 479001600.0 + x * (-1.0 / 87178291200.0)))))));
        sin_over_n = 0.5 * sinc;
    } else {
        sin_over_n = sin(h) / n;
        c = cos(h);
    }
}

__device__ __forceinline__ void quat_boxplus(const double q[4], const double r[3], double o[4])
{
    // (the polynomials of half_angle need only n^2: no square root on the small-angle path, which is the chain's)
    const double n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    double s, c;
    if (n2 < 0.25) {
        const double x = 0.25 * n2;
        s = 0.5 * (1.0 + x * (-1.0 / 6.0 + x * (1.0 / 120.0 + x * (-1.0 / 5040.0 + x * (1.0 / 362880.0 +
                   x * (-1.0 / 39916800.0 + x * (1.0 / 6227020800.0)))))));
        c = 1.0 + x * (-0.5 + x * (1.0 / 24.0 + x * (-1.0 / 720.0 + x * (1.0 / 40320.0 + x * (-1.0 / 3628800.0 +
            x * (1.0 / 479001600.0 + x * (-1.0 / 87178291200.0)))))));
    } else {
        half_angle(sqrt(n2), s, c);   // (n = 0 gives (0.5, 1) above: the identity rotation, as the reference's branch does)
    }
    const double qr[4] = {c, s * r[0], s * r[1], s * r[2]};
    quat_mul(qr, q, o);
}

__device__ __forceinline__ double fast_rcp(double d);

__device__ __forceinline__ void quat_diff(const double a[4], const double b[4], double o[3])
{
    const double bc[4] = {b[0], -b[1], -b[2], -b[3]};
    double p[4];
    quat_mul(a, bc, p);
    const double n2 = p[1] * p[1] + p[2] * p[2] + p[3] * p[3];
    const double sgn = (p[0] < 0.0) ? -1.0 : 1.0;
    double k;
    // rotation vector = (2 atan2(n, |p0|) / n) p_vec.  With t = n / |p0| this is (2 / |p0|) (atan t / t) p_vec, and for
    // t^2 <= 0.01 (angles below 0.2 rad) nine terms of the alternating series give atan t / t to 5e-20: no square
    // root, no atan2
    // (one reciprocal -- hardware seed + two Newton steps, below -- instead of two IEEE divisions one after the other)
    const double ip0 = fast_rcp(fabs(p[0]));
    const double t2 = n2 * (ip0 * ip0);
    if (t2 <= 0.01) {
        const double at = 1.0 + t2 * (-1.0 / 3.0 + t2 * (1.0 / 5.0 + t2 * (-1.0 / 7.0 + t2 * (1.0 / 9.0 + t2 * (-1.0 / 11.0 +
                          t2 * (1.0 / 13.0 + t2 * (-1.0 / 15.0 + t2 * (1.0 / 17.0))))))));
        k = sgn * 2.0 * at * ip0;
    } else {
        const double n = sqrt(n2);
        if (n == 0.0) { o[0] = o[1] = o[2] = 0.0; return; }
        k = sgn * 2.0 * atan2(n, fabs(p[0])) / n;
    }
    o[0] = k * p[1]; o[1] = k * p[2]; o[2] = k * p[3];
}

// ---- LDS layout -----------------------------------------------------------------------------------
constexpr int kCols = 50;  // >= 2 * 24 + 1

struct UkfLds {
    double P[144];       // matrix being decomposed (destroyed)
    double V[144];       // its eigenvectors (columns)
    double wP[12];       // its eigenvalues
    double S[144];       // matrix square root the sigma points are drawn from (see decompose_state_cov)
    uint2 jtab[11 * 36]; // Jacobi 12x12: per round and 2x2 block, the four entry offsets (16 bit each)
    unsigned short tri12[78], tri6[21];   // upper triangles in row-major order: (i << 8 | j), i <= j
    double cs[2][6][2];  // rotations (c, s) of the current round, double-buffered by round parity
    double Q[100];       // process noise block padded to 10 x 10
    double VQ[100];
    double wQ[10];
    double rc[12], rs[12];
    int rp[12], rq[12];
    double mean[13];     // mean the sigma points are drawn around
    double cov[144];     // ... and its covariance (kept: P is destroyed)
    double Y[13 * kCols];
    double D[12 * kCols];
    double X[12 * kCols];
    double ymean[13];
    double M4[16], V4[16], w4[4];
    double Py[144], Pxy[144], K[144], KPy[144];
    double aug[12 * 24];
    double innov[12], Kin[12];
    // staged once per step together with the belief (a global load inside a phase costs that phase ~2 k cycles):
    double par[24];      // head of ObjParams: sigma_ang_vel[3] psd_lin_acc[3] v_q[6] R_v[3] R_w[3] R_x[3] R_q[3]
    double meas[13];     // twist of the step [6], pose measurement x [3], q [4]
    double red[4];
    int flag;
    int twist_timeout;   // ukf_one_step: the twist this step was told to wait for never appeared (frame-granular hand-over)
    long long t0;
    long long dbg[32];
};

// 1/x and 1/sqrt(x) from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~24 good bits) plus two Newton
// steps: full double accuracy in ~10 FMAs instead of the ~40-instruction IEEE division / square root
// sequences, which otherwise dominate the serial rotation set-up of every Jacobi round.
__device__ __forceinline__ double fast_rcp(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
}

__device__ __forceinline__ double fast_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    return y;
}

// Symmetric Schur rotation (almost) annihilating apq (apq != 0):
//   tau = (aqq - app) / (2 apq),  t = sgn(tau) / (|tau| + sqrt(1 + tau^2)),  c = 1/sqrt(1 + t^2),  s = t c
// evaluated division-free as t = sgn * |b| / (|a| + sqrt(a^2 + b^2)) with a = aqq - app, b = 2 apq.
// The angle only steers convergence, so t is computed in SINGLE precision (v_rsq_f32 / v_rcp_f32, after scaling
// a and b by a power of two into float range); what must hold to double accuracy is c^2 + s^2 = 1, and c comes
// from a double Newton refinement of the float seed of 1/sqrt(1 + t^2).  The caller applies the rotation to the
// pivot block like to any other block (the off-diagonal is left at ~1e-7 |apq| instead of being forced to 0).
__device__ __forceinline__ void schur_rotation(double app, double aqq, double apq, double& c, double& s)
{
    const double a = aqq - app, b = 2.0 * apq;
    const int e = __builtin_amdgcn_frexp_exp(fmax(fabs(a), fabs(b)));
    const float af = fabsf((float)ldexp(a, -e)), bf = fabsf((float)ldexp(b, -e));   // max(af, bf) in [0.5, 1)
    const float h2 = af * af + bf * bf;
    const float hyp = h2 * __builtin_amdgcn_rsqf(h2);
    const float tf = bf * __builtin_amdgcn_rcpf(af + hyp);
    const double t = ((a >= 0.0) == (b >= 0.0) || a == 0.0) ? (double)tf : -(double)tf;
    const double x = fma(t, t, 1.0);
    double y = (double)__builtin_amdgcn_rsqf((float)x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    c = y;
    s = t * y;
}

#ifndef ROFT_JACOBI_TOL
#define ROFT_JACOBI_TOL 1e-6
#endif
constexpr double kJacobiTol = ROFT_JACOBI_TOL;

// pair i (0 <= i < n/2) of round `round` of the round-robin ordering, p < q
__device__ __forceinline__ void jacobi_pair(int n, int round, int i, int& p, int& q)
{
    if (i == 0) { p = n - 1; q = round; }
    else {
        p = round + i; if (p >= n - 1) p -= n - 1;
        q = round - i; if (q < 0) q += n - 1;
    }
    if (p > q) { const int t = p; p = q; q = t; }
}

// Symmetric Jacobi eigen-decomposition in LDS, n even (<= 12).  On return diag(A) = eigenvalues (plus an
// off-diagonal rest below kJacobiTol), columns of V = eigenvectors.
//
// Round-robin parallel ordering: each round rotates n/2 disjoint index pairs (p_i, q_i).  With
// all pairs fixed, A' = J'AJ decomposes into (n/2)^2 independent 2x2 blocks,
//   block(a, b) = rows {p_a, q_a} x cols {p_b, q_b}   ->   R_a' * block * R_b,
// and V' = V J into as many blocks (rows {p_a, q_a} x cols {p_b, q_b}) * R_b.
//
// jacobi_wave(): generic n, executed by the lanes of WAVE 0 only (the other waves must not call it); lane (a, b)
// owns one block of A and one of V, rotations travel by wave shuffle.  Used for the 10x10 explicit process noise
// of the operator-level entry point.
__device__ __noinline__ void jacobi_wave(double* A, double* V, int n, UkfLds& L)
{
    const int lane = threadIdx.x;
    for (int i = lane; i < n * n; i += 64) V[i] = ((i / n) == (i % n)) ? 1.0 : 0.0;
    wave_sync();
    const int half = n / 2;
    const bool active = lane < half * half;
    const int ba = active ? lane / half : 0, bb = active ? lane % half : 0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = lane; i < n * n; i += 64) {
            const int r = i / n, cidx = i % n;
            const double v = A[i];
            if (r == cidx) dg += v * v; else if (r < cidx) off += v * v;
        }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dg += __shfl_xor(dg, o, 64); }
        if (off <= kJacobiTol * kJacobiTol * dg || off == 0.0) break;
        for (int round = 0; round < n - 1; ++round) {
            int pa, qa, pb, qb;
            jacobi_pair(n, round, ba, pa, qa);
            jacobi_pair(n, round, bb, pb, qb);
            double a00 = 0, a01 = 0, a10 = 0, a11 = 0, v00 = 0, v01 = 0, v10 = 0, v11 = 0;
            if (active) {
                a00 = A[pa * n + pb]; a01 = A[pa * n + qb]; a10 = A[qa * n + pb]; a11 = A[qa * n + qb];
                v00 = V[pa * n + pb]; v01 = V[pa * n + qb]; v10 = V[qa * n + pb]; v11 = V[qa * n + qb];
            }
            double c = 1.0, s = 0.0;
            if (active && ba == bb && a01 != 0.0) schur_rotation(a00, a11, a01, c, s);
            const double ca = __shfl(c, ba * half + ba, 64), sa = __shfl(s, ba * half + ba, 64);
            const double cb = __shfl(c, bb * half + bb, 64), sb = __shfl(s, bb * half + bb, 64);
            if (active) {
                double b00 = cb * a00 - sb * a01, b01 = sb * a00 + cb * a01;
                double b10 = cb * a10 - sb * a11, b11 = sb * a10 + cb * a11;
                double c00 = ca * b00 - sa * b10, c10 = sa * b00 + ca * b10;
                double c01 = ca * b01 - sa * b11, c11 = sa * b01 + ca * b11;
                if (ba == bb) { c01 = c10 = 0.5 * (c01 + c10); }   // keep the pivot block symmetric
                A[pa * n + pb] = c00; A[pa * n + qb] = c01; A[qa * n + pb] = c10; A[qa * n + qb] = c11;
                V[pa * n + pb] = cb * v00 - sb * v01; V[pa * n + qb] = sb * v00 + cb * v01;
                V[qa * n + pb] = cb * v10 - sb * v11; V[qa * n + qb] = sb * v10 + cb * v11;
            }
            wave_sync();
        }
    }
    wave_sync();
}

// Entry offsets of the 36 blocks of the 11 rounds of a 12x12 sweep, built once per launch.
__device__ void jacobi12_table(UkfLds& L)
{
    for (int i = threadIdx.x; i < 11 * 36; i += kUkfThreads) {
        const int round = i / 36, l = i % 36;
        int pa, qa, pb, qb;
        jacobi_pair(12, round, l / 6, pa, qa);
        jacobi_pair(12, round, l % 6, pb, qb);
        L.jtab[i] = make_uint2((uint32_t)(pa * 12 + pb) | ((uint32_t)(pa * 12 + qb) << 16),
                               (uint32_t)(qa * 12 + pb) | ((uint32_t)(qa * 12 + qb) << 16));
    }
    for (int e = threadIdx.x; e < 78 + 21; e += kUkfThreads) {
        const int m = e < 78 ? 12 : 6;
        int u = e < 78 ? e : e - 78, i = 0;
        while (u >= m - i) { u -= m - i; ++i; }
        (e < 78 ? L.tri12[e] : L.tri6[e - 78]) = (unsigned short)((i << 8) | (i + u));
    }
}

// jacobi12(): the 12x12 state covariance, called by the WHOLE workgroup.  Wave 0 owns the 36 blocks of A and
// computes the six rotations of a round from its diagonal blocks; wave 1 owns the 36 blocks of V.  The rotations
// cross from wave 0 to wave 1 through L.cs (double-buffered by round parity), so a round costs ONE workgroup
// barrier and each wave issues about half of the instructions a single wave would (the rounds are bound by the
// instruction latency of one wave, not by throughput).  Waves 2 and 3 only keep the barrier count.
__device__ __noinline__ void jacobi12(double* A, double* V, UkfLds& L)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 144; i += kUkfThreads) V[i] = ((i / 12) == (i % 12)) ? 1.0 : 0.0;
    __syncthreads();
    const bool act = lane < 36 && wave < 2;
    const int ba = act ? lane / 6 : 0, bb = act ? lane % 6 : 0;
    double* M = (wave == 0) ? A : V;
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (wave == 0) {
            double off = 0.0, dg = 0.0;
            for (int i = lane; i < 144; i += 64) {
                const int r = i / 12, cidx = i % 12;
                const double v = A[i];
                if (r == cidx) dg += v * v; else if (r < cidx) off += v * v;
            }
            for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dg += __shfl_xor(dg, o, 64); }
            // stop when ||off-diagonal|| <= kJacobiTol ||diagonal||; decompose_state_cov() absorbs the remaining
            // off-diagonal part into the square root to first order
            if (lane == 0) L.flag = (off <= kJacobiTol * kJacobiTol * dg || off == 0.0) ? 1 : 0;
        }
        __syncthreads();
        if (L.flag) break;   // (rewritten only after the barriers of the next rounds)
#ifdef ROFT_UKF_PROFILE
        if (tid == 0) L.dbg[16] += 1;
#endif
        for (int round = 0; round < 11; ++round) {
            int o00 = 0, o01 = 0, o10 = 0, o11 = 0;
            double x00 = 0, x01 = 0, x10 = 0, x11 = 0;
            if (act) {
                const uint2 pk = L.jtab[round * 36 + lane];
                o00 = pk.x & 0xFFFF; o01 = pk.x >> 16; o10 = pk.y & 0xFFFF; o11 = pk.y >> 16;
                x00 = M[o00]; x01 = M[o01]; x10 = M[o10]; x11 = M[o11];
                if (wave == 0 && ba == bb) {   // diagonal block: (app, apq; apq, aqq)
                    double c = 1.0, s = 0.0;
                    if (x01 != 0.0) schur_rotation(x00, x11, x01, c, s);
                    L.cs[round & 1][ba][0] = c;
                    L.cs[round & 1][ba][1] = s;
                }
            }
            __syncthreads();
            if (act) {
                const double cb = L.cs[round & 1][bb][0], sb = L.cs[round & 1][bb][1];
                // columns (p_b, q_b): x_p' = c x_p - s x_q, x_q' = s x_p + c x_q
                double b00 = cb * x00 - sb * x01, b01 = sb * x00 + cb * x01;
                double b10 = cb * x10 - sb * x11, b11 = sb * x10 + cb * x11;
                if (wave == 0) {
                    const double ca = L.cs[round & 1][ba][0], sa = L.cs[round & 1][ba][1];
                    // rows (p_a, q_a)
                    const double c00 = ca * b00 - sa * b10, c10 = sa * b00 + ca * b10;
                    const double c01 = ca * b01 - sa * b11, c11 = sa * b01 + ca * b11;
                    b00 = c00; b11 = c11;
                    b01 = c01; b10 = c10;
                    if (ba == bb) { b01 = b10 = 0.5 * (c01 + c10); }   // keep the pivot block symmetric
                }
                M[o00] = b00; M[o01] = b01; M[o10] = b10; M[o11] = b11;
            }
        }
    }
    __syncthreads();
}

// sum_c a[c] w_c b[c] over the sigma columns (column 0 carries wc0, the rest the common weight wci), with four
// independent partial sums so that the additions do not form one 40-deep dependency chain; eight columns (16 LDS
// reads) per iteration of the main loop.
__device__ __forceinline__ double weighted_dot(const double* ar, const double* br, int ncols, double wc0, double wci)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int c = 1;
    for (; c + 7 < ncols; c += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = ar[c + u]; bv[u] = br[c + u]; }
        s0 = fma(av[0], bv[0], s0); s1 = fma(av[1], bv[1], s1); s2 = fma(av[2], bv[2], s2); s3 = fma(av[3], bv[3], s3);
        s0 = fma(av[4], bv[4], s0); s1 = fma(av[5], bv[5], s1); s2 = fma(av[6], bv[6], s2); s3 = fma(av[7], bv[7], s3);
    }
    for (; c + 3 < ncols; c += 4) {
        s0 = fma(ar[c], br[c], s0);
        s1 = fma(ar[c + 1], br[c + 1], s1);
        s2 = fma(ar[c + 2], br[c + 2], s2);
        s3 = fma(ar[c + 3], br[c + 3], s3);
    }
    for (; c < ncols; ++c) s0 = fma(ar[c], br[c], s0);
    return fma(ar[0] * br[0], wc0, ((s0 + s1) + (s2 + s3)) * wci);
}

// The same sum shared by TWO neighbouring lanes (an even / odd pair of one wave): `part` 0 takes the columns [1, mid),
// part 1 the columns [mid, ncols), the halves meet through a quad_perm swap, and both lanes return the whole sum.  A
// 12 x 12 covariance from 43 sigma columns is 78 distinct sums; one lane each they all take one 43-term chain of LDS
// reads and dependent FMAs -- two lanes each they take half of it (156 lanes).  Every lane of the wave must call.
__device__ __forceinline__ double weighted_dot_pair(const double* ar, const double* br, int ncols, double wc0, double wci, int part)
{
    const int mid = 1 + ((ncols - 1) >> 1);
    int c = part ? mid : 1;
    const int end = part ? ncols : mid;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (; c + 7 < end; c += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = ar[c + u]; bv[u] = br[c + u]; }
        s0 = fma(av[0], bv[0], s0); s1 = fma(av[1], bv[1], s1); s2 = fma(av[2], bv[2], s2); s3 = fma(av[3], bv[3], s3);
        s0 = fma(av[4], bv[4], s0); s1 = fma(av[5], bv[5], s1); s2 = fma(av[6], bv[6], s2); s3 = fma(av[7], bv[7], s3);
    }
    for (; c + 3 < end; c += 4) {
        s0 = fma(ar[c], br[c], s0);
        s1 = fma(ar[c + 1], br[c + 1], s1);
        s2 = fma(ar[c + 2], br[c + 2], s2);
        s3 = fma(ar[c + 3], br[c + 3], s3);
    }
    for (; c < end; ++c) s0 = fma(ar[c], br[c], s0);
    double sum = (s0 + s1) + (s2 + s3);
    sum += __shfl_xor(sum, 1, 64);
    return fma(ar[0] * br[0], wc0, sum * wci);
}

// Weighted means of `nrows` linear rows of Y (row stride kCols), eight lanes per row: lane l of the calling group
// (l = 0 .. 8 nrows - 1, all inside one wave and 8-aligned) sums the columns l % 8, l % 8 + 8, ..., then the eight
// partial sums meet through row_shr DPP moves.  One lane per row reading 40-odd columns one after the other paid an
// LDS latency per column.
__device__ __forceinline__ void linear_means8(const double* Y, int nrows, int ncols, double wm0, double wmi, int l,
                                              double* out)
{
    const int row = l >> 3, part = l & 7;
    double sum = 0.0;
    if (row < nrows) {
        const double* y = Y + row * kCols;
        double v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) v[u] = y[min(part + 8 * u, ncols - 1)];   // kCols <= 56 columns
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int c = part + 8 * u;
            if (c < ncols) sum = fma(v[u], (c == 0) ? wm0 : wmi, sum);
        }
    }
    // lanes 8k .. 8k+7 -> lane 8k+7 (row_shr stays inside a 16-lane DPP row; 8-aligned groups never straddle one)
    sum += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(sum), 0x111, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(sum), 0x111, 0xf, 0xf, false));
    sum += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(sum), 0x112, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(sum), 0x112, 0xf, 0xf, false));
    sum += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(sum), 0x114, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(sum), 0x114, 0xf, 0xf, false));
    if (row < nrows && part == 7) out[row] = sum;
}

// dominant eigenvector of M = sum_c wm_c q_c q_c' (rows qrow..qrow+3 of Y), sign aligned with column 0.
// The sigma quaternions are a tight cluster, so M is a rank-one matrix plus a perturbation of the size of
// the rotational covariance: eigenvalue gap ratio r = lambda_2 / lambda_1 << 1.  Power iteration on
// M^16 (four squarings, 16 lanes, renormalised) converges like r^16 per step from the central sigma point;
// iterate to a fixed point in double.  Same vector as a 4x4 eigen-solver returns, at a tenth of the cost of
// a Jacobi sweep sequence.
// Called by the whole workgroup; the work is done by wave 0 (the other waves are free to do something else
// before the closing barrier, e.g. the linear rows of the mean -- see the callers).
__device__ void quaternion_mean(const double* Y, int qrow, int ncols, double wm0, double wmi, double out[4],
                                UkfLds& L)
{
    const int lane = threadIdx.x;
    if (lane < 64) {
        if (lane < 16) {
            const int i = lane / 4, j = lane % 4;
            L.M4[lane] = weighted_dot(Y + (qrow + i) * kCols, Y + (qrow + j) * kCols, ncols, wm0, wmi);
        }
        wave_sync();
        // Tight cluster (the usual case: eigenvalue ratio ~ var(theta) / 4): plain power iteration from the central
        // sigma point gains five digits per step and is at its fixed point after three or four -- every lane of the
        // wave runs it redundantly, so the outcome is wave-uniform without a broadcast.  Wide clusters fall through
        // to the squaring scheme below.
        {
            double m[16], v[4];
#pragma unroll
            for (int i = 0; i < 16; ++i) m[i] = L.M4[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = Y[(qrow + i) * kCols + 0];
            bool done = false;
            for (int it = 0; it < 6 && !done; ++it) {
                double w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = m[i * 4] * v[0] + m[i * 4 + 1] * v[1] + m[i * 4 + 2] * v[2] + m[i * 4 + 3] * v[3];
                const double inv = fast_rsqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3]);
                double diff = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) { w[i] *= inv; diff = fmax(diff, fabs(w[i] - v[i])); v[i] = w[i]; }
                done = diff < 4e-16;
            }
            if (done) {
                if (lane == 0)
                    for (int i = 0; i < 4; ++i) L.w4[i] = v[i];
                goto finished;
            }
        }
        for (int it = 0; it < 4; ++it) {  // M <- M^2 / trace-normalised, ping-pong M4 <-> V4
            double* src = (it & 1) ? L.V4 : L.M4;
            double* dst = (it & 1) ? L.M4 : L.V4;
            if (lane < 16) {
                const int i = lane / 4, j = lane % 4;
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += src[i * 4 + k] * src[k * 4 + j];
                const double tr = src[0] * src[0] + src[5] * src[5] + src[10] * src[10] + src[15] * src[15];
                dst[lane] = s * fast_rcp(tr);   // keeps the entries O(1); any positive scale is fine
            }
            wave_sync();
        }
        if (lane == 0) {
            const double* M16 = L.M4;  // after 4 ping-pongs the result is back in M4
            double v[4];
            for (int i = 0; i < 4; ++i) v[i] = Y[(qrow + i) * kCols + 0];
            for (int it = 0; it < 32; ++it) {
                double w[4];
                for (int i = 0; i < 4; ++i) w[i] = M16[i * 4] * v[0] + M16[i * 4 + 1] * v[1] + M16[i * 4 + 2] * v[2] + M16[i * 4 + 3] * v[3];
                const double inv = fast_rsqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3]);
                double diff = 0.0;
                for (int i = 0; i < 4; ++i) { w[i] *= inv; diff = fmax(diff, fabs(w[i] - v[i])); v[i] = w[i]; }
                if (diff < 4e-16) break;
            }
            for (int i = 0; i < 4; ++i) L.w4[i] = v[i];
        }
    finished:;
    }
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[i] = L.w4[i];
}

// C (ra x rb, leading dim ldc) = A diag(w) B'
__device__ void weighted_outer(const double* A, int ra, const double* B, int rb, int ncols, double wc0, double wci,
                               double* C, int ldc)
{
    for (int e = threadIdx.x; e < ra * rb; e += kUkfThreads) {
        const int i = e / rb, j = e % rb;
        C[i * ldc + j] = weighted_dot(A + i * kCols, B + j * kCols, ncols, wc0, wci);
    }
}

// lane SRC of every row of 16 lanes -> all lanes of the row: DPP row_newbcast (gfx90a+).  Two 32-bit DPP moves per double: no
// SGPR round trip and no hazard wait states as with v_readlane (two of them + an s_nop per use).  (v_fmac_f64_dpp takes the
// control directly -- one instruction per update -- and is no faster: measured, DESIGN.md App. A.)
template <int SRC>
__device__ __forceinline__ double rowbcast_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x150 + SRC, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x150 + SRC, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int M, int J, int K>
__device__ __forceinline__ void chol_update(double (&a)[M], double lij)
{
    if constexpr (K < M) {
        a[K] = fma(-lij, rowbcast_f64<K>(lij), a[K]);
        chol_update<M, J, K + 1>(a, lij);
    }
}

template <int M, int J>
__device__ __forceinline__ void chol_columns(double (&a)[M], double (&rinv)[M], int& ok, int lane)
{
    if constexpr (J < M) {
        const double d = rowbcast_f64<J>(a[J]);   // A_jj - sum_k<j L_jk^2
        if (!(d > 0.0)) ok = 0;
        const double r = fast_rsqrt(d);
        rinv[J] = r;
        const double lij = (lane == J) ? d * r : a[J] * r;
        a[J] = lij;
        chol_update<M, J, J + 1>(a, lij);
        chol_columns<M, J + 1>(a, rinv, ok, lane);
    }
}

// Cholesky factor of the SPD M x M matrix A (ld M) into the lower triangle of Lc (ld M; the strict upper triangle
// is left untouched) + reciprocal diagonal.  Called by the whole workgroup; the work is done by lanes 0..M-1 of
// wave 0, lane = row, the row lives in registers and the column being eliminated is broadcast over the row of 16 lanes by DPP row_newbcast
// (no LDS round trip inside the factorisation).  Returns false if A is not positive definite.
template <int M>
__device__ __forceinline__ bool cholesky_rows(const double* A, double* Lc, double* inv_diag, UkfLds& L)
{
    const int lane = threadIdx.x;
    if (lane < 64) {
        double a[M];
#pragma unroll
        for (int k = 0; k < M; ++k) a[k] = (lane < M) ? A[lane * M + k] : 0.0;
        double rinv[M];
        int ok = 1;
        static_assert(M <= 16, "the rows of the matrix sit in one DPP row of 16 lanes");
        chol_columns<M, 0>(a, rinv, ok, lane);
        if (lane < M) {
#pragma unroll
            for (int k = 0; k < M; ++k) if (k <= lane) Lc[lane * M + k] = a[k];
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < M; ++k) inv_diag[k] = rinv[k];
            L.flag = ok;
        }
    }
    __syncthreads();
    return L.flag != 0;
}

// K (12 x M) = Pxy (12 x M) * inv(Lc Lc'), one lane per row of K: two triangular solves in registers
template <int M>
__device__ void chol_solve_rows(const double* Lc, const double* inv_diag, const double* Pxy, double* K)
{
    const int lane = threadIdx.x;
    if (lane >= 12) return;
    double w[M];
#pragma unroll
    for (int r = 0; r < M; ++r) {
        double s = Pxy[lane * M + r];
#pragma unroll
        for (int k = 0; k < r; ++k) s -= Lc[r * M + k] * w[k];
        w[r] = s * inv_diag[r];
    }
#pragma unroll
    for (int r = M - 1; r >= 0; --r) {
        double s = w[r];
#pragma unroll
        for (int k = r + 1; k < M; ++k) s -= Lc[k * M + r] * w[k];
        w[r] = s * inv_diag[r];
    }
#pragma unroll
    for (int r = 0; r < M; ++r) K[lane * M + r] = w[r];
}

struct UtW {
    double c, wm0, wc0, wi;
    double sc;          // sqrt(c): scale of the square-root columns
    int ncols;
};

// Unscented-transform weights for an augmented dimension n.  Evaluated on the HOST at launch for the three
// dimensions the filter uses (12 + 6, 12 + 9, 12 + 12) and handed to the kernel by value: three double divisions
// and a square root at the top of the prediction and of every correction are ~1 k cycles of a single wave each.
// (Host and device both round these IEEE operations correctly, so the values are the same.)
inline UtW ut_weights(int n, const roft_ut_params& ut)
{
    UtW w;
    const double lambda = ut.alpha * ut.alpha * (n + ut.kappa) - n;
    w.c = n + lambda;
    w.ncols = 2 * n + 1;
    w.wm0 = lambda / (n + lambda);
    w.wc0 = lambda / (n + lambda) + (1.0 - ut.alpha * ut.alpha + ut.beta);
    w.wi = 1.0 / (2.0 * (n + lambda));
    w.sc = std::sqrt(w.c);
    return w;
}

struct UtTable {
    UtW w[3];           // n = 18, 21, 24
};

__device__ __forceinline__ const UtW& ut_lookup(const UtTable& t, int n) { return t.w[(n - 18) / 3]; }

// perturbation of sigma column `col`: state part d[12] (from the decomposition of L.P, already done)
// and noise part dn[r].  noise_eig: use (L.VQ, L.wQ) (process noise) else a diagonal noise covariance whose
// entry for THIS column is noise_var.
__device__ void sigma_perturbation(int col, int r, double sc, bool noise_eig, double noise_var,
                                   const UkfLds& L, double d[12], double dn[12])
{
    const int n = 12 + r;
    // column 0 is the mean; columns 1..n add, columns n+1..2n subtract the k-th square-root column
    const int k = (col <= n) ? col - 1 : col - 1 - n;
    const double amp = (col == 0) ? 0.0 : ((col <= n) ? sc : -sc);
    const bool state_col = (col > 0) && (k < 12);
    const int kk = (k >= 12) ? k - 12 : 0;
    // one square root for every lane, in front of the divergent part (noise columns only use it)
    const double var = noise_eig ? L.wQ[min(kk, 9)] : noise_var;
    const double sdev = sqrt(fabs(var));
#pragma unroll
    for (int i = 0; i < 12; ++i) { d[i] = 0.0; dn[i] = 0.0; }
    if (state_col) {
#pragma unroll
        for (int i = 0; i < 12; ++i) d[i] = amp * L.S[i * 12 + k];
    } else if (col > 0) {
        if (noise_eig) {
            for (int i = 0; i < r; ++i) dn[i] = amp * L.VQ[i * 10 + kk] * sdev;
        } else {
            const double v = amp * 1.0 * sdev;
#pragma unroll
            for (int i = 0; i < 12; ++i) if (i == kk) dn[i] = v;
        }
    }
}

// Decompose L.cov and form the square root L.S the sigma points are drawn from (called by the whole workgroup).
//
// Warm start: consecutive frames have nearly the same covariance, so the previous eigenvector basis
// V0 almost diagonalises it; Jacobi on B = V0' P V0 then needs 1-2 sweeps instead of ~8, and
// V = V0 VB.  Any orthogonal V0 is valid (B is similar to P), so this changes the result only by
// rounding.  A cold start every kWarmRefresh uses stops the orthogonality error of the running
// product from accumulating.
//
// Square root: the Jacobi sweeps stop at B = V' cov V = D + E with a small off-diagonal rest E
// (||E|| <= kJacobiTol ||D||) instead of running one more sweep to annihilate it.  The symmetric square root of
// B is D^1/2 + X1 + X2 + O(E^3) with X1_ij = E_ij / (sqrt(d_i) + sqrt(d_j)) and X2_ij = -(X1^2)_ij / (sqrt(d_i) +
// sqrt(d_j)) -- no eigenvalue gap in any denominator -- so S = V (D^1/2 + X1 + X2) satisfies S S' = cov up to
// O(E^3), and S equals the exact U sqrt(Lambda) times an orthogonal matrix within O(E) of the identity: the sigma
// set the reference draws, to rounding.
__device__ __forceinline__ void decompose_state_cov(UkfLds& L, double* warm, int* warm_age)
{
    const int tid = threadIdx.x;
    const int age = warm ? *warm_age : 0;
    const bool use = warm && age > 0 && age < kWarmRefresh;
    double* V0 = L.K;      // scratch: K / KPy / Pxy are free until the correction computes them
    double* T = L.KPy;
    double* M = L.Pxy;
    const double* Vfull = L.V;
    if (use) {
        if (tid < 144) V0[tid] = warm[tid];
        __syncthreads();
        if (tid < 144) {
            const int i = tid / 12, j = tid % 12;
            double s = 0.0;
            for (int k = 0; k < 12; ++k) s += L.cov[i * 12 + k] * V0[k * 12 + j];
            T[tid] = s;
        }
        __syncthreads();
        if (tid < 144) {
            const int r = tid / 12, c = tid % 12;
            const int i = r < c ? r : c, j = r < c ? c : r;   // both triangles evaluate the same sum
            double s = 0.0;
            for (int k = 0; k < 12; ++k) s += V0[k * 12 + i] * T[k * 12 + j];
            L.P[tid] = s;
        }
        __syncthreads();
        jacobi12(L.P, L.V, L);   // L.V = VB
        if (tid < 144) {
            const int i = tid / 12, j = tid % 12;
            double s = 0.0;
            for (int k = 0; k < 12; ++k) s += V0[i * 12 + k] * L.V[k * 12 + j];
            T[tid] = s;          // V = V0 VB
        }
        Vfull = T;
    } else {
        if (tid < 144) L.P[tid] = L.cov[tid];
        __syncthreads();
        jacobi12(L.P, L.V, L);
    }
    double inv_ss = 0.0, m1 = 0.0;
    if (tid < 144) {
        const int i = tid / 12, j = tid % 12;
        const double si = sqrt(fabs(L.P[i * 13])), sj = sqrt(fabs(L.P[j * 13]));
        inv_ss = (si + sj > 0.0) ? fast_rcp(si + sj) : 0.0;
        if (i == j) m1 = si;
        else m1 = 0.5 * (L.P[i * 12 + j] + L.P[j * 12 + i]) * inv_ss;
        M[tid] = (i == j) ? 0.0 : m1;      // X1 (zero diagonal)
    }
    __syncthreads();
    if (tid < 144) {
        // second-order term: D^1/2 X2 + X2 D^1/2 = -X1^2
        const int i = tid / 12, j = tid % 12;
        double x1sq = 0.0;
        for (int k = 0; k < 12; ++k) x1sq += M[i * 12 + k] * M[k * 12 + j];
        m1 -= x1sq * inv_ss;
    }
    __syncthreads();
    if (tid < 144) M[tid] = m1;            // D^1/2 + X1 + X2
    __syncthreads();
    if (tid < 144) {
        const int i = tid / 12, j = tid % 12;
        double acc = 0.0;
        for (int k = 0; k < 12; ++k) acc += Vfull[i * 12 + k] * M[k * 12 + j];
        L.S[tid] = acc;
        if (warm) warm[tid] = Vfull[tid];
    }
    if (tid < 12) L.wP[tid] = L.P[tid * 13];
    if (warm && tid == 0) *warm_age = use ? age + 1 : 1;
    __syncthreads();
}

// ---- prediction: (L.mean, L.cov) -> (L.mean, L.cov) ------------------------------------------------
// Square root of L.cov when the sigma set may come from any square root (see roft_config::ukf_cholesky_guard*):
// lower Cholesky factor into L.S.  `for_correction`: the guard of the correction (else of the prediction over the
// sampling time T).  Returns false (nothing usable in L.S) when the guard does not hold or the matrix is not
// numerically positive definite; the caller then decomposes.
__device__ __noinline__ bool cholesky_state_sqrt(UkfLds& L, bool for_correction, double T, double guard_rot, double guard_bil)
{
    if (!(guard_rot > 0.0)) return false;
    const double th = fmax(L.cov[9 * 13], fmax(L.cov[10 * 13], L.cov[11 * 13]));   // var(theta)
    const double wv = fmax(L.cov[3 * 13], fmax(L.cov[4 * 13], L.cov[5 * 13]));     // var(omega)
    const double xx = fmax(L.cov[6 * 13], fmax(L.cov[7 * 13], L.cov[8 * 13]));     // var(x)
    // (uniform over the workgroup; comparisons written so that NaN fails them)
    const bool ok = for_correction ? (th <= guard_rot && wv * xx <= guard_bil) : (th + T * T * wv <= guard_rot);
#ifdef ROFT_UKF_PROFILE
    if (threadIdx.x == 0 && for_correction) { L.dbg[20] = (long long)(th * 1e9); L.dbg[21] = (long long)(wv * 1e9); L.dbg[22] = (long long)(xx * 1e9); }
#endif
    if (!ok) return false;
    if (threadIdx.x < 144) L.S[threadIdx.x] = 0.0;
    __syncthreads();
    return cholesky_rows<12>(L.cov, L.S, L.rc, L);
}

// Additive noise in closed form (round 5).  The process noise enters the motion model additively on its linear OUTPUT rows
// ([v, w, x]' = [v, w, x] + n with v, w taken before the noise, CartesianQuaternionModel.cpp:94-103) and the augmented
// covariance is block diagonal, so the 18 sigma columns that perturb a noise dimension are f(mean) +- sqrt(c) a_k on the linear
// rows and f(mean) on the quaternion: in the weighted sums they are column 0 eighteen more times -- weights wm0 + 18 wi and
// wc0 + 18 wi -- plus, in the covariance, wi sum_k 2 c a_k a_k' = Q exactly (wi = 1 / 2c, sum_k a_k a_k' = Q for ANY square root
// of Q; the +- cross terms cancel).  The model is evaluated on the mean and the 24 state columns only (25 instead of 43), Q(T)
// is added as it is -- no square root of Q at all --, and what changes is the rounding of sums that the reference takes over
// 43 terms: the oracle keeps the 43 columns, the parity bar (1e-9) is unchanged.
__device__ void ukf_predict(UkfLds& L, const ObjParams& prm, double T, const UtTable& ut, double* warm,
                            int* warm_age, double chol_guard)
{
    const int lane = threadIdx.x;
    const int n = 21;
    const UtW& w = ut_lookup(ut, n);
    const double sc = w.sc;
    constexpr int kPredCols = 25;                    // the mean + 12 state dof, plus and minus
    const double wm0 = w.wm0 + 18.0 * w.wi, wc0 = w.wc0 + 18.0 * w.wi;   // (column 0 stands for the 18 noise columns as well)

    // process noise block Q(T) (CartesianQuaternionModel.cpp:127-141), padded to 10 x 10: L.Q, rows / columns [v(3) w(3) x(3)]
    if (prm.q_override) {
        for (int i = lane; i < 100; i += kUkfThreads) L.Q[i] = 0.0;
        __syncthreads();
        for (int i = lane; i < 81; i += kUkfThreads) L.Q[(i / 9) * 10 + (i % 9)] = prm.q_override[i];
    } else if (lane >= 192 && lane < 195) {
        // Q(T) couples only (v_i, x_i); three lanes of wave 3, concurrently with the square root of the state covariance on
        // wave 0 (L.Q is zero from the kernel's start and only these fifteen entries change; the barrier that closes the
        // square-root phase orders them before they are read)
        const int i = lane - 192;
        const double psd = L.par[3 + i];
        L.Q[i * 10 + i] = psd * T;
        L.Q[(6 + i) * 10 + (6 + i)] = psd * (T * T * T / 3.0);
        L.Q[i * 10 + (6 + i)] = L.Q[(6 + i) * 10 + i] = psd * (T * T / 2.0);
        L.Q[(3 + i) * 10 + (3 + i)] = L.par[i];
    }
    TICK(L, 1);
    if (!cholesky_state_sqrt(L, false, T, chol_guard, 0.0)) decompose_state_cov(L, warm, warm_age);
    TICK(L, 2);

    // fan-out + motion model, lane = sigma point
    if (lane < kPredCols) {
        double d[12], dn[12];
        sigma_perturbation(lane, 0, sc, false, 0.0, L, d, dn);   // (r = 0: columns 1 .. 12 add, 13 .. 24 subtract a state column)
        double v[3], wv[3], x[3], q[4];
        for (int i = 0; i < 3; ++i) {
            v[i] = L.mean[i] + d[i];
            wv[i] = L.mean[3 + i] + d[3 + i];
            x[i] = L.mean[6 + i] + d[6 + i];
        }
        quat_boxplus(L.mean + 9, d + 9, q);
        for (int i = 0; i < 3; ++i) {
            L.Y[i * kCols + lane] = v[i];
            L.Y[(3 + i) * kCols + lane] = wv[i];
            L.Y[(6 + i) * kCols + lane] = x[i] + v[i] * T;  // v without noise (cpp:94-97)
        }
        const double norm_w = sqrt(wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2]) + 2.220446049250313e-16;
        double s, c;   // sin(|w| T / 2) / |w| and cos(|w| T / 2)
        half_angle(norm_w * T, s, c);
        s *= T;
        L.Y[9 * kCols + lane] = c * q[0] + s * (-wv[0] * q[1] - wv[1] * q[2] - wv[2] * q[3]);
        L.Y[10 * kCols + lane] = c * q[1] + s * (wv[0] * q[0] - wv[2] * q[2] + wv[1] * q[3]);
        L.Y[11 * kCols + lane] = c * q[2] + s * (wv[1] * q[0] + wv[2] * q[1] - wv[0] * q[3]);
        L.Y[12 * kCols + lane] = c * q[3] + s * (wv[2] * q[0] - wv[1] * q[1] + wv[0] * q[2]);
    }
    __syncthreads();

    TICK(L, 3);
    // mean: the linear rows on wave 1 while wave 0 works on the quaternion rows
    if (lane >= 64 && lane < 192) linear_means8(L.Y, 9, kPredCols, wm0, w.wi, lane - 64, L.ymean);   // waves 1 and 2
    double qm[4];
    quaternion_mean(L.Y, 9, kPredCols, wm0, w.wi, qm, L);
    TICK(L, 4);
    if (lane == 64)
        for (int i = 0; i < 4; ++i) L.ymean[9 + i] = qm[i];
    // deviations
    if (lane < kPredCols) {
        for (int i = 0; i < 9; ++i) L.D[i * kCols + lane] = L.Y[i * kCols + lane] - L.ymean[i];
        const double q[4] = {L.Y[9 * kCols + lane], L.Y[10 * kCols + lane], L.Y[11 * kCols + lane],
                             L.Y[12 * kCols + lane]};
        double dq[3];
        quat_diff(q, qm, dq);
        for (int i = 0; i < 3; ++i) L.D[(9 + i) * kCols + lane] = dq[i];
    }
    __syncthreads();
    if (lane < 192) {   // (whole waves: the pair sum crosses lanes) 78 distinct entries x 2 lanes
        const int e = min(lane >> 1, 77), ij = L.tri12[e], i = ij >> 8, j = ij & 0xFF;
        double v = weighted_dot_pair(L.D + i * kCols, L.D + j * kCols, kPredCols, wc0, w.wi, lane & 1);
        if (j < 9) v += L.Q[i * 10 + j];   // (i <= j: the linear rows carry the process noise)
        if (lane < 156) L.cov[(lane & 1) ? j * 12 + i : i * 12 + j] = v;
    }
    if (lane >= 192 && lane < 205) L.mean[lane - 192] = L.ymean[lane - 192];
    __syncthreads();
    TICK(L, 5);
}

// ---- correction of (L.mean, L.cov) [decomposition already in L.S] -> out ------------------------
// returns status: 0 corrected, 1 no measurement, 2 singular Py
__device__ int ukf_correct(UkfLds& L, int type, const UtTable& ut, PoseBelief* out)
{
    const int lane = threadIdx.x;
    if (type == ROFT_MEAS_NONE) {
        for (int i = lane; i < 144; i += kUkfThreads) out->cov[i] = L.cov[i];
        if (lane < 13) out->mean[lane] = L.mean[lane];
        return 1;
    }
    const bool has_vel = (type == ROFT_MEAS_VELOCITY || type == ROFT_MEAS_POSE_VELOCITY);
    const bool has_pose = (type == ROFT_MEAS_POSE || type == ROFT_MEAS_POSE_VELOCITY);
    const int r = (has_vel ? 6 : 0) + (has_pose ? 6 : 0);
    const int m = r;
    const int mtot = (has_vel ? 6 : 0) + (has_pose ? 7 : 0);
    const int nlin = mtot - (has_pose ? 4 : 0);
    const int n = 12 + r;
    const UtW& w = ut_lookup(ut, n);
    const double sc = w.sc;
    // Additive noise in closed form (round 5, see ukf_predict): the noise of the velocity rows and of the position rows is added
    // to the measurement function's OUTPUT (CartesianQuaternionMeasurement.cpp:381-415: `+ n`), so their 2 x n_add sigma columns
    // are h(mean) +- sqrt(c) sigma_k e_k -- column 0 that many more times in the weighted sums, and R_kk on the diagonal of Py.
    // Only the rotation noise of a pose measurement acts through q [+] n (:369-379) and keeps its six columns.  Columns: the
    // mean and the 24 state columns (0 .. 24), then -- pose measurements -- +x +y +z -x -y -z of the rotation noise (25 .. 30):
    // 25 or 31 instead of 37 or 49.  The input deviations of the noise columns are zero: they never enter Pxy.
    const int n_add = r - (has_pose ? 3 : 0);
    const int ncols = 25 + (has_pose ? 6 : 0);
    const double wm0 = w.wm0 + 2.0 * n_add * w.wi, wc0 = w.wc0 + 2.0 * n_add * w.wi;

    // measurement vector in measurement order (velocity first)
    auto meas_at = [&](int k) -> double {
        return L.meas[has_vel ? k : 6 + k];
    };
    if (lane < ncols) {
        double d[12], dn[12];
        sigma_perturbation(min(lane, 24), 0, sc, false, 0.0, L, d, dn);   // (r = 0: columns 1 .. 12 add, 13 .. 24 subtract a state column)
        // rotation noise of the pose measurement (R_q sits at L.par[21 ..]): this column's rotation vector
        double rv[3] = {0.0, 0.0, 0.0};
        if (lane >= 25) {
            const int k = (lane - 25) % 3;
            const double amp = (lane < 28) ? sc : -sc;
            const double v = amp * sqrt(fabs(L.par[21 + k]));
#pragma unroll
            for (int i = 0; i < 3; ++i) { if (i == k) rv[i] = v; }
#pragma unroll
            for (int i = 0; i < 12; ++i) d[i] = 0.0;
        }
#ifdef ROFT_UKF_PROFILE
        if (lane == 0) { long long _t = clock64(); L.dbg[24] += _t - L.t0; L.t0 = _t; }
#endif
        double v[3], wv[3], x[3], q[4];
        for (int i = 0; i < 3; ++i) {
            v[i] = L.mean[i] + d[i];
            wv[i] = L.mean[3 + i] + d[3 + i];
            x[i] = L.mean[6 + i] + d[6 + i];
        }
        quat_boxplus(L.mean + 9, d + 9, q);
        // input deviations of the state dof rows: bfl recomputes them from the sigma points, (mean + d) - mean and
        // log(exp(d) q q^-1) -- which is d again, to the last bit or two, as long as the rotation offset is shorter than pi;
        // beyond that the shortest-arc logarithm wraps and d does not: recompute it the way bfl does
        for (int i = 0; i < 9; ++i) L.X[i * kCols + lane] = d[i];
        if (d[9] * d[9] + d[10] * d[10] + d[11] * d[11] < 9.0) {   // |d_rot| < 3 < pi
            for (int i = 9; i < 12; ++i) L.X[i * kCols + lane] = d[i];
        } else {
            double dq[3];
            quat_diff(q, L.mean + 9, dq);
            for (int i = 0; i < 3; ++i) L.X[(9 + i) * kCols + lane] = dq[i];
        }

        int row = 0;
        if (has_vel) {
            const double p[3] = {-x[0], -x[1], -x[2]};
            const double cr[3] = {wv[1] * p[2] - wv[2] * p[1], wv[2] * p[0] - wv[0] * p[2], wv[0] * p[1] - wv[1] * p[0]};
            for (int i = 0; i < 3; ++i) {
                L.Y[(row + i) * kCols + lane] = v[i] + cr[i];
                L.Y[(row + 3 + i) * kCols + lane] = wv[i];
            }
            row += 6;
        }
        if (has_pose) {
            for (int i = 0; i < 3; ++i) L.Y[(row + i) * kCols + lane] = x[i];
            double qo[4];
            quat_boxplus(q, rv, qo);
            for (int i = 0; i < 4; ++i) L.Y[(row + 3 + i) * kCols + lane] = qo[i];
        }
    }
    __syncthreads();

    TICK(L, 8);
    // means: the linear rows on wave 1 while wave 0 works on the quaternion rows
    if (lane >= 64 && lane < 192) linear_means8(L.Y, nlin, ncols, wm0, w.wi, lane - 64, L.ymean);   // waves 1 and 2
    double qm[4] = {1.0, 0.0, 0.0, 0.0};
    if (has_pose) {
        quaternion_mean(L.Y, nlin, ncols, wm0, w.wi, qm, L);   // ends with a workgroup barrier
        if (lane == 64)
            for (int i = 0; i < 4; ++i) L.ymean[nlin + i] = qm[i];
    } else {
        __syncthreads();
    }
    TICK(L, 23);
    if (lane < ncols) {
        for (int i = 0; i < nlin; ++i) L.D[i * kCols + lane] = L.Y[i * kCols + lane] - L.ymean[i];
        if (has_pose) {
            const double q[4] = {L.Y[nlin * kCols + lane], L.Y[(nlin + 1) * kCols + lane],
                                 L.Y[(nlin + 2) * kCols + lane], L.Y[(nlin + 3) * kCols + lane]};
            double dq[3];
            quat_diff(q, qm, dq);
            for (int i = 0; i < 3; ++i) L.D[(nlin + i) * kCols + lane] = dq[i];
        }
    }
    // innovation (idle threads of wave 3)
    if (lane >= 224 && lane < 224 + nlin) L.innov[lane - 224] = -(L.ymean[lane - 224] - meas_at(lane - 224));
    if (has_pose && lane == 255) {
        const double mq[4] = {meas_at(nlin), meas_at(nlin + 1), meas_at(nlin + 2), meas_at(nlin + 3)};
        double dq[3];
        quat_diff(mq, qm, dq);
        for (int i = 0; i < 3; ++i) L.innov[nlin + i] = dq[i];
    }
    __syncthreads();
    TICK(L, 9);
    // Pxy (12 x m) on the first 12 m threads, the upper triangle of the symmetric Py (m x m) on the next
    // m (m + 1) / 2: at most 144 + 78 threads, one pass.  The additive rows get their noise variance on the diagonal of Py
    // (R_v R_w R_x R_q sit at L.par[12 ..] in measurement order; without a velocity measurement the order starts at R_x).
    const int r_base = has_vel ? 12 : 18;
    if (m == 6) {   // 72 entries of Pxy + 21 distinct entries of Py, two lanes each (186 lanes, whole waves call)
        if (lane < 192) {
            const int e = min(lane >> 1, 92);
            const double *ar, *br;
            int o0, o1 = -1;
            double add = 0.0;
            if (e < 72) { ar = L.X + (e / 6) * kCols; br = L.D + (e % 6) * kCols; o0 = e; }
            else {
                const int ij = L.tri6[e - 72], i = ij >> 8, j = ij & 0xFF;
                ar = L.D + i * kCols; br = L.D + j * kCols; o0 = i * 6 + j; o1 = j * 6 + i;
                if (i == j && i < n_add) add = fabs(L.par[r_base + i]);   // |R|: what the sigma columns sqrt(|R|) of rounds 1 - 4 (and the rotation columns above) contribute
            }
            const double v = weighted_dot_pair(ar, br, ncols, wc0, w.wi, lane & 1) + add;
            if (lane < 186) {
                if (e < 72) { if (!(lane & 1)) L.Pxy[o0] = v; }
                else L.Py[(lane & 1) ? o1 : o0] = v;
            }
        }
    } else {
        const int nxy = 12 * m, ntri = m * (m + 1) / 2;
        if (lane < nxy) {
            const int i = lane / m, j = lane % m;
            L.Pxy[lane] = weighted_dot(L.X + i * kCols, L.D + j * kCols, ncols, wc0, w.wi);
        } else if (lane - nxy < ntri) {
            int u = lane - nxy, i = 0;
            while (u >= m - i) { u -= m - i; ++i; }
            const int j = i + u;
            double sum = weighted_dot(L.D + i * kCols, L.D + j * kCols, ncols, wc0, w.wi);
            if (i == j && i < n_add) sum += fabs(L.par[r_base + i]);
            L.Py[i * m + j] = sum;
            L.Py[j * m + i] = sum;
        }
    }
    __syncthreads();

    TICK(L, 10);
    // K = Pxy Py^-1 through the Cholesky factor of the SPD innovation covariance (the reference inverts
    // Py by LU, UKFCorrection.cpp:118; same K up to rounding), then KPy = K Py
    if (!((m == 6) ? cholesky_rows<6>(L.Py, L.aug, L.rc, L) : cholesky_rows<12>(L.Py, L.aug, L.rc, L))) {
        for (int i = lane; i < 144; i += kUkfThreads) out->cov[i] = L.cov[i];
        if (lane < 13) out->mean[lane] = L.mean[lane];
        return 2;
    }
    TICK(L, 11);
    if (m == 6) chol_solve_rows<6>(L.aug, L.rc, L.Pxy, L.K);
    else chol_solve_rows<12>(L.aug, L.rc, L.Pxy, L.K);
    __syncthreads();
    TICK(L, 25);
    if (lane < 12 * m) {
        const int i = lane / m, j = lane % m;
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += L.K[i * m + k] * L.Py[k * m + j];
        L.KPy[i * m + j] = s;
    } else if (lane >= 192 && lane < 192 + 12) {   // K * innovation on wave 3
        const int i = lane - 192;
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += L.K[i * m + k] * L.innov[k];
        L.Kin[i] = s;
    }
    __syncthreads();
    TICK(L, 26);
    if (lane < 144) {
        const int i = lane / 12, j = lane % 12;
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += L.KPy[i * m + k] * L.K[j * m + k];
        out->cov[lane] = L.cov[lane] - s;
    } else if (lane >= 192 && lane < 192 + 9) {
        out->mean[lane - 192] = L.mean[lane - 192] + L.Kin[lane - 192];
    } else if (lane == 192 + 9) {
        double qo[4];
        quat_boxplus(L.mean + 9, L.Kin + 9, qo);
        for (int i = 0; i < 4; ++i) out->mean[9 + i] = qo[i];
    }
    __syncthreads();
    TICK(L, 12);
    return 0;
}

// One launch = one StepDesc per object: optional prediction, then 0, 1 or 2 corrections of it.
// Returns false when the step was NOT applied: the twist it had to wait for did not appear within two seconds (the beliefs are
// left as they were; the caller abandons the rest of the batch and the host reports ROFT_ERR_DEVICE at its next synchronisation).
__device__ bool ukf_one_step(const EngineArrays& a, const FrameCtrl& c, int obj, int step, const UtTable& ut, UkfLds& L)
{
    ObjState& st = a.state[obj];
    // (field by field into registers: a copy of the struct, whose arrays the correction loop indexes, would live in scratch memory
    //  -- a store and a dependent load through the vector memory path at the head of every step)
    struct {
        int op, src, do_predict, n_corr, type0, type1, dst0, dst1, twist_slot;
    } sd;
    {
        const StepDesc& d = c.steps[step];
        sd.op = d.op; sd.src = d.src; sd.do_predict = d.do_predict; sd.n_corr = d.n_corr;
        sd.type0 = d.type[0]; sd.type1 = d.type[1]; sd.dst0 = d.dst[0]; sd.dst1 = d.dst[1]; sd.twist_slot = d.twist_slot;
    }
    if (!sd.op) return true;
    const ObjParams& prm = a.params[obj];
    const int lane = threadIdx.x;
    const int lin = c.lane, cur = c.cur_slot;

#ifdef ROFT_UKF_PROFILE
    if (lane < 32) L.dbg[lane] = 0;
    if (lane == 0) L.t0 = clock64();
    __syncthreads();
#endif
    const PoseBelief& src = st.belief[sd.src];
    for (int i = lane; i < 144; i += kUkfThreads) L.cov[i] = src.cov[i];
    if (lane < 13) L.mean[lane] = src.mean[lane];
    static_assert(offsetof(ObjParams, R_q) == 21 * sizeof(double), "L.par mirrors the head of ObjParams");
    if (lane >= 160 && lane < 184) L.par[lane - 160] = reinterpret_cast<const double*>(&prm)[lane - 160];
    if (lane >= 192) {   // wave 3
        const int i = lane - 192;
        if (a.handoff) {
            // The velocity filter of this batch may still be running (frame-granular hand-over): wait for the tag of the twist
            // -- frame index + 1 of the latest frame <= this one whose twist lives in the slot -- and read the six values with
            // agent-coherent loads (they were written through; nothing of them may come from this XCD's L2).
            const int want = c.frame_idx - ((c.frame_idx - sd.twist_slot) & (kTwistRing - 1)) + 1;
            if (lane == 192) {
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                int timed_out = 0;
                while (__hip_atomic_load(&st.twist_tag[sd.twist_slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 4095u) == 0u && wall_clock64() - t0 > 200000000ll) {   // two seconds (100 MHz): give up
                        if (a.dev_error) __hip_atomic_store(a.dev_error, ROFT_DEV_ERROR_TWIST_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        timed_out = 1;
                        break;
                    }
                }
                L.twist_timeout = timed_out;
            }
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            if (i < 6) L.meas[i] = __longlong_as_double((long long)__hip_atomic_load(
                reinterpret_cast<const unsigned long long*>(&st.twist_hist[sd.twist_slot][i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        } else if (i < 6) {
            L.meas[i] = st.twist_hist[sd.twist_slot][i];
        }
        if (i >= 6 && i < 13) L.meas[i] = (i < 9) ? c.pose_x[i - 6] : c.pose_q[i - 9];
    }
    __syncthreads();
    if (a.handoff && L.twist_timeout) return false;   // (uniform: nothing of the step is applied to a belief)

    TICK(L, 0);
    if (sd.do_predict) {
        ukf_predict(L, prm, c.dt, ut, st.warm_V[cur][0], &st.warm_age[cur][0], a.ukf_chol_guard);
        PoseBelief& pr = st.belief[B_PRED + lin];
        for (int i = lane; i < 144; i += kUkfThreads) pr.cov[i] = L.cov[i];
        if (lane < 13) pr.mean[lane] = L.mean[lane];
    }
    if (sd.n_corr == 0) {
        PoseBelief& d = st.belief[sd.dst0];
        for (int i = lane; i < 144; i += kUkfThreads) d.cov[i] = L.cov[i];
        if (lane < 13) d.mean[lane] = L.mean[lane];
        if (roft_object_output* row = log_row(a, c, obj))
            if (lane < 13 && sd.dst0 == cur) row->pose[lane] = L.mean[lane];
        return true;
    }
    TICK(L, 6);
    // square root of the predicted covariance, shared by both corrections of an outlier-rejection step
    if (!cholesky_state_sqrt(L, true, c.dt, a.ukf_chol_guard, a.ukf_chol_guard_bil))
        decompose_state_cov(L, st.warm_V[cur][1], &st.warm_age[cur][1]);
    TICK(L, 7);
    int status = 0;
    for (int k = 0; k < sd.n_corr; ++k) {
        const int rc = ukf_correct(L, k == 0 ? sd.type0 : sd.type1, ut, &st.belief[k == 0 ? sd.dst0 : sd.dst1]);
        status |= rc << (4 * k);
        __syncthreads();
    }
    if (lane == 0) st.lane[lin].ukf_status = status;
    // output log: the corrected belief after this frame's last step is what ROFTFilter logs
    if (roft_object_output* row = log_row(a, c, obj)) {
        __syncthreads();
        if (lane < 13) row->pose[lane] = st.belief[cur].mean[lane];
    }
#ifdef ROFT_UKF_PROFILE
    __syncthreads();
    if (lane < 32) st.dbg[lane] = L.dbg[lane];
#endif
    return true;
}

// Pose chain of a batch, one lane (BeliefSlot) per launch: one workgroup per object runs the UKF steps of the frames
// that belong to lane `lin`, frame after frame.  A step followed by the depth-render outlier test ends
// the segment -- the host enqueues launch_outlier and another segment behind it, which resumes at the lane's cursor
// (PoseLane::pc_frame / pc_step) -- otherwise the segment runs to the end of the batch.
#ifndef PRIO_UKF
#define PRIO_UKF 3
#endif
#ifdef UKF_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(UKF_WAVES_PER_EU, UKF_WAVES_PER_EU)))
#endif
__global__ __launch_bounds__(kUkfThreads) void ukf_chain_kernel(EngineArrays a, UtTable ut, int first_segment, int lin)
{
    // static LDS on purpose: with `extern __shared__` the compiler re-reads the dynamic-LDS base address from a
    // table in global memory inside every Jacobi round (two dependent global loads per round)
    ROFT_RESIDENT(a, RK_UKF_CHAIN);
    __shared__ UkfLds L;
    __shared__ unsigned s_mine;
    // A pose step is one long chain of dependent instructions on four waves; the CU it runs on is shared with the wide
    // kernels of the other chains (mask walks, rasteriser, flow measurement), whose waves compete for the issue slots of
    // the same SIMDs.  Highest wave priority: the chain's next instruction goes first whenever it is ready.
    __builtin_amdgcn_s_setprio(PRIO_UKF);
#ifdef ROFT_UKF_WALL
    const long long w_entry = wall_clock64();   // (kernel entry: dbg[26] sums entry -> first step, dbg[27] the first steps, dbg[19] counts them)
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) {
        s_ticket = atomicAdd(reinterpret_cast<unsigned*>(&a.state[0].dbg[30 + 0]) + lin, 1u);
        long long* rec = a.state[(s_ticket / a.n_obj) % a.n_obj].dbg + lin * 8;
        const long long now = wall_clock64();
        atomicMax(reinterpret_cast<unsigned long long*>(&rec[0]), (unsigned long long)((1ll << 62) - now));
        atomicMax(reinterpret_cast<unsigned long long*>(&rec[1]), (unsigned long long)now);
    }
    __syncthreads();
#endif
    const int obj = blockIdx.x;
    ObjState& st = a.state[obj];
    PoseLane& pl = st.lane[lin];
    int t = first_segment ? 0 : pl.pc_frame, step = first_segment ? 0 : pl.pc_step;
    if (t >= a.T) return;   // (this lane's chain of the batch ended in an earlier segment)
    // frames of the batch that belong to this lane (one load per frame, all in flight together)
    if (threadIdx.x == 0) s_mine = 0u;
    __syncthreads();
    if ((int)threadIdx.x < a.T && frame_ctrl(a, threadIdx.x, obj).lane == lin) atomicOr(&s_mine, 1u << threadIdx.x);
    __syncthreads();
    const unsigned mine = s_mine;
    if ((mine >> t) == 0u) {   // nothing (left) to do for this lane in this batch
        if (threadIdx.x == 0) { pl.pending_frame = -1; pl.pc_frame = a.T; pl.pc_step = 0; }
        return;
    }
    if (!first_segment && pl.pending_frame >= 0) {
        // The outlier test between the segments (outlier_fused_kernel) left the likelihood of both alternatives:
        // pick_best_alternative's decision (ROFTFilter.cpp:581-583) and the chosen belief -> p_corr_belief_ (:670-675)
        const FrameCtrl& pc = frame_ctrl(a, pl.pending_frame, obj);
        // likelihood of an alternative = mean |depth - render| over its samples: the bands' partial sums in band order;
        // no sample at all -> DBL_MAX (ROFTFilter.cpp:569-574); the gain is a bool in the reference, i.e. 1 (ROFTFilter.h:64)
        double Lk[2];
        for (int k = 0; k < 2; ++k) {
            long long hi = 0, lo = 0;
            double n2 = 0.0;
            for (int p = 0; p < pl.n_parts[k]; ++p) { hi += pl.part_hi[k][p]; lo += pl.part_lo[k][p]; n2 += pl.part_cnt[k][p]; }
            const double e = LikelihoodSum::value(hi, lo);   // the exact sum of the float terms, rounded here once
            Lk[k] = (n2 == 0.0) ? 1.7976931348623157e308 : (e / n2) / 1.0;
            if (threadIdx.x == 0) { pl.outlier_L[k] = Lk[k]; pl.outlier_cnt[k] = n2; }
        }
        const double L0 = Lk[0], L1 = Lk[1];
        const int sel = (L0 > 2.0 * L1) ? 1 : 0;
        const PoseBelief& src = st.belief[b_alt(lin, sel)];
        PoseBelief& dst = st.belief[pc.cur_slot];
        for (int i = threadIdx.x; i < 144; i += kUkfThreads) dst.cov[i] = src.cov[i];
        if (threadIdx.x < 13) dst.mean[threadIdx.x] = src.mean[threadIdx.x];
        roft_object_output* row = log_row(a, pc, obj);
        if (row && threadIdx.x < 13) row->pose[threadIdx.x] = src.mean[threadIdx.x];
        if (threadIdx.x == 0) {
            pl.outlier_selected = sel;
            if (row) { row->outlier_selected = sel; row->outlier_L[0] = L0; row->outlier_L[1] = L1; }
        }
        __syncthreads();   // the next step of this workgroup reads the chosen belief
    }
    jacobi12_table(L);
    for (int i = threadIdx.x; i < 100; i += kUkfThreads) L.Q[i] = 0.0;   // see ukf_predict: only the entries of Q(T) are rewritten per step
    __syncthreads();
    bool pending = false;
    __shared__ FrameCtrl s_c;
    int staged = -1;
#ifdef ROFT_UKF_WALL   // wall time inside the step loop and steps walked, summed per object (tools/ukf_wall.py)
    const long long w_t0 = wall_clock64();
    int w_steps = 0;
    // per launch of this lane (ticket / n_obj; launches of a lane are serialised): first / last workgroup start, last
    // end, most and total steps -> dbg[lane * 8 ..] of object (launch % n_obj)
    long long* w_rec = a.state[(s_ticket / a.n_obj) % a.n_obj].dbg + lin * 8;
#endif
    while (t < a.T) {
        if (!((mine >> t) & 1u)) { ++t; step = 0; continue; }
        if (staged != t) {   // this frame's control block -> LDS
            __syncthreads();     // (nobody still reads the previous frame's)
            stage_ctrl(&s_c, frame_ctrl(a, t, obj));
            staged = t;
            __syncthreads();
        }
        const FrameCtrl& c = s_c;
        if (step == 0 && threadIdx.x == 0) {
            pl.outlier_selected = -1;  // set again by the decision behind an outlier test
            if (roft_object_output* row0 = log_row(a, c, obj)) row0->outlier_selected = -1;
        }
        if (step >= c.n_steps) { ++t; step = 0; continue; }
#ifdef ROFT_UKF_WALL
        const long long w_s0 = wall_clock64();
#endif
        if (c.steps[step].op && !ukf_one_step(a, c, obj, step, ut, L)) {
            // a twist never arrived (ROFT_DEV_ERROR_TWIST_WAIT is raised): no further step of this batch is applied and none waits
            // another two seconds; the host refuses the results at its next synchronisation
            __syncthreads();
            t = a.T;
            step = 0;
            pending = false;
            break;
        }
#ifdef ROFT_UKF_WALL
        if (threadIdx.x == 0 && w_steps == 0) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&st.dbg[26]), (unsigned long long)(w_s0 - w_entry));
            atomicAdd(reinterpret_cast<unsigned long long*>(&st.dbg[27]), (unsigned long long)(wall_clock64() - w_s0));
            atomicAdd(reinterpret_cast<unsigned long long*>(&st.dbg[19]), 1ull);
        }
        if (threadIdx.x == 0 && w_steps > 0) {   // histogram of step durations (not the first step of a launch: cold)
            const long long d = wall_clock64() - w_s0;
            const int bin = d < 1800 ? 0 : (d < 2200 ? 1 : (d < 3000 ? 2 : 3));
            atomicAdd(reinterpret_cast<unsigned long long*>(&st.dbg[20 + bin]), 1ull);
            if (bin == 3) atomicAdd(reinterpret_cast<unsigned long long*>(&st.dbg[24]), (unsigned long long)d);
            if (bin == 3) atomicAdd(reinterpret_cast<unsigned long long*>(&st.dbg[25]), (unsigned long long)(c.steps[step].n_corr * 100 + c.steps[step].type[0]));
        }
        ++w_steps;
#endif
        __syncthreads();   // beliefs written by this step are read by the next one (same workgroup)
        pending = (step == c.outlier_step);
        ++step;
        if (pending) break;
    }
#ifdef ROFT_UKF_WALL
    if (threadIdx.x == 0) {
        const long long w_t1 = wall_clock64();
        atomicAdd(reinterpret_cast<unsigned long long*>(&st.dbg[28]), (unsigned long long)(w_t1 - w_t0));
        atomicAdd(reinterpret_cast<unsigned long long*>(&st.dbg[29]), (unsigned long long)w_steps);
        atomicMax(reinterpret_cast<unsigned long long*>(&w_rec[2]), (unsigned long long)w_t1);
        atomicMax(reinterpret_cast<unsigned long long*>(&w_rec[3]), (unsigned long long)w_steps);
        atomicAdd(reinterpret_cast<unsigned long long*>(&w_rec[4]), (unsigned long long)w_steps);
        atomicAdd(reinterpret_cast<unsigned long long*>(&w_rec[5]), 1ull);
        if (w_steps > 0) atomicMax(reinterpret_cast<unsigned long long*>(&w_rec[6]), (unsigned long long)((w_t1 - w_t0) / w_steps));
        if (w_steps > 0) atomicMax(reinterpret_cast<unsigned long long*>(&w_rec[7]), (unsigned long long)(w_t1 - w_t0));
    }
#endif
    if (threadIdx.x == 0) {
        pl.pending_frame = pending ? t : -1;
        pl.pc_frame = t;     // == a.T when the chain of this batch is complete
        pl.pc_step = step;
    }
}

void launch_ukf_chain(const EngineArrays& a, roft_ut_params ut, bool first_segment, int lin, hipStream_t s, hipEvent_t stop)
{
    UtTable tab;
    for (int k = 0; k < 3; ++k) tab.w[k] = ut_weights(18 + 3 * k, ut);
    hipExtLaunchKernelGGL(ukf_chain_kernel, dim3(a.n_obj), dim3(kUkfThreads), 0, s, nullptr, stop, 0, a, tab,
                          first_segment ? 1 : 0, lin);
}

}  // namespace roft

/*
 * ro_velocity.c -- CPU oracle (test infrastructure) for the velocity stage.
 *
 * Follows, line by line:
 *   ImageOpticalFlowMeasurement<T>::freeze   include/ROFT/ImageOpticalFlowMeasurement.hpp:231-283
 *   OpticalFlowUtils::is_flow_valid          include/ROFT/OpticalFlowUtilities.h:19-22
 *   bfl::KFPrediction + SpatialVelocityModel src/roft-lib/src/SpatialVelocityModel.cpp:15-27
 *   SKFCorrection::correctStep               src/roft-lib/src/SKFCorrection.cpp:37-153
 */
#include "roft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

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

static inline int is_flow_valid(float fx, float fy)
{
    return !isnan(fx) && !isnan(fy) && fabs(fx) < 1e9 && fabs(fy) < 1e9;
}

int ro_flow_measurement(const ro_camera* cam, const uint8_t* prev_mask, const float* prev_depth,
                        const ro_flow* flow, double dt, float radius, double depth_max,
                        int capacity, int32_t* uv, double* y, double* H)
{
    const int W = cam->width, Hh = cam->height;
    int n = 0;
    /* cv::findNonZero order is row-major (v outer, u inner); the reference walks that list with
     * `for (size_t i = 0; i < total; i += segmentation_radius_)` where the radius is a float
     * (hpp:99,237): the index is advanced in float arithmetic. */
    size_t rank = 0, next = 0;
    for (int v = 0; v < Hh; v++) {
        for (int u = 0; u < W; u++) {
            if (prev_mask[(size_t)v * W + u] == 0) continue;
            if (rank == next) {
                next = (size_t)((float)next + radius);
                float z = prev_depth[(size_t)v * W + u];
                float dx, dy;
                flow_at(flow, v / flow->grid, u / flow->grid, &dx, &dy);
                if (is_flow_valid(dx, dy) && z > 0 && z < depth_max) {
                    if (n >= capacity) return -1;
                    uv[2 * n] = u;
                    uv[2 * n + 1] = v;
                    y[2 * n] = dx;
                    y[2 * n + 1] = dy;
                    double uu = (u - cam->cx);
                    double vv = (v - cam->cy);
                    double* h = H + (size_t)12 * n;
                    h[0] = cam->fx / z;
                    h[1] = 0.0;
                    h[2] = -uu / z;
                    h[3] = -uu * vv / cam->fy;
                    h[4] = cam->fx + uu * uu / cam->fx;
                    h[5] = -vv * cam->fx / cam->fy;
                    h[6] = 0.0;
                    h[7] = cam->fy / z;
                    h[8] = -vv / z;
                    h[9] = -(cam->fy + vv * vv / cam->fy);
                    h[10] = vv * uu / cam->fx;
                    h[11] = uu * cam->fy / cam->fx;
                    for (int k = 0; k < 12; k++) h[k] *= dt;
                    n++;
                }
            }
            rank++;
        }
    }
    return n;
}

void ro_kf_predict(const double x[6], const double P[36], const double Qdiag[6], double xo[6],
                   double Po[36])
{
    memcpy(xo, x, sizeof(double) * 6);
    memcpy(Po, P, sizeof(double) * 36);
    for (int i = 0; i < 6; i++) Po[i * 6 + i] += Qdiag[i];
}

static int cmp_double(const void* a, const void* b)
{
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

int ro_skf_correct(const double x_pred[6], const double P_pred[36], int N, const double* y,
                   const double* H, const double Rdiag[2], int reweight, double x_out[6],
                   double P_out[36])
{
    memcpy(x_out, x_pred, sizeof(double) * 6);
    memcpy(P_out, P_pred, sizeof(double) * 36);
    if (N <= 0) return 1; /* SKFCorrection.cpp:61-69: empty measurement -> corr = pred */

    const int M = 2 * N;
    double* innov = (double*)malloc(sizeof(double) * M);
    for (int r = 0; r < M; r++) {
        double pred = 0.0;
        for (int k = 0; k < 6; k++) pred += H[(size_t)r * 6 + k] * x_pred[k];
        innov[r] = -(pred - y[r]);
    }

    double* lik = (double*)malloc(sizeof(double) * N);
    for (int j = 0; j < N; j++) lik[j] = 1.0;

    if (reweight) {
        /* SKFCorrection.cpp:93-94: the 2N innovation vector is re-interpreted as an N x 2
         * COLUMN-MAJOR matrix, so row k pairs innov[k] with innov[N + k] (not the two
         * components of point k).  Median and scale are fitted to those norms ... */
        double* norms = (double*)malloc(sizeof(double) * N);
        for (int k = 0; k < N; k++) norms[k] = sqrt(innov[k] * innov[k] + innov[N + k] * innov[N + k]);
        qsort(norms, N, sizeof(double), cmp_double);
        double mi = norms[N / 2];
        if ((N % 2) == 0) mi = 0.5 * (norms[N / 2 - 1] + norms[N / 2]);
        double b = 0.0;
        for (int k = 0; k < N; k++) b += fabs(norms[k] - mi);
        b /= N;
        if (b > 1e-4) {
            /* ... while the likelihoods use the true per-point norm (cpp:111). */
            double mx = 0.0;
            for (int j = 0; j < N; j++) {
                double nj = sqrt(innov[2 * j] * innov[2 * j] + innov[2 * j + 1] * innov[2 * j + 1]);
                double l = 1.0 / (2 * b) * exp(-fabs(nj - mi) / b);
                if (l < 1e-6) l = 1e-6;
                lik[j] = l;
                if (j == 0 || l > mx) mx = l;
            }
            for (int j = 0; j < N; j++) lik[j] /= mx;
        }
        free(norms);
    }

    double x[6], P[36];
    memcpy(x, x_pred, sizeof(x));
    memcpy(P, P_pred, sizeof(P));

    for (int j = 0; j < N; j++) {
        const double* Hj = H + (size_t)12 * j; /* 2 x 6 */
        double R0 = Rdiag[0], R1 = Rdiag[1];
        if (reweight) { R0 /= lik[j]; R1 /= lik[j]; }
        /* PHt = P * Hj' (6 x 2) */
        double PHt[12];
        for (int i = 0; i < 6; i++)
            for (int c = 0; c < 2; c++) {
                double s = 0.0;
                for (int k = 0; k < 6; k++) s += P[i * 6 + k] * Hj[c * 6 + k];
                PHt[i * 2 + c] = s;
            }
        /* Py = Hj * P * Hj' + R_j  -- evaluated as (Hj * P) * Hj' like the Eigen expression */
        double HP[12];
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 6; c++) {
                double s = 0.0;
                for (int k = 0; k < 6; k++) s += Hj[r * 6 + k] * P[k * 6 + c];
                HP[r * 6 + c] = s;
            }
        double Py[4];
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 2; c++) {
                double s = 0.0;
                for (int k = 0; k < 6; k++) s += HP[r * 6 + k] * Hj[c * 6 + k];
                Py[r * 2 + c] = s;
            }
        Py[0] += R0;
        Py[3] += R1;
        double Pyi[4];
        ro_inverse(2, Py, Pyi);
        double K[12]; /* 6 x 2 */
        for (int i = 0; i < 6; i++)
            for (int c = 0; c < 2; c++)
                K[i * 2 + c] = PHt[i * 2 + 0] * Pyi[0 * 2 + c] + PHt[i * 2 + 1] * Pyi[1 * 2 + c];
        double e[2];
        for (int r = 0; r < 2; r++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += Hj[r * 6 + k] * x[k];
            e[r] = y[2 * j + r] - s;
        }
        for (int i = 0; i < 6; i++) x[i] += K[i * 2] * e[0] + K[i * 2 + 1] * e[1];
        /* P = (I - K Hj) P */
        double IKH[36];
        for (int i = 0; i < 6; i++)
            for (int c = 0; c < 6; c++)
                IKH[i * 6 + c] = ((i == c) ? 1.0 : 0.0) - (K[i * 2] * Hj[c] + K[i * 2 + 1] * Hj[6 + c]);
        double Pn[36];
        for (int i = 0; i < 6; i++)
            for (int c = 0; c < 6; c++) {
                double s = 0.0;
                for (int k = 0; k < 6; k++) s += IKH[i * 6 + k] * P[k * 6 + c];
                Pn[i * 6 + c] = s;
            }
        memcpy(P, Pn, sizeof(P));
    }
    memcpy(x_out, x, sizeof(x));
    memcpy(P_out, P, sizeof(P));
    free(innov);
    free(lik);
    return 0;
}

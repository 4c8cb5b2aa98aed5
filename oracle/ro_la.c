/*
 * ro_la.c -- tiny dense algebra + quaternion helpers for the CPU oracle (test infrastructure).
 *
 * Stands in for the Eigen / bfl::utils calls made on the hot path:
 *   - JacobiSVD of a symmetric PSD covariance (bfl sigma_point(), called from
 *     src/roft-lib/src/UKFCorrection.cpp:88 and bfl::UKFPrediction) -> ro_jacobi_eig
 *   - MatrixXd::inverse() (SKFCorrection.cpp:140, UKFCorrection.cpp:118) -> ro_inverse
 *   - bfl::utils::sum_quaternion_rotation_vector (UKFCorrection.cpp:128,
 *     CartesianQuaternionMeasurement.cpp:378)                                -> ro_quat_boxplus
 *   - bfl::utils::diff_quaternion (CartesianQuaternionMeasurement.cpp:456)  -> ro_quat_diff
 *   - Eigen::AngleAxisd(Quaterniond) (ROFTFilter.cpp:389-392,520-525)       -> ro_quat_to_axis_angle
 * bfl is not vendored and not version-pinned (dockerfiles/Dockerfile:39-42): the quaternion
 * conventions below (left-multiplication, shortest-arc logarithm) are restated from its published
 * algorithm; "parity unpinned".
 */
#include "roft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void ro_jacobi_eig(int n, const double* A_in, double* w, double* V)
{
    double* A = (double*)malloc(sizeof(double) * n * n);
    memcpy(A, A_in, sizeof(double) * n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;

    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; i++) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-32 * diag || off == 0.0) break;

        for (int p = 0; p < n - 1; p++) {
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                /* classical symmetric Schur rotation */
                double tau = (aqq - app) / (2.0 * apq);
                double t = (tau >= 0.0) ? 1.0 / (tau + sqrt(1.0 + tau * tau))
                                        : -1.0 / (-tau + sqrt(1.0 + tau * tau));
                double c = 1.0 / sqrt(1.0 + t * t);
                double s = t * c;
                for (int k = 0; k < n; k++) { /* columns p, q */
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) { /* rows p, q */
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                A[p * n + q] = 0.0;
                A[q * n + p] = 0.0;
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    free(A);
}

int ro_inverse(int n, const double* A_in, double* Ainv)
{
    /* Gauss-Jordan on [A | I] with partial pivoting */
    double* M = (double*)malloc(sizeof(double) * n * 2 * n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            M[i * 2 * n + j] = A_in[i * n + j];
            M[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    }
    for (int c = 0; c < n; c++) {
        int piv = c;
        double best = fabs(M[c * 2 * n + c]);
        for (int r = c + 1; r < n; r++) {
            double v = fabs(M[r * 2 * n + c]);
            if (v > best) { best = v; piv = r; }
        }
        if (best == 0.0) { free(M); return -1; }
        if (piv != c) {
            for (int j = 0; j < 2 * n; j++) {
                double tmp = M[c * 2 * n + j];
                M[c * 2 * n + j] = M[piv * 2 * n + j];
                M[piv * 2 * n + j] = tmp;
            }
        }
        double d = 1.0 / M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; j++) M[c * 2 * n + j] *= d;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            double f = M[r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; j++) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ainv[i * n + j] = M[i * 2 * n + n + j];
    free(M);
    return 0;
}

void ro_quat_mul(const double a[4], const double b[4], double out[4])
{
    double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    out[0] = w; out[1] = x; out[2] = y; out[3] = z;
}

void ro_quat_boxplus(const double q[4], const double r[3], double out[4])
{
    double n = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double qr[4] = {1.0, 0.0, 0.0, 0.0};
    if (n > 0.0) {
        double s = sin(n / 2.0) / n;
        qr[0] = cos(n / 2.0);
        qr[1] = s * r[0];
        qr[2] = s * r[1];
        qr[3] = s * r[2];
    }
    ro_quat_mul(qr, q, out);
}

void ro_quat_diff(const double a[4], const double b[4], double out[3])
{
    double bc[4] = {b[0], -b[1], -b[2], -b[3]};
    double p[4];
    ro_quat_mul(a, bc, p);
    double n = sqrt(p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
    if (n == 0.0) { out[0] = out[1] = out[2] = 0.0; return; }
    /* shortest-arc logarithm: q and -q are the same rotation */
    double angle = 2.0 * atan2(n, fabs(p[0]));
    double sgn = (p[0] < 0.0) ? -1.0 : 1.0;
    double k = sgn * angle / n;
    out[0] = k * p[1];
    out[1] = k * p[2];
    out[2] = k * p[3];
}

void ro_quat_to_axis_angle(const double q[4], double axis[3], double* angle)
{
    double n = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n != 0.0) {
        *angle = 2.0 * atan2(n, fabs(q[0]));
        if (q[0] < 0.0) n = -n;
        axis[0] = q[1] / n; axis[1] = q[2] / n; axis[2] = q[3] / n;
    } else {
        *angle = 0.0;
        axis[0] = 1.0; axis[1] = 0.0; axis[2] = 0.0;
    }
}

void ro_axis_angle_to_quat(const double axis[3], double angle, double q[4])
{
    double s = sin(angle / 2.0);
    q[0] = cos(angle / 2.0);
    q[1] = s * axis[0]; q[2] = s * axis[1]; q[3] = s * axis[2];
}

void ro_quat_to_rotmat(const double q[4], double R[9])
{
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z); R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z); R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y); R[7] = 2.0 * (y * z + w * x); R[8] = 1.0 - 2.0 * (x * x + y * y);
}

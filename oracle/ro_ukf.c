/*
 * ro_ukf.c -- CPU oracle (test infrastructure) for the pose UKF.
 *
 * ROFT side (in /root/reference):
 *   CartesianQuaternionModel::motion / Q(T)   src/roft-lib/src/CartesianQuaternionModel.cpp:86-141
 *   CartesianQuaternionMeasurement::predictedMeasure / innovation
 *                                             src/roft-lib/src/CartesianQuaternionMeasurement.cpp:357-487
 *   ROFT::UKFCorrection::correctStep          src/roft-lib/src/UKFCorrection.cpp:54-133
 *
 * Third-party side: robotology/bayes-filters-lib ("bfl"), NOT in /root/reference and not version
 * pinned (cloned at HEAD by dockerfiles/Dockerfile:39-42).  Restated from its published algorithm
 * (sigma_point.cpp / utils.cpp of the quaternion-enabled bfl line used by ROFT):
 *   UTWeight(n, a, b, k):  lambda = a^2 (n + k) - n; c = n + lambda;
 *                          wm0 = lambda / c; wc0 = wm0 + 1 - a^2 + b; wm_i = wc_i = 1 / (2 c)
 *   sigma_point(state, c): A = U sqrt(S) from the Jacobi SVD of the (augmented) covariance;
 *                          perturbations [0, +sqrt(c) A, -sqrt(c) A]; linear rows: mean + d;
 *                          quaternion rows: exp(d_rot) (x) q_mean; noise rows: d
 *   unscented_transform:   propagate; linear mean = sum wm y; quaternion mean = dominant
 *                          eigenvector of sum wm q q'; deviations (linear difference /
 *                          diff_quaternion); Py = D diag(wc) D'; Pxy = X diag(wc) D' with X the
 *                          input deviations of the state dof rows (noise rows excluded)
 *   UKFPrediction (generic StateModel): augment with getNoiseCovarianceMatrix(), UT through motion()
 * Call sites that anchor these semantics: UKFCorrection.cpp:81-88,118-132;
 * ROFTFilter.cpp:163-166 (UKFPrediction), 64-67 (Gaussian(9,1,true)).
 * Free choices documented here because they cannot be checked against bfl ("parity unpinned"):
 *   - the sign of the quaternion-mean eigenvector is fixed so that it has a non-negative dot
 *     product with the propagated central sigma point;
 *   - diff_quaternion uses the shortest-arc logarithm (ro_la.c).
 */
#include "roft_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXN 24          /* 12 state dof + up to 12 noise dof */
#define MAXCOLS (2 * MAXN + 1)

typedef struct {
    double c;
    double wm[MAXCOLS];
    double wc[MAXCOLS];
    int ncols;
} ut_weight;

static void ut_weights(int n, const ro_ut_params* ut, ut_weight* w)
{
    double lambda = ut->alpha * ut->alpha * (n + ut->kappa) - n;
    w->c = n + lambda;
    w->ncols = 2 * n + 1;
    w->wm[0] = lambda / (n + lambda);
    w->wc[0] = lambda / (n + lambda) + (1.0 - ut->alpha * ut->alpha + ut->beta);
    for (int i = 1; i < w->ncols; i++) w->wm[i] = w->wc[i] = 1.0 / (2.0 * (n + lambda));
}

/* Sigma points of Gaussian(9 linear, 1 quaternion) augmented with r noise dof.
 * sp: (13 + r) x ncols row-major; also returns the rotation/linear perturbations of the 12 state
 * dof rows (dX: 12 x ncols), which are what bfl recomputes as input deviations. */
static void sigma_points(const double mean[13], const double P[144], const double* Qn, int r,
                         double c, double* sp, int ncols)
{
    int n = 12 + r;
    double Pa[MAXN * MAXN];
    memset(Pa, 0, sizeof(Pa));
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) Pa[i * n + j] = P[i * 12 + j];
    for (int i = 0; i < r; i++)
        for (int j = 0; j < r; j++) Pa[(12 + i) * n + (12 + j)] = Qn[i * r + j];

    double w[MAXN], V[MAXN * MAXN];
    ro_jacobi_eig(n, Pa, w, V);
    double sc = sqrt(c);

    for (int col = 0; col < ncols; col++) {
        double d[MAXN];
        if (col == 0) {
            for (int i = 0; i < n; i++) d[i] = 0.0;
        } else {
            int k = (col - 1) % n;
            double sgn = (col <= n) ? 1.0 : -1.0;
            double s = sqrt(fabs(w[k]));
            for (int i = 0; i < n; i++) d[i] = sgn * sc * V[i * n + k] * s;
        }
        for (int i = 0; i < 9; i++) sp[i * ncols + col] = mean[i] + d[i];
        double q[4];
        ro_quat_boxplus(mean + 9, d + 9, q);
        for (int i = 0; i < 4; i++) sp[(9 + i) * ncols + col] = q[i];
        for (int i = 0; i < r; i++) sp[(13 + i) * ncols + col] = d[12 + i];
    }
}

/* dominant eigenvector of sum_i wm_i q_i q_i' ; q: 4 x ncols rows at stride ncols */
static void quaternion_mean(const double* q, int ncols, const double* wm, double out[4])
{
    double M[16];
    memset(M, 0, sizeof(M));
    for (int c = 0; c < ncols; c++)
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) M[i * 4 + j] += wm[c] * q[i * ncols + c] * q[j * ncols + c];
    double w[4], V[16];
    ro_jacobi_eig(4, M, w, V);
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (w[i] > w[best]) best = i;
    double dot = 0.0;
    for (int i = 0; i < 4; i++) {
        out[i] = V[i * 4 + best];
        dot += out[i] * q[i * ncols + 0];
    }
    if (dot < 0.0)
        for (int i = 0; i < 4; i++) out[i] = -out[i];
}

/* Conformance kit (tests/ref_kit): the unscented-transform weights and the sigma set the oracle draws, so that somebody who
 * has bfl can hold bfl::UTWeight / bfl::sigma_point against them (recalled, UNVERIFIED). */
void ro_ut_weights(int n, const ro_ut_params* ut, double out[4])
{
    ut_weight w;
    ut_weights(n, ut, &w);
    out[0] = w.c; out[1] = w.wm[0]; out[2] = w.wc[0]; out[3] = w.wm[1];
}

int ro_sigma_points(const double mean[13], const double P[144], const double* Qn, int r, const ro_ut_params* ut,
                    double* sp /* (13 + r) x (2 (12 + r) + 1), row-major */)
{
    if (r < 0 || 12 + r > MAXN) return -1;
    ut_weight w;
    ut_weights(12 + r, ut, &w);
    sigma_points(mean, P, Qn, r, w.c, sp, w.ncols);
    return w.ncols;
}

void ro_pose_process_noise(const double psd[3], const double sig_w[3], double T, double Q[81])
{
    memset(Q, 0, sizeof(double) * 81);
    for (int i = 0; i < 3; i++) {
        Q[i * 9 + i] = psd[i] * T;
        Q[(3 + i) * 9 + (3 + i)] = sig_w[i];
        Q[(6 + i) * 9 + (6 + i)] = psd[i] * (pow(T, 3.0) / 3.0);
        Q[i * 9 + (6 + i)] = psd[i] * (pow(T, 2.0) / 2.0);
        Q[(6 + i) * 9 + i] = psd[i] * (pow(T, 2.0) / 2.0);
    }
}

/* CartesianQuaternionModel::motion for one column: in = [v w x q | n(9)] */
static void motion(const double* in, int stride, double T, double* out, int ostride)
{
    double v[3], w[3], x[3], q[4], nz[9];
    for (int i = 0; i < 3; i++) {
        v[i] = in[i * stride];
        w[i] = in[(3 + i) * stride];
        x[i] = in[(6 + i) * stride];
    }
    for (int i = 0; i < 4; i++) q[i] = in[(9 + i) * stride];
    for (int i = 0; i < 9; i++) nz[i] = in[(13 + i) * stride];

    for (int i = 0; i < 3; i++) {
        out[i * ostride] = v[i] + nz[i];
        out[(3 + i) * ostride] = w[i] + nz[3 + i];
        /* position: x + n_x, then += v * T with v WITHOUT noise (cpp:94-97) */
        out[(6 + i) * ostride] = (x[i] + nz[6 + i]) + v[i] * T;
    }
    /* quaternion: w without noise (cpp:103) */
    double norm_w = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) + DBL_EPSILON;
    double c = cos(norm_w * T / 2.0);
    double s = sin(norm_w * T / 2.0) / norm_w;
    /* (c I + s Omega(w)) q, Omega = left product matrix of (0, w) */
    double qo[4];
    qo[0] = c * q[0] + s * (-w[0] * q[1] - w[1] * q[2] - w[2] * q[3]);
    qo[1] = c * q[1] + s * (w[0] * q[0] - w[2] * q[2] + w[1] * q[3]);
    qo[2] = c * q[2] + s * (w[1] * q[0] + w[2] * q[1] - w[0] * q[3]);
    qo[3] = c * q[3] + s * (w[2] * q[0] - w[1] * q[1] + w[0] * q[2]);
    for (int i = 0; i < 4; i++) out[(9 + i) * ostride] = qo[i];
}

/* mean (9 lin + quat) and 12 x ncols deviations of 13 x ncols propagated points */
static void pose_mean_dev(const double* Y, int ncols, const ut_weight* w, double mean[13], double* D)
{
    for (int i = 0; i < 9; i++) {
        double s = 0.0;
        for (int c = 0; c < ncols; c++) s += Y[i * ncols + c] * w->wm[c];
        mean[i] = s;
    }
    quaternion_mean(Y + 9 * ncols, ncols, w->wm, mean + 9);
    for (int c = 0; c < ncols; c++) {
        for (int i = 0; i < 9; i++) D[i * ncols + c] = Y[i * ncols + c] - mean[i];
        double q[4] = {Y[9 * ncols + c], Y[10 * ncols + c], Y[11 * ncols + c], Y[12 * ncols + c]};
        double d[3];
        ro_quat_diff(q, mean + 9, d);
        for (int i = 0; i < 3; i++) D[(9 + i) * ncols + c] = d[i];
    }
}

/* C (ra x rb) = A (ra x ncols) diag(wc) B' (rb x ncols) */
static void weighted_outer(const double* A, int ra, const double* B, int rb, int ncols,
                           const double* wc, double* C)
{
    for (int i = 0; i < ra; i++)
        for (int j = 0; j < rb; j++) {
            double s = 0.0;
            for (int c = 0; c < ncols; c++) s += A[i * ncols + c] * wc[c] * B[j * ncols + c];
            C[i * rb + j] = s;
        }
}

void ro_ukf_predict(const double mean[13], const double P[144], const double Q[81], double T,
                    const ro_ut_params* ut, double mean_out[13], double P_out[144])
{
    const int r = 9, n = 21;
    ut_weight w;
    ut_weights(n, ut, &w);
    const int ncols = w.ncols;
    double sp[(13 + 12) * MAXCOLS];
    sigma_points(mean, P, Q, r, w.c, sp, ncols);

    double Y[13 * MAXCOLS];
    for (int c = 0; c < ncols; c++) motion(sp + c, ncols, T, Y + c, ncols);

    double D[12 * MAXCOLS];
    pose_mean_dev(Y, ncols, &w, mean_out, D);
    weighted_outer(D, 12, D, 12, ncols, w.wc, P_out);
}

int ro_ukf_correct(const double mean[13], const double P[144], int type, const double* meas,
                   const double* Rdiag, const ro_ut_params* ut, double mean_out[13],
                   double P_out[144])
{
    memcpy(mean_out, mean, sizeof(double) * 13);
    memcpy(P_out, P, sizeof(double) * 144);
    if (type == RO_MEAS_NONE) return 1; /* UKFCorrection.cpp:64-68 */

    const int has_vel = (type == RO_MEAS_VELOCITY || type == RO_MEAS_POSE_VELOCITY);
    const int has_pose = (type == RO_MEAS_POSE || type == RO_MEAS_POSE_VELOCITY);
    const int r = (has_vel ? 6 : 0) + (has_pose ? 6 : 0);   /* noise dof */
    const int m = r;                                         /* innovation size */
    const int mtot = (has_vel ? 6 : 0) + (has_pose ? 7 : 0); /* measurement total size */
    const int n = 12 + r;

    double Rn[12 * 12];
    memset(Rn, 0, sizeof(Rn));
    for (int i = 0; i < r; i++) Rn[i * r + i] = Rdiag[i];

    ut_weight w;
    ut_weights(n, ut, &w);
    const int ncols = w.ncols;
    double sp[(13 + 12) * MAXCOLS];
    sigma_points(mean, P, Rn, r, w.c, sp, ncols);

    /* CartesianQuaternionMeasurement::predictedMeasure */
    double Y[13 * MAXCOLS];
    for (int c = 0; c < ncols; c++) {
        const double* s = sp + c;
        const double* nz = sp + 13 * ncols + c;
        int row = 0;
        if (has_vel) {
            double v[3] = {s[0 * ncols], s[1 * ncols], s[2 * ncols]};
            double wv[3] = {s[3 * ncols], s[4 * ncols], s[5 * ncols]};
            double p[3] = {-s[6 * ncols], -s[7 * ncols], -s[8 * ncols]};
            /* v_object + w x (-p)   (cpp:410) */
            double cr[3] = {wv[1] * p[2] - wv[2] * p[1], wv[2] * p[0] - wv[0] * p[2],
                            wv[0] * p[1] - wv[1] * p[0]};
            for (int i = 0; i < 3; i++) {
                Y[(row + i) * ncols + c] = (v[i] + cr[i]) + nz[i * ncols];
                Y[(row + 3 + i) * ncols + c] = wv[i] + nz[(3 + i) * ncols];
            }
            row += 6;
        }
        if (has_pose) {
            int off = has_vel ? 6 : 0; /* noise.segment<3>(6) for PoseVelocity, head<3>() for Pose */
            for (int i = 0; i < 3; i++) Y[(row + i) * ncols + c] = s[(6 + i) * ncols] + nz[(off + i) * ncols];
            double q[4] = {s[9 * ncols], s[10 * ncols], s[11 * ncols], s[12 * ncols]};
            double rv[3] = {nz[(r - 3) * ncols], nz[(r - 2) * ncols], nz[(r - 1) * ncols]};
            double qo[4];
            ro_quat_boxplus(q, rv, qo);
            for (int i = 0; i < 4; i++) Y[(row + 3 + i) * ncols + c] = qo[i];
        }
    }

    /* predicted measurement mean + deviations (m x ncols) */
    double ymean[13];
    double D[12 * MAXCOLS];
    const int nlin = mtot - (has_pose ? 4 : 0);
    for (int i = 0; i < nlin; i++) {
        double sacc = 0.0;
        for (int c = 0; c < ncols; c++) sacc += Y[i * ncols + c] * w.wm[c];
        ymean[i] = sacc;
    }
    if (has_pose) quaternion_mean(Y + nlin * ncols, ncols, w.wm, ymean + nlin);
    for (int c = 0; c < ncols; c++) {
        for (int i = 0; i < nlin; i++) D[i * ncols + c] = Y[i * ncols + c] - ymean[i];
        if (has_pose) {
            double q[4] = {Y[nlin * ncols + c], Y[(nlin + 1) * ncols + c], Y[(nlin + 2) * ncols + c],
                           Y[(nlin + 3) * ncols + c]};
            double d[3];
            ro_quat_diff(q, ymean + nlin, d);
            for (int i = 0; i < 3; i++) D[(nlin + i) * ncols + c] = d[i];
        }
    }
    double Py[144];
    weighted_outer(D, m, D, m, ncols, w.wc, Py);

    /* input deviations of the 12 state dof rows */
    double X[12 * MAXCOLS];
    for (int c = 0; c < ncols; c++) {
        for (int i = 0; i < 9; i++) X[i * ncols + c] = sp[i * ncols + c] - mean[i];
        double q[4] = {sp[9 * ncols + c], sp[10 * ncols + c], sp[11 * ncols + c], sp[12 * ncols + c]};
        double d[3];
        ro_quat_diff(q, mean + 9, d);
        for (int i = 0; i < 3; i++) X[(9 + i) * ncols + c] = d[i];
    }
    double Pxy[144];
    weighted_outer(X, 12, D, m, ncols, w.wc, Pxy);

    /* innovation (CartesianQuaternionMeasurement.cpp:436-487) */
    double innov[12];
    for (int i = 0; i < nlin; i++) innov[i] = -(ymean[i] - meas[i]);
    if (has_pose) {
        double d[3];
        ro_quat_diff(meas + nlin, ymean + nlin, d);
        for (int i = 0; i < 3; i++) innov[nlin + i] = d[i];
    }

    double Pyi[144];
    if (ro_inverse(m, Py, Pyi) != 0) return 2;
    double K[144]; /* 12 x m */
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < m; j++) {
            double sacc = 0.0;
            for (int k = 0; k < m; k++) sacc += Pxy[i * m + k] * Pyi[k * m + j];
            K[i * m + j] = sacc;
        }
    double Kin[12];
    for (int i = 0; i < 12; i++) {
        double sacc = 0.0;
        for (int k = 0; k < m; k++) sacc += K[i * m + k] * innov[k];
        Kin[i] = sacc;
    }
    for (int i = 0; i < 9; i++) mean_out[i] = mean[i] + Kin[i];
    ro_quat_boxplus(mean + 9, Kin + 9, mean_out + 9);

    /* P - K Py K' evaluated as (K Py) K' */
    double KPy[144];
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < m; j++) {
            double sacc = 0.0;
            for (int k = 0; k < m; k++) sacc += K[i * m + k] * Py[k * m + j];
            KPy[i * m + j] = sacc;
        }
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) {
            double sacc = 0.0;
            for (int k = 0; k < m; k++) sacc += KPy[i * m + k] * K[j * m + k];
            P_out[i * 12 + j] = P[i * 12 + j] - sacc;
        }
    return 0;
}

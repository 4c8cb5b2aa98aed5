/*
 * ro_metrics.c -- CPU oracle (test infrastructure): ADD, ADD-S (= BOP "adi") and the AUC of
 * evaluation/metrics.py.
 *
 * Follows tools/third_party/bop_pose_error.py: add :73-87, adi :89-108 (nearest neighbour from
 * pts_gt into pts_est, here by brute force instead of a KD-tree), VOCap :12-27; and
 * evaluation/metrics.py:303-344 (threshold 0.1 m -> inf, sort, accuracy = cumsum(1)/n as float32,
 * VOCap * 100).  PINNED: checked against fixtures produced by importing bop_pose_error.py itself
 * (tests/golden/make_bop_fixtures.py -> tests/golden/bop_fixtures.json).
 */
#include "roft_oracle.h"

#include <math.h>
#include <stdlib.h>

static void transform(const double R[9], const double t[3], const double* p, double* o)
{
    for (int i = 0; i < 3; i++) o[i] = R[i * 3] * p[0] + R[i * 3 + 1] * p[1] + R[i * 3 + 2] * p[2] + t[i];
}

double ro_add(const double R_est[9], const double t_est[3], const double R_gt[9],
              const double t_gt[3], const double* pts, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) {
        double a[3], b[3];
        transform(R_est, t_est, pts + 3 * i, a);
        transform(R_gt, t_gt, pts + 3 * i, b);
        s += sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
    }
    return s / n;
}

double ro_adds(const double R_est[9], const double t_est[3], const double R_gt[9],
               const double t_gt[3], const double* pts, int n)
{
    double* est = (double*)malloc(sizeof(double) * 3 * n);
    for (int i = 0; i < n; i++) transform(R_est, t_est, pts + 3 * i, est + 3 * i);
    double s = 0.0;
    for (int i = 0; i < n; i++) {
        double g[3];
        transform(R_gt, t_gt, pts + 3 * i, g);
        double best = INFINITY;
        for (int j = 0; j < n; j++) {
            const double* e = est + 3 * j;
            double d = (e[0] - g[0]) * (e[0] - g[0]) + (e[1] - g[1]) * (e[1] - g[1]) + (e[2] - g[2]) * (e[2] - g[2]);
            if (d < best) best = d;
        }
        s += sqrt(best);
    }
    free(est);
    return s / n;
}

static int cmp_double(const void* a, const void* b)
{
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

double ro_auc(const double* distances, int n)
{
    if (n <= 0) return 0.0;
    double* d = (double*)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) d[i] = (distances[i] > 0.1) ? INFINITY : distances[i];
    qsort(d, n, sizeof(double), cmp_double);
    /* VOCap over the finite recalls; precision = float32(k) / n */
    int m = 0;
    while (m < n && isfinite(d[m])) m++;
    double ap = 0.0;
    if (m > 0) {
        /* mrec = [0, rec..., 0.1], mpre = [0, prec..., prec[-1]] then running max */
        double* mrec = (double*)malloc(sizeof(double) * (m + 2));
        double* mpre = (double*)malloc(sizeof(double) * (m + 2));
        mrec[0] = 0.0; mpre[0] = 0.0;
        for (int i = 0; i < m; i++) {
            mrec[i + 1] = d[i];
            mpre[i + 1] = (double)((float)(i + 1) / (float)n);
        }
        mrec[m + 1] = 0.1;
        mpre[m + 1] = mpre[m];
        for (int i = 1; i < m + 2; i++)
            if (mpre[i] < mpre[i - 1]) mpre[i] = mpre[i - 1];
        for (int i = 1; i < m + 2; i++)
            if (mrec[i] != mrec[i - 1]) ap += (mrec[i] - mrec[i - 1]) * mpre[i];
        ap *= 10.0;
        free(mrec); free(mpre);
    }
    free(d);
    return ap * 100.0;
}

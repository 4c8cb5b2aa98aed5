"""Pose-error metrics of the reference's evaluation: ADD, ADD-S (BOP "adi") and their AUC.

Definitions follow tools/third_party/bop_pose_error.py:73-108 (add / adi, nearest neighbour from the
ground-truth points into the estimated points) and evaluation/metrics.py:303-344 (threshold 0.1 m ->
inf, sort, accuracy = cumsum(1)/n, VOCap * 100).  Pinned by tests/golden/bop_fixtures.json.
"""
import numpy as np
from scipy import spatial


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def add(R_est, t_est, R_gt, t_gt, pts):
    pe = pts @ np.asarray(R_est).T + np.asarray(t_est)
    pg = pts @ np.asarray(R_gt).T + np.asarray(t_gt)
    return float(np.linalg.norm(pe - pg, axis=1).mean())


def adds(R_est, t_est, R_gt, t_gt, pts):
    pe = pts @ np.asarray(R_est).T + np.asarray(t_est)
    pg = pts @ np.asarray(R_gt).T + np.asarray(t_gt)
    d, _ = spatial.cKDTree(pe).query(pg, k=1)
    return float(d.mean())


def auc(distances, threshold=0.1):
    d = np.array(distances, dtype=np.float64)
    d[d > threshold] = np.inf
    d = np.sort(d)
    n = len(d)
    if n == 0:
        return 0.0
    acc = np.cumsum(np.ones((n,), np.float32)) / n
    fin = np.isfinite(d)
    rec, prec = d[fin], acc[fin]
    if len(rec) == 0:
        return 0.0
    mrec = np.concatenate([[0.0], rec, [threshold]])
    mpre = np.concatenate([[0.0], prec, [prec[-1]]])
    mpre = np.maximum.accumulate(mpre)
    i = np.where(mrec[1:] != mrec[:-1])[0] + 1
    return float(np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) * 10.0 * 100.0)


def trajectory_adds(pose_est, pose_ref, pts):
    """pose_*: [F, 7] (x, q wxyz).  Returns the per-frame ADD-S distances."""
    out = np.zeros(len(pose_est))
    for k in range(len(pose_est)):
        out[k] = adds(quat_to_rot(pose_est[k, 3:]), pose_est[k, :3], quat_to_rot(pose_ref[k, 3:]), pose_ref[k, :3], pts)
    return out

"""Generates tests/golden/bop_fixtures.json by IMPORTING the reference's own metric code
(/root/reference/tools/third_party/bop_pose_error.py: add, adi, VOCap) in the dev container.
Only the resulting input/output vectors are committed; the reference source does not travel.

    python tests/golden/make_bop_fixtures.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/tools/third_party")
import bop_pose_error as bop  # noqa: E402


def rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def main():
    rng = np.random.default_rng(20221)
    pts = rng.uniform(-0.1, 0.1, size=(100, 3))
    cases = []
    for i in range(20):
        R_gt, t_gt = rot(rng), rng.uniform(-0.2, 0.2, 3) + np.array([0, 0, 0.7])
        scale = [1e-3, 1e-2, 5e-2, 0.3][i % 4]
        dq = rng.normal(size=3) * scale * 3
        a = np.linalg.norm(dq)
        K = np.array([[0, -dq[2], dq[1]], [dq[2], 0, -dq[0]], [-dq[1], dq[0], 0]])
        dR = np.eye(3) + np.sin(a) / a * K + (1 - np.cos(a)) / a ** 2 * K @ K
        R_est, t_est = dR @ R_gt, t_gt + rng.normal(size=3) * scale
        cases.append(dict(R_est=R_est.tolist(), t_est=t_est.tolist(), R_gt=R_gt.tolist(), t_gt=t_gt.tolist(),
                          add=float(bop.add(R_est, t_est, R_gt, t_gt, pts)),
                          adi=float(bop.adi(R_est, t_est, R_gt, t_gt, pts))))
    # AUC as evaluation/metrics.py:327-334 computes it from a distance vector, via bop.VOCap
    aucs = []
    for n, spread in ((30, 0.02), (57, 0.08), (11, 0.3), (5, 1.0)):
        d = np.abs(rng.normal(size=n)) * spread
        dd = d.copy()
        dd[dd > 0.1] = np.inf
        sd = np.sort(dd)
        acc = np.cumsum(np.ones((n,), np.float32)) / n
        aucs.append(dict(distances=d.tolist(), auc=float(bop.VOCap(sd, acc) * 100.0)))
    out = dict(points=pts.tolist(), cases=cases, aucs=aucs,
               source="tools/third_party/bop_pose_error.py (add :73-87, adi :89-108, VOCap :12-27), "
                      "evaluation/metrics.py:327-334")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bop_fixtures.json"), "w") as f:
        json.dump(out, f)
    print("wrote", len(cases), "pose cases and", len(aucs), "AUC cases")


if __name__ == "__main__":
    main()

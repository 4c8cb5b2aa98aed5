#!/usr/bin/env python3
"""Metrics table of tracking results, the step right after the filtering path: what evaluation/evaluate.py of the
reference computes for its 'ours' entries -- per object and pooled over ALL -- printed as a Markdown table.

  evaluate_results.py --results DIR --dataset DIR [--objects NAME ...] [--points DIR] [--metrics m1,m2,...] [--of-ms 0]

DIR/<object>/{pose_estimate[_ycb].txt, velocity_estimate.txt, execution_times.txt} are the log files ROFT-tracker (or
ROFT-tracker-batch, or tools/run_sequence.py with --out DIR/<object>/) leaves; dataset/<object>/gt/poses.txt holds the ground-truth
poses (x y z axis angle) and, optionally, gt/velocities.txt the ground-truth velocities.  ADD / ADD-S use <points>/<object>/points.xyz
when --points is given (the reference's YCB_Video_Models layout), else every k-th vertex of dataset/<object>/model.obj.
As in evaluate.py: the leading six velocity columns of the pose log are dropped (data_loader.py:238-241), the estimates are compared
with as many ground-truth rows as there are estimates (from the row the run started at: --first-frame), the linear velocity is
moved from the camera origin to the object position before it is compared (v = v_O + w x r, evaluate.py:514-521), and --of-ms is
added to the execution times (the reference adds what its optical-flow source costs per frame, evaluate.py:470-484).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roft_amd import io, metrics  # noqa: E402

DEFAULT = "rmse_cartesian_3d,rmse_angular,add,adi,rmse_linear_velocity,rmse_angular_velocity,time,excess_33_ms"
UNITS = dict(rmse_cartesian_3d="cm", rmse_cartesian_x="cm", rmse_cartesian_y="cm", rmse_cartesian_z="cm", rmse_angular="deg", add="AUC %", adi="AUC %",
             rmse_linear_velocity="cm/s", rmse_angular_velocity="deg/s", max_linear_velocity="m/s", max_angular_velocity="deg/s", time="ms", excess_33_ms="frames")


def first_existing(*paths):
    for p in paths:
        if os.path.exists(p):
            return p
    return None


def load_object(results, dataset, name, first_frame, of_ms):
    d = os.path.join(results, name)
    pose_path = first_existing(os.path.join(d, "pose_estimate_ycb.txt"), os.path.join(d, "pose_estimate.txt"), os.path.join(d, "pose_estimate"))
    if pose_path is None:
        raise FileNotFoundError("no pose_estimate in " + d)
    pose = io.read_log(pose_path, skip_cols=6)
    gt_all = np.loadtxt(os.path.join(dataset, name, "gt", "poses.txt"), ndmin=2)
    gt = gt_all[first_frame:first_frame + len(pose)]
    pose = pose[:len(gt)]
    out = dict(pose=pose, gt_pose=gt)
    vel_path = first_existing(os.path.join(d, "velocity_estimate.txt"), os.path.join(d, "velocity_estimate"))
    gt_vel_path = os.path.join(dataset, name, "gt", "velocities.txt")
    if vel_path and os.path.exists(gt_vel_path):
        vel = io.read_log(vel_path)[:len(gt)]
        out["vel"] = metrics.object_velocity_from_twist(vel, gt[:, :3])
        out["gt_vel"] = np.loadtxt(gt_vel_path, ndmin=2)[first_frame:first_frame + len(vel)]
    t_path = first_existing(os.path.join(d, "execution_times.txt"), os.path.join(d, "execution_times"))
    if t_path:
        t = io.read_log(t_path)
        t[:, 0] += of_ms
        out["time"] = t
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--results", required=True)
    ap.add_argument("--dataset", required=True)
    ap.add_argument("--objects", nargs="*", default=None)
    ap.add_argument("--points", default=None)
    ap.add_argument("--metrics", default=DEFAULT)
    ap.add_argument("--first-frame", type=int, default=0)
    ap.add_argument("--of-ms", type=float, default=0.0)
    ap.add_argument("--json", default=None, help="also write the numbers to this file")
    args = ap.parse_args(argv)
    names = args.objects or sorted(n for n in os.listdir(args.results) if os.path.isdir(os.path.join(args.results, n)))
    data = {n: load_object(args.results, args.dataset, n, args.first_frame, args.of_ms) for n in names}
    points = {}
    for n in names:
        if args.points:
            points[n] = np.loadtxt(os.path.join(args.points, n, "points.xyz"), ndmin=2)
        else:
            mesh = first_existing(os.path.join(args.dataset, n, "model.obj"), os.path.join(args.dataset, n, n + ".obj"))
            if mesh:
                v, _ = io.load_obj(mesh)
                points[n] = v.astype(np.float64)[:: max(1, len(v) // 500)]
    wanted = [m for m in args.metrics.split(",") if m]
    table = {}
    for m in wanted:
        metric = metrics.Metric(m, auc_points=points)
        vel_metric = "velocity" in m
        row = {}
        have = [n for n in names if ("vel" in data[n] if vel_metric else True) and (m not in ("time", "excess_33_ms") or "time" in data[n])
                and (m not in ("add", "adi") or n in points)]
        for n in have:
            ref, sig = (data[n]["gt_vel"], data[n]["vel"]) if vel_metric else (data[n]["gt_pose"], data[n]["pose"])
            row[n] = metric.evaluate(n, ref, sig, data[n].get("time"))
        if have:
            pick = (lambda k: {n: data[n][k] for n in have})
            ref, sig = (pick("gt_vel"), pick("vel")) if vel_metric else (pick("gt_pose"), pick("pose"))
            row["ALL"] = metric.evaluate("ALL", ref, sig, {n: data[n]["time"] for n in have} if all("time" in data[n] for n in have) else None)
        table[m] = row
    cols = [m for m in wanted if table[m]]
    print("| object | " + " | ".join("%s (%s)" % (m, UNITS[m]) for m in cols) + " |")
    print("|---|" + "---|" * len(cols))
    for n in names + ["ALL"]:
        print("| %s | " % n + " | ".join(("%.3f" % table[m][n]) if n in table[m] else "-" for m in cols) + " |")
    if args.json:
        json.dump(table, open(args.json, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Runs the MI355X engine on one object of a Fast-YCB / HO-3D style sequence directory and writes the reference's log
files (`pose_estimate`, `velocity_estimate`, ROFTFilter.cpp:386-394) -- what `test/test.sh` does with `ROFT-tracker`.

  run_sequence.py --root DIR --object NAME --mesh model.obj [--flow-set nvof_1_slow] [--mask-set NAME]
                  [--pose-set dope] [--out PREFIX] [--compute-flow nvof1|nvof2] [--no-delay] [--init-pose x y z qw qx qy qz]
                  [--start-at-first-detection]
                  [--from config_fast_ycb.cfg [--group::key value ...]]

--from reads the filter parameters from one of the reference's configuration files (config/config_fast_ycb.cfg,
config/config_ho3d.cfg) and applies `--a::b::c value` overrides exactly as ROFT-tracker's ConfigParser does
(roft_amd/config.py); without it the defaults of those files are used (roft_default_config / roft_default_object).

The camera comes from DIR/cam_K.json (width, height, fx, fy, cx, cy).  --compute-flow first runs tools/flow_dumper.py
on DIR/rgb (the MI355X replacement of the NVOF dumper) into DIR/optical_flow/<flow-set>.  With DIR/gt/poses.txt present
the ADD-S / ADD AUC and the RMSE metrics of evaluation/metrics.py are printed as one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--object", required=True)
    ap.add_argument("--mesh", required=True)
    ap.add_argument("--flow-set", default="nvof_1_slow")
    ap.add_argument("--mask-set", default="mrcnn_ycbv_bop_pbr")
    ap.add_argument("--pose-set", default="dope")
    ap.add_argument("--out", default=None)
    ap.add_argument("--compute-flow", choices=["nvof1", "nvof2"], default=None)
    ap.add_argument("--no-delay", action="store_true")
    ap.add_argument("--start-at-first-detection", action="store_true",
                    help="start where test/test_ho3d.sh:68-80 starts the tracker: at the frame and with the pose "
                         "tools/dataset/dope_pose_finder/pose_finder.py reports for the 5 fps pose source")
    ap.add_argument("--init-pose", type=float, nargs=7, default=None, metavar=("X", "Y", "Z", "QW", "QX", "QY", "QZ"),
                    help="initial_condition.pose (default: the first valid detection)")
    ap.add_argument("--from", dest="cfg_file", default=None, help="ROFT configuration file (libconfig), overrides as --a::b::c value")
    args, overrides = ap.parse_known_args(argv)

    from roft_amd import _lib as L
    from roft_amd import engine as E
    from roft_amd import io, metrics

    L.require_device()
    cam = json.load(open(os.path.join(args.root, "cam_K.json")))
    W, H = int(cam["width"]), int(cam["height"])
    if args.compute_flow:
        out = os.path.join(args.root, "optical_flow", args.flow_set)
        rc = subprocess.call([sys.executable, os.path.join(ROOT, "tools", "flow_dumper.py"), args.root, "txt", "png", "1", "0",
                              str(W), str(H), args.compute_flow, out])
        if rc != 0:
            return rc
    start = 0
    if args.start_at_first_detection:
        found = io.find_initial_pose(os.path.join(args.root, args.pose_set, "poses.txt"), 5.0)
        if found is None:
            sys.stderr.write("no valid detection on the 5 fps grid\n")
            return 1
        start = found[0]
        aa = [float(v) for v in found[1].split()]
        if args.init_pose is None:
            args.init_pose = aa[:3] + list(io.axis_angle_to_quat(np.array(aa[3:6]), aa[6]))
    seq = io.Sequence(args.root, args.object, flow_set=args.flow_set, mask_set=args.mask_set, pose_set=args.pose_set,
                      width=W, height=H, delayed=not args.no_delay, first_frame=start)
    first = None
    for k in range(len(seq)):
        ok, first = io.read_flow(os.path.join(seq.flow_dir, "%d.float" % k))
        if ok:
            break
    if first is None:
        sys.stderr.write("no optical flow frames in %s\n" % seq.flow_dir)
        return 1
    ftype, grid, scale = io.flow_format(first, W)
    init_from_cfg = False
    if args.cfg_file:
        from roft_amd import config as K
        # the sequence's own camera (cam_K.json) unless the command line says otherwise, as test/test.sh passes it
        cam_over = []
        for k in ("width", "height", "fx", "fy", "cx", "cy"):
            if "--camera_dataset::" + k not in overrides:
                cam_over += ["--camera_dataset::" + k, str(cam[k])]
        init_from_cfg = any(o.startswith("--initial_condition::pose::") for o in overrides)
        cfg, d, _extras, rest = K.load(args.cfg_file, cam_over + overrides, flow_type=ftype, flow_grid=grid, flow_scale=scale)
        if rest:
            ap.error("unknown arguments: %s" % " ".join(rest))
        if (cfg.cam.width, cfg.cam.height) != (W, H):
            ap.error("camera_dataset::width / height do not match the sequence")
    else:
        if overrides:
            ap.error("unknown arguments: %s (settings need --from FILE)" % " ".join(overrides))
        cfg = E.default_config(W, H, ftype, max_objects=1)
        cfg.cam.fx, cfg.cam.fy, cfg.cam.cx, cfg.cam.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
        cfg.flow_grid, cfg.flow_scale = grid, scale
        d = E.default_object()
    eng = E.ROFTFilterBatch(cfg)
    verts, tris = io.load_obj(args.mesh)
    # initial condition: the configuration's when it was given on the command line (test/test.sh:120-123 passes the first
    # detection that way), else the first valid detection of the sequence
    if args.start_at_first_detection:
        init_from_cfg = False
    if not init_from_cfg:
        k0 = int(np.argmax(seq.pose_ok)) if seq.pose_ok.any() else 0
        init = args.init_pose if args.init_pose is not None else list(seq.poses[k0])
        for i in range(7):
            d.p_mean0[6 + i] = init[i]
    eng.add_object(d, verts, tris)
    n = len(seq) - start
    if n <= 0:
        sys.stderr.write("the first detection arrives after the last frame\n")
        return 1
    eng.enable_log(n)
    for k in range(start, len(seq)):
        eng.submit([seq.frame(k)])
        eng.step()
    pose, twist, npts, sel = eng.get_log(0, n)
    eng.close()
    prefix = args.out if args.out is not None else os.path.join(args.root, "roft_mi355x_")
    io.write_estimate_logs(prefix, pose[:, 0], twist[:, 0])
    report = dict(frames=n, first_frame=start, logs=[prefix + "pose_estimate", prefix + "velocity_estimate"], flow_type=int(ftype), flow_grid=int(grid))
    gt_path = os.path.join(args.root, "gt", "poses.txt")
    if os.path.exists(gt_path):
        gt, _ = io.read_poses(gt_path)
        est = np.concatenate([pose[:, 0, 6:9], pose[:, 0, 9:13]], 1)
        pts = verts.astype(np.float64)[:: max(1, len(verts) // 500)]
        g = gt[start + 12:start + n]
        dist = metrics.trajectory_adds(est[12:], g, pts)
        report.update(adds_mm_mean=1e3 * float(dist.mean()), adds_auc=metrics.auc(dist),
                      rmse_position_cm=metrics.rmse_cartesian_3d(g[:, :3], est[12:, :3]),
                      rmse_orientation_deg=metrics.rmse_angular(g[:, 3:], est[12:, 3:]))
    print(json.dumps(report))
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Per-frame time of the reference's executable (tests/cpp/_ref_build/ROFT-tracker: its own main.cpp over this engine, built by
__graft_entry__.build() where the reference checkout is) on a synthetic Fast-YCB-shaped sequence directory, 1280x720 CV_16SC2
grid 4 (config_fast_ycb.cfg's shape): microseconds per frame outside data loading (ROFT_FILTER_TIMING=1) with the images read in
place from the pinned pool (the default) and staged through the HOST upload path (ROFT_FACADE_STAGED=1), and the mean of the
integer-millisecond `execution_times` the reference logs.   python tools/tracker_timing.py [frames] [--out file.json]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from roft_amd import config as K
from roft_amd import io, synth
import util

REF_BIN = os.path.join(ROOT, "tests", "cpp", "_ref_build", "ROFT-tracker")
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
if not os.path.exists(REF_BIN):
    raise SystemExit("tests/cpp/_ref_build/ROFT-tracker is built where the reference checkout is (python __graft_entry__.py)")
import copy
st = copy.copy(util.stream(704, n, 1, shape="B", flow_type=synth.FLOW_S16C2, mesh_n=24, with_gray=True, device="cuda"))
st.pose_meas = st.pose_meas.copy()
st.pose_meas[0] = st.pose_meas[6]
tmp = tempfile.mkdtemp(prefix="roft_trk_")
root = os.path.join(tmp, "seq")
mesh = io.write_sequence(root, st, "box", flow_set="analytic")
c = st.camera
cfg_path = os.path.join(tmp, "config.cfg")
open(cfg_path, "w").write(K.tracker_text(c.width, c.height, 1.0, 1.0, 0.0, 0.0))
m0 = synth.initial_pose_from_stream(st)
axis, angle = io.quat_to_axis_angle(m0[9:13])
res = {}
for tag, extra in (("in_place", {}), ("staged", {"ROFT_FACADE_STAGED": "1"})):
    out_dir = os.path.join(tmp, "out_" + tag)
    os.makedirs(out_dir)
    args = ["--from", cfg_path,
            "--camera_dataset::fx", repr(c.fx), "--camera_dataset::fy", repr(c.fy), "--camera_dataset::cx", repr(c.cx), "--camera_dataset::cy", repr(c.cy),
            "--camera_dataset::path", root,
            "--initial_condition::pose::x", ",".join("%.17g" % v for v in m0[6:9]),
            "--initial_condition::pose::axis_angle", ",".join("%.17g" % v for v in list(axis) + [angle]),
            "--kinematic_model::pose::sigma_angular", "1.0,1.0,1.0", "--log::path", out_dir,
            "--measurement_model::pose::cov_q", "0.0001,0.0001,0.0001",
            "--measurement_model::use_pose", "true", "--measurement_model::use_pose_resync", "true", "--measurement_model::use_velocity", "true",
            "--model::name", "box", "--model::use_internal_db", "false", "--model::external_path", mesh,
            "--optical_flow_dataset::path", root, "--optical_flow_dataset::set", "analytic/", "--outlier_rejection::enable", "true",
            "--pose_dataset::path", os.path.join(root, "dope", "poses.txt"),
            "--segmentation_dataset::flow_aided", "true", "--segmentation_dataset::path", root, "--segmentation_dataset::set", "gt"]
    r = subprocess.run([REF_BIN] + args, capture_output=True, text=True, timeout=600, env=dict(os.environ, ROFT_FILTER_TIMING="1", **extra))
    if r.returncode != 0:
        raise SystemExit(r.stdout[-1500:] + r.stderr[-1500:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("ROFTFilter:")][-1]
    times = io.read_log(os.path.join(out_dir, "execution_times.txt"))
    res[tag] = dict(report=line, us_per_frame=float(line.split("frames,")[1].split("us")[0]),
                    engine_us_per_frame=float(line.split("read-back:")[1].split("us")[0]),
                    execution_times_ms_mean=float(times[:, 0].mean()), frames=int(times.shape[0]),
                    pose=io.read_log(os.path.join(out_dir, "pose_estimate.txt")).tolist())
same = res["in_place"]["pose"] == res["staged"]["pose"]
for v in res.values():
    del v["pose"]
out = dict(what="tests/cpp/_ref_build/ROFT-tracker (the reference's main.cpp over this engine), 1280x720 CV_16SC2 grid 4, one object, %d frames" % n,
           in_place=res["in_place"], staged=res["staged"], identical_pose_logs=same)
print(json.dumps(out, indent=1))
if "--out" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--out") + 1], "w"), indent=1)

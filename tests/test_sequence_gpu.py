"""End to end over files (SURVEY section 8f rows 1-3 together): a synthetic stream written as a Fast-YCB style directory
-> tools/flow_dumper.py (HIP flow producer, the NVOF dumper's counterpart) -> tools/run_sequence.py (engine fed through
roft_amd.io.Sequence) -> the reference's log files -> ADD-S / RMSE."""
import importlib.util
import json
import os

import numpy as np
import pytest

from roft_amd import io

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_sequence_on_disk_with_stream_flow_equals_in_memory_engine(tmp_path, capsys):
    from test_engine_gpu import make_engine
    import copy
    n = 40
    st = copy.copy(util.stream(700, n, 2, with_gray=True))
    # a pose file holds ONE detection per source frame; the generator draws the frame-0 detection twice (delivered at
    # frame 0 and, delayed, at frame 6) with independent noise -- make them the same detection
    assert st.pose_valid[0] and st.pose_valid[6]
    st.pose_meas = st.pose_meas.copy()
    st.pose_meas[0] = st.pose_meas[6]
    root = str(tmp_path / "seq")
    mesh = io.write_sequence(root, st, "box", flow_set="analytic")
    rs = load_tool("run_sequence")
    from roft_amd import synth
    m0 = synth.initial_pose_from_stream(st)
    assert rs.main(["--root", root, "--object", "box", "--mesh", mesh, "--flow-set", "analytic", "--mask-set", "gt",
                    "--out", str(tmp_path / "a_"), "--init-pose"] + ["%.17g" % v for v in m0[6:13]]) == 0
    rep = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rep["frames"] == n and rep["adds_auc"] > 80.0 and rep["flow_type"] == 13
    est = np.loadtxt(str(tmp_path / "a_pose_estimate"))
    vel = np.loadtxt(str(tmp_path / "a_velocity_estimate"))
    assert est.shape == (n, 13) and vel.shape == (n, 6)
    # the same stream straight from memory
    eng = make_engine([st])
    eng.enable_log(n)
    for k in range(n):
        depth, flow, mask, pose = util.frame_inputs(st, k)
        eng.submit([dict(depth=depth, flow=flow, mask=mask, pose=pose, dt=st.dt)])
        eng.step()
    pose_log, twist_log, _, _ = eng.get_log(0, n)
    eng.close()
    # (the poses file stores axis-angle and the logs 12 significant digits: equal up to that rounding)
    assert np.abs(est[:, 6:9] - pose_log[:, 0, 6:9]).max() < 1e-9
    assert np.abs(vel - twist_log[:, 0]).max() < 1e-9
    # the same run configured the way test/test.sh configures ROFT-tracker: a configuration file in the reference's
    # format plus `--group::key value` overrides for the initial condition and one filter switch
    from roft_amd import config as K
    c = st.camera
    cfg_path = str(tmp_path / "config.cfg")
    open(cfg_path, "w").write(K.default_text(c.width, c.height, 1.0, 1.0, 0.0, 0.0))   # (camera: cam_K.json of the sequence wins)
    axis, angle = io.quat_to_axis_angle(m0[9:13])
    common = ["--root", root, "--object", "box", "--mesh", mesh, "--flow-set", "analytic", "--mask-set", "gt", "--from", cfg_path,
              "--initial_condition::pose::x", ", ".join("%.17g" % v for v in m0[6:9]),
              "--initial_condition::pose::axis_angle", ", ".join("%.17g" % v for v in list(axis) + [angle])]
    assert rs.main(common + ["--out", str(tmp_path / "b_")]) == 0
    capsys.readouterr()
    est_b = np.loadtxt(str(tmp_path / "b_pose_estimate"))
    assert np.abs(est_b - est).max() < 1e-9          # (the axis-angle round trip of the initial orientation)
    assert rs.main(common + ["--out", str(tmp_path / "c_"), "--measurement_model::use_pose_resync", "false"]) == 0
    capsys.readouterr()
    est_c = np.loadtxt(str(tmp_path / "c_pose_estimate"))
    assert np.abs(est_c - est).max() > 1e-6          # the switch reached the filter


def test_sequence_on_disk_with_produced_flow(tmp_path, capsys):
    n = 40
    st = util.stream(701, n, 2, with_gray=True)
    root = str(tmp_path / "seq")
    mesh = io.write_sequence(root, st, "box")
    rs = load_tool("run_sequence")
    assert rs.main(["--root", root, "--object", "box", "--mesh", mesh, "--flow-set", "lk_2", "--mask-set", "gt",
                    "--compute-flow", "nvof2"]) == 0
    rep = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert sorted(os.listdir(os.path.join(root, "optical_flow", "lk_2")))[0] == "1.float"
    assert len(os.listdir(os.path.join(root, "optical_flow", "lk_2"))) == n - 1
    assert rep["adds_auc"] > 75.0 and rep["rmse_position_cm"] < 3.0
    # started where test/test_ho3d.sh starts the tracker: the first detection on the 5 fps grid, six frames (the delay of
    # the source) after the frame it was computed on
    pp = os.path.join(root, "dope", "poses.txt")
    rows = open(pp).read().splitlines()
    rows[0] = "0.0 0.0 0.0 0.0 0.0 0.0 0.0"
    open(pp, "w").write("\n".join(rows) + "\n")
    assert rs.main(["--root", root, "--object", "box", "--mesh", mesh, "--flow-set", "lk_2", "--mask-set", "gt",
                    "--start-at-first-detection", "--out", str(tmp_path / "d_")]) == 0
    rep = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rep["first_frame"] == 12 and rep["frames"] == n - 12
    assert np.loadtxt(str(tmp_path / "d_pose_estimate")).shape == (n - 12, 13)
    assert rep["rmse_position_cm"] < 5.0

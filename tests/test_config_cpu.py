"""Configuration files of the reference -> roft_config / roft_object_desc (roft_amd/config.py): the libconfig subset, the
`--a::b::c value` overrides of ConfigParser, and every key ROFTFilter's constructor consumes (src/roft/src/main.cpp:43-147,
:286-325).  The reference's own config/*.cfg are read where the reference checkout is present (the build container)."""
import ctypes as C
import glob
import math
import os

import pytest

from roft_amd import _lib as L
from roft_amd import config as K

REF_CFG = sorted(glob.glob("/root/reference/config/*.cfg"))

SMALL = """
# a file in the reference's format, written for this test
sample_time = 0.04;
camera_dataset: { width = 640; height = 480; fx = 614.0; fy = 615.5; cx = 320.0; cy = 240.5; path = "?"; }  // trailing comment
initial_condition:
{
    pose: { v = [0.0, 0.0, 0.0]; w = [0.0, 0.0, 0.1]; x = [0.1, -0.2, 0.7]; axis_angle = [0.0, 0.0, 1.0, 1.5707963267948966];
            cov_v = [0.001, 0.001, 0.001]; cov_w = [0.002, 0.002, 0.002]; cov_x = [0.003, 0.003, 0.003]; cov_q = [0.004, 0.004, 0.004]; }
    velocity: { v = [0.0, 0.0, 0.0]; w = [0.0, 0.0, 0.0]; cov_v = [0.01, 0.01, 0.01]; cov_w = [0.02, 0.02, 0.02]; }
}
kinematic_model: { pose: { sigma_linear = [2.0, 2.0, 2.0]; sigma_angular = [3.0, 3.0, 3.0]; }
                   velocity: { sigma_linear = [0.1, 0.1, 0.1]; sigma_angular = [0.2, 0.2, 0.2]; } }
measurement_model:
{
    pose: { cov_v = [0.1, 0.1, 0.1]; cov_w = [0.0001, 0.0001, 0.0001]; cov_x = [0.001, 0.001, 0.001]; cov_q = [0.0002, 0.0002, 0.0002]; }
    velocity: { cov_flow = [1.0, 2.0]; depth_maximum = 1.5; subsampling_radius = 20.0; weight_flow = false; }
    use_pose = true; use_pose_resync = false; use_velocity = true;
}
outlier_rejection: { enable = true; gain = 0.01; }
pose_dataset: { path = "/data/poses.txt"; fps_reduction = true; delay = true; original_fps = 30.0; desired_fps = 10.0; }
segmentation_dataset: { path = "?"; set = "mrcnn"; fps_reduction = false; delay = false; original_fps = 30.0; desired_fps = 5.0; flow_aided = true; }
unscented_transform: { alpha = 1.0; beta = 2.0; kappa = 0.5; }
"""


def test_small_file_every_field():
    cfg = K.parse_cfg(SMALL)
    c, o, extras = K.to_engine(cfg, L.FLOW_S16C2, flow_grid=4, flow_scale=32.0)
    assert (c.cam.width, c.cam.height, c.cam.fx, c.cam.fy, c.cam.cx, c.cam.cy) == (640, 480, 614.0, 615.5, 320.0, 240.5)
    assert (c.flow_type, c.flow_grid, c.flow_scale) == (L.FLOW_S16C2, 4, 32.0)
    assert c.sample_time == 0.04 and (c.ut.alpha, c.ut.beta, c.ut.kappa) == (1.0, 2.0, 0.5)
    assert (c.depth_maximum, c.subsampling_radius, c.flow_weighting) == (1.5, 20.0, 0)
    assert (c.use_pose, c.use_pose_resync, c.use_velocity, c.outlier_rejection, c.flow_aided_segmentation) == (1, 0, 1, 1, 1)
    assert c.pose_frames_between == 3          # int(30 / 10), DatasetTransformDelayed
    assert c.mask_frames_between == -1         # neither delayed nor reduced: the plain data-set source, rate unknown
    s = math.sqrt(0.5)
    assert list(o.p_mean0)[:9] == [0, 0, 0, 0, 0, 0.1, 0.1, -0.2, 0.7]
    assert max(abs(a - b) for a, b in zip(list(o.p_mean0)[9:], [s, 0.0, 0.0, s])) < 1e-15
    assert list(o.p_cov0_diag) == [0.001] * 3 + [0.002] * 3 + [0.003] * 3 + [0.004] * 3
    assert list(o.v_cov0_diag) == [0.01] * 3 + [0.02] * 3 and list(o.v_q_diag) == [0.1] * 3 + [0.2] * 3
    # sigma_linear is the PSD of the linear acceleration, sigma_angular the variance of the angular velocity (main.cpp:78-79)
    assert list(o.p_psd_lin_acc) == [2.0] * 3 and list(o.p_sigma_ang_vel) == [3.0] * 3
    assert list(o.p_meas_cov_q) == [0.0002] * 3 and list(o.v_meas_cov_flow) == [1.0, 2.0]
    assert extras["pose_dataset.path"] == "/data/poses.txt" and extras["outlier_rejection.gain"] == 0.01


def test_command_line_overrides():
    cfg = K.parse_cfg(SMALL)
    rest = K.apply_overrides(cfg, ["--from", "x.cfg", "--measurement_model::use_pose_resync", "true", "--initial_condition::pose::x",
                                   "0.5, 0.25, 1.0", "--camera_dataset::width", "1280", "--pose_dataset::path", "/p", "--unknown", "7",
                                   "--sample_time", "0.1"])
    assert rest == ["--unknown", "7"]
    c, o, extras = K.to_engine(cfg, L.FLOW_F32C2)
    assert c.use_pose_resync == 1 and c.cam.width == 1280 and c.sample_time == 0.1
    assert list(o.p_mean0)[6:9] == [0.5, 0.25, 1.0] and extras["pose_dataset.path"] == "/p"
    with pytest.raises(ValueError):
        K.apply_overrides(cfg, ["--initial_condition::pose::x", "1.0, 2.0"])
    with pytest.raises(KeyError):
        K.lookup(cfg, "measurement_model::nothing")


@pytest.mark.skipif(not REF_CFG, reason="the reference checkout is not here")
@pytest.mark.parametrize("path", REF_CFG)
def test_reference_config_files(path):
    cfg = K.parse_cfg(open(path).read())
    keys = set(K.all_keys(cfg))
    # every setting of the file is one main.cpp reads, and every setting the filter consumes is in the file
    assert keys <= set(K.FILTER_KEYS) | set(K.DATASET_KEYS), sorted(keys - set(K.FILTER_KEYS) - set(K.DATASET_KEYS))
    assert set(K.FILTER_KEYS) <= keys, sorted(set(K.FILTER_KEYS) - keys)
    c, o, extras = K.to_engine(cfg, L.FLOW_F32C2)
    # the filter parameters of both files are the defaults of the ABI (roft_default_config / roft_default_object)
    d = L.Config()
    L.check(L.lib().roft_default_config(C.byref(d), c.cam.width, c.cam.height, L.FLOW_F32C2))
    for f in ("depth_maximum", "subsampling_radius", "flow_weighting", "use_pose", "use_pose_resync", "use_velocity",
              "outlier_rejection", "flow_aided_segmentation", "mask_frames_between", "pose_frames_between"):
        assert getattr(c, f) == getattr(d, f), f
    assert abs(c.sample_time - d.sample_time) < 1e-9 and (c.ut.alpha, c.ut.beta, c.ut.kappa) == (d.ut.alpha, d.ut.beta, d.ut.kappa)
    do = L.ObjectDesc()
    L.check(L.lib().roft_default_object(C.byref(do)))
    for f in ("p_cov0_diag", "v_mean0", "v_cov0_diag", "p_sigma_ang_vel", "p_psd_lin_acc", "v_q_diag", "p_meas_cov_v", "p_meas_cov_w",
              "p_meas_cov_x", "p_meas_cov_q", "v_meas_cov_flow"):
        assert list(getattr(o, f)) == list(getattr(do, f)), f
    assert list(o.p_mean0) == [0.0] * 9 + [1.0, 0.0, 0.0, 0.0]       # axis_angle = [1, 0, 0, 0]
    if path.endswith("config_fast_ycb.cfg"):
        assert (c.cam.width, c.cam.height) == (1280, 720) and abs(c.cam.fx - 1229.4285612615463) < 1e-12
    assert extras["model.internal_db_name"] == "DOPE" or "model.internal_db_name" in extras


def test_default_text_round_trips_to_the_abi_defaults():
    text = K.default_text(640, 480, 614.7, 614.7, 320.0, 240.0)
    c, o, _ = K.to_engine(K.parse_cfg(text), L.FLOW_F32C2)
    d, do = L.Config(), L.ObjectDesc()
    L.check(L.lib().roft_default_config(C.byref(d), 640, 480, L.FLOW_F32C2))
    L.check(L.lib().roft_default_object(C.byref(do)))
    for f, _t in L.Config._fields_:
        if f not in ("cam", "ut", "max_objects", "device", "max_batch_frames"):
            assert getattr(c, f) == getattr(d, f), f
    for f, _t in L.ObjectDesc._fields_:
        if f != "mesh":
            assert list(getattr(o, f)) == list(getattr(do, f)), f

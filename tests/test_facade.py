"""C++ facade (include/ROFT/): every class of the reference's filter API -- SpatialVelocityModel, CartesianQuaternionModel,
ImageOpticalFlowMeasurement<T>, CartesianQuaternionMeasurement, SKFCorrection, UKFCorrection,
ImageSegmentationOFAidedSource<T>, ROFTFilter with the constructor of ROFTFilter.h:42-73 -- compiles against the C ABI,
fails loudly without a device, and on the GPU returns bit for bit what the operator-level ABI / the engine return."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "roft_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(tmp_path, name):
    from roft_amd import _lib
    _lib.build()
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L", CSRC, "-lroft_hip", "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def problem(tmp_path):
    rng = np.random.default_rng(9)
    n = 40
    x = rng.normal(size=6) * 0.05
    P = np.eye(6) * 1e-3
    q = np.full(6, 0.1)
    H = rng.normal(size=(2 * n, 6)) * 20
    y = H @ (x + rng.normal(size=6) * 0.02) + rng.normal(size=2 * n)
    pm = np.zeros(13)
    pm[6:9] = [0.0, 0.1, 0.7]
    pm[9] = 1.0
    pm[3:6] = [0.2, -0.1, 0.3]
    pP = np.eye(12) * 1e-3
    # measurement of the pose filter: twist (v_O, w) and pose (x, q) near the state
    qm = np.array([0.999, 0.02, -0.03, 0.01])
    meas = np.concatenate([[0.01, -0.02, 0.03, 0.19, -0.12, 0.31], [0.004, 0.103, 0.702], qm / np.linalg.norm(qm)])
    path = str(tmp_path / "in.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("i", n))
        for a in (x, P, q, y, H, pm, pP, meas):
            f.write(np.ascontiguousarray(a, np.float64).tobytes())
    return path, (x, P, q, y, H, pm, pP, meas)


def dump_stream(path, st, n):
    import util
    from roft_amd import synth
    verts, tris = st.mesh
    c = st.camera
    with open(path, "wb") as f:
        f.write(struct.pack("5i", c.width, c.height, n, len(verts), len(tris)))
        f.write(struct.pack("4d", c.fx, c.fy, c.cx, c.cy))
        f.write(np.asarray(synth.initial_pose_from_stream(st), np.float64).tobytes())
        f.write(np.ascontiguousarray(verts, np.float32).tobytes())
        f.write(np.ascontiguousarray(tris, np.int32).tobytes())
        for k in range(n):
            depth, flow, mask, pose = util.frame_inputs(st, k)
            f.write(struct.pack("d3i", st.dt, flow is not None, mask is not None, pose is not None))
            f.write(np.ascontiguousarray(depth, np.float32).tobytes())
            if flow is not None:
                f.write(np.ascontiguousarray(flow, np.float32).tobytes())
            if mask is not None:
                f.write(np.ascontiguousarray(mask, np.uint8).tobytes())
            if pose is not None:
                f.write(np.concatenate([pose[0], pose[1]]).astype(np.float64).tobytes())


def test_facade_classes_compile_and_fail_loudly_without_device(tmp_path):
    import util
    from roft_amd import _lib
    exe = build(tmp_path, "facade_check")
    if _lib.lib().roft_device_count() > 0:
        pytest.skip("a HIP device is present")
    inp, _ = problem(tmp_path)
    st = util.stream(33, 3, 4)
    dump_stream(str(tmp_path / "s.bin"), st, 3)
    r = subprocess.run([exe, inp, str(tmp_path / "out.bin"), str(tmp_path / "s.bin")], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout


def test_facade_headers_name_every_reference_class():
    """One header per reference header: the class names, the constructor arity and the virtuals a maintainer's
    code would use are there."""
    inc = os.path.join(ROOT, "include", "ROFT")
    want = {
        "SKFCorrection.h": ["class SKFCorrection : public bfl::GaussianCorrection", "std::unique_ptr<bfl::LinearMeasurementModel> measurement_model",
                            "void correctStep(const bfl::GaussianMixture& pred_state, bfl::GaussianMixture& corr_state) override"],
        "UKFCorrection.h": ["class UKFCorrection : public bfl::GaussianCorrection", "std::unique_ptr<bfl::MeasurementModel> meas_model",
                            "void correctStep(const bfl::GaussianMixture& pred_state, bfl::GaussianMixture& corr_state) override"],
        "ImageOpticalFlowMeasurement.hpp": ["class ImageOpticalFlowMeasurement : public bfl::LinearMeasurementModel", "bool freeze(const bfl::Data& data",
                                            "predictedMeasure(", "innovation(", "getMeasurementMatrix() const override",
                                            "getNoiseCovarianceMatrix() const override", "getMeasurementDescription() const override",
                                            "bool setProperty(const std::string& property) override"],
        "CartesianQuaternionMeasurement.h": ["class CartesianQuaternionMeasurement : public bfl::MeasurementModel",
                                             "enum class MeasurementMode { Standard, RepeatOnlyVelocity, PopBufferedMeasurement }"],
        "CartesianQuaternionModel.h": ["class CartesianQuaternionModel : public bfl::StateModel", "bool setSamplingTime(const double& sample_time) override"],
        "SpatialVelocityModel.h": ["class SpatialVelocityModel : public bfl::LinearStateModel", "getStateTransitionMatrix() override"],
        "ImageSegmentationOFAidedSource.hpp": ["class ImageSegmentationOFAidedSource : public RobotsIO::Utils::Segmentation", "bool step_frame() override"],
        "ROFTFilter.h": ["class ROFTFilter : public bfl::FilteringAlgorithm", "const ModelParameters& model_parameters",
                         "const bool pose_outlier_rejection_gain", "void filtering_step() override"],
    }
    for name, needles in want.items():
        text = " ".join(open(os.path.join(inc, name)).read().split())
        for needle in needles:
            assert needle in text, (name, needle)


@pytest.mark.gpu
def test_facade_classes_match_operator_abi(tmp_path):
    import util
    from roft_amd import _lib as L
    from roft_amd import ops
    exe = build(tmp_path, "facade_check")
    inp, (x, P, q, y, H, pm, pP, meas) = problem(tmp_path)
    n_frames = 8
    st = util.stream(35, n_frames, 2)
    dump_stream(str(tmp_path / "s.bin"), st, n_frames)
    out = str(tmp_path / "out.bin")
    subprocess.check_call([exe, inp, out, str(tmp_path / "s.bin")])
    raw = open(out, "rb").read()
    pos = [0]

    def take(count, dtype=np.float64):
        a = np.frombuffer(raw, dtype, count, pos[0])
        pos[0] += a.nbytes
        return a

    # velocity filter
    xp, Pp = ops.kf_predict(x, P, q)
    _, xc, Pc = ops.skf_correct(xp, Pp, y, H, (1.0, 1.0), True)
    assert np.array_equal(take(6), xc) and np.array_equal(take(36).reshape(6, 6), Pc)
    # pose filter
    Q = ops.process_noise([1.0] * 3, [1.0] * 3, 1.0 / 30.0)
    m1, P1 = ops.ukf_predict(pm, pP, Q, 1.0 / 30.0)
    assert np.array_equal(take(13), m1) and np.array_equal(take(144).reshape(12, 12), P1)
    r_vel, r_pose = [0.1] * 3 + [1e-4] * 3, [1e-3] * 3 + [1e-4] * 3
    _, m2, P2 = ops.ukf_correct(m1, P1, L.MEAS_POSE_VELOCITY, meas, r_vel + r_pose)
    assert np.array_equal(take(13), m2) and np.array_equal(take(144).reshape(12, 12), P2)
    _, m3, P3 = ops.ukf_correct(m1, P1, L.MEAS_VELOCITY, meas[:6], r_vel)
    assert np.array_equal(take(13), m3) and np.array_equal(take(144).reshape(12, 12), P3)
    # host-side h(x) / innovation of the measurement model at the mean column: velocity v + w x (-x), pose difference
    innov = take(12)
    v, w, xx = m1[0:3], m1[3:6], m1[6:9]
    want = np.concatenate([meas[0:3] - (v + np.cross(w, -xx)), meas[3:6] - w, meas[6:9] - xx])
    assert np.abs(innov[:9] - want).max() < 1e-15
    from roft_amd import synth
    dq = synth.quat_mul(meas[9:13], m1[9:13] * np.array([1.0, -1.0, -1.0, -1.0]))
    rot = 2.0 * np.arctan2(np.linalg.norm(dq[1:]), abs(dq[0])) * dq[1:] / np.linalg.norm(dq[1:]) * np.sign(dq[0])
    assert np.abs(innov[9:] - rot).max() < 1e-14
    # flow measurement model: frame 1 against mask and depth of frame 0
    depth0, _, mask0, _ = util.frame_inputs(st, 0)
    _, flow1, _, _ = util.frame_inputs(st, 1)
    c = st.camera
    n_ref, uv, yy, HH = ops.flow_measurement(L.Camera(c.width, c.height, c.fx, c.fy, c.cx, c.cy),
                                             np.where(mask0 > 1, 255, 0).astype(np.uint8), depth0, flow1, st.dt)
    n = int(take(1)[0])
    assert n == n_ref and n >= 3
    assert np.array_equal(take(2 * n), yy.ravel()) and np.array_equal(take(12 * n).reshape(2 * n, 6), HH)
    # flow-aided segmentation source: frames 0 .. 7 with the new mask of frame 6
    m = None
    for k in range(n_frames):
        _, flow, mask, _ = util.frame_inputs(st, k)
        if k == 0:
            m = mask.copy()
        elif mask is not None:
            m = ops.mask_propagate(mask, [util.frame_inputs(st, j)[1] for j in range(1, k + 1)], 6)
        else:
            m = m.copy()
            m[0, 0] = 0
            m = ops.mask_propagate(m, [flow], 6)
    assert st.mask_delivery[6] >= 0
    got = take(st.camera.width * st.camera.height, np.uint8).reshape(st.camera.height, st.camera.width)
    assert np.array_equal(got, m)
    assert pos[0] == len(raw)


def test_whole_filter_facade_compiles_and_fails_loudly_without_device(tmp_path):
    import util
    from roft_amd import _lib
    exe = build(tmp_path, "filter_check")
    if _lib.lib().roft_device_count() > 0:
        pytest.skip("a HIP device is present")
    st = util.stream(33, 3, 4)
    dump_stream(str(tmp_path / "s.bin"), st, 3)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout


@pytest.mark.gpu
def test_whole_filter_facade_equals_the_python_engine(tmp_path):
    """ROFT::ROFTFilter (C++ facade, constructor of ROFTFilter.h:42-73 over in-memory sources, boot() / run()) ==
    roft_amd.engine.ROFTFilterBatch on the same stream, and both == the oracle."""
    import util
    from oracle import binding as ob
    from test_engine_gpu import make_engine
    n = 20
    st = util.stream(34, n, 2)
    dump_stream(str(tmp_path / "s.bin"), st, n)
    exe = build(tmp_path, "filter_check")
    subprocess.check_call([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")])
    got = np.fromfile(str(tmp_path / "o.bin"), np.float64).reshape(n, 19)
    eng = make_engine([st])
    ref = util.run_oracle_tracker(ob, st, n)
    for k in range(n):
        depth, flow, mask, pose = util.frame_inputs(st, k)
        eng.submit([dict(depth=depth, flow=flow, mask=mask, pose=pose, dt=st.dt)])
        eng.step()
        p, _, tw, _ = eng.state(0)
        assert np.array_equal(got[k, :13], p) and np.array_equal(got[k, 13:], tw), k
        assert np.abs(got[k, :13] - ref[k]["pose"]).max() < 1e-8
    eng.close()


def test_logger_probe_and_png_stand_ins(tmp_path):
    """bfl::Logger / RobotsIO::Utils::Probe(Container) / ImageFileProbe of Compat.h without a device: the log files carry
    what was logged in Eigen's default matrix format, a probe receives what set_data() hands it, and the PNG files of the
    image probe decode to the images written."""
    from roft_amd import io as rio
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "ROFT/Compat.h"
#include <cstdio>
struct L : bfl::Logger {
    std::vector<std::string> log_file_names(const std::string& p, const std::string& n) override { return {p + "/" + n + "a", p + "/" + n + "b"}; }
};
struct Keep : RobotsIO::Utils::Probe { int n = 0; double last = 0; void on_new_data() override { ++n; last = std::any_cast<Eigen::VectorXd>(get_data())(1); } };
int main(int, char** argv) {
    L l;
    Eigen::VectorXd v(3); v(0) = 1.0; v(1) = -0.25; v(2) = 1234567.0;
    l.logger(v.transpose(), v.transpose());                 // not enabled yet: nothing written
    if (!l.enable_log(argv[1], "x_")) return 1;
    if (l.enable_log(argv[1], "y_")) return 1;              // already enabled
    l.logger(v.transpose(), 7);
    l.logger(v.transpose(), 8);
    l.disable_log();
    RobotsIO::Utils::ProbeContainer c;
    if (c.is_probe("p")) return 1;
    auto* k = new Keep();
    c.set_probe("p", std::unique_ptr<RobotsIO::Utils::Probe>(k));
    if (!c.is_probe("p")) return 1;
    c.get_probe("p").set_data(v);
    if (k->n != 1 || k->last != -0.25) return 1;
    RobotsIO::Utils::ImageFileProbe ip(std::string(argv[1]) + "/img", "", "png");
    cv::Mat g(5, 300, CV_8UC1), bgr(70, 1000, CV_8UC3);    // the second one needs several stored deflate blocks
    for (int i = 0; i < 1500; ++i) g.data[i] = (unsigned char)(i * 7);
    for (int i = 0; i < 210000; ++i) bgr.data[i] = (unsigned char)(i * 13 + i / 3000);
    RobotsIO::Utils::Probe& p = ip;
    p.set_data(g);
    p.set_data(bgr);
    return 0;
}''')
    exe = str(tmp_path / "t")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    os.mkdir(tmp_path / "img")
    subprocess.check_call([exe, str(tmp_path)])
    rows = (tmp_path / "x_a.txt").read_text().splitlines()
    # Eigen's default IOFormat: 6 significant digits, every coefficient padded to the width of the widest one
    assert rows == [" ".join(c.rjust(11) for c in ("1", "-0.25", "1.23457e+06"))] * 2
    assert (tmp_path / "x_b.txt").read_text().split() == ["7", "8"]
    g = rio.read_png(str(tmp_path / "img" / "0.png"))
    assert g.shape == (5, 300) and np.array_equal(g.ravel(), (np.arange(1500) * 7).astype(np.uint8))
    c = rio.read_png(str(tmp_path / "img" / "1.png"))
    i = np.arange(210000)
    want = ((i * 13 + i // 3000) & 255).astype(np.uint8).reshape(70, 1000, 3)[:, :, ::-1]    # written BGR -> RGB
    assert c.shape == (70, 1000, 3) and np.array_equal(c, want)


def test_tracker_tail_compiles_against_the_facade(tmp_path):
    """src/roft/src/main.cpp:393-424 -- constructor call, set_probe, enable_log, boot / run / wait -- compiles against
    ROFT::ROFTFilter : bfl::FilteringAlgorithm, RobotsIO::Utils::ProbeContainer; without a device it fails loudly."""
    import util
    from roft_amd import _lib
    exe = build(tmp_path, "tracker_tail_check")
    if _lib.lib().roft_device_count() > 0:
        pytest.skip("a HIP device is present")
    st = util.stream(33, 3, 4)
    dump_stream(str(tmp_path / "s.bin"), st, 3)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout


@pytest.mark.gpu
def test_tracker_tail_logs_match_the_engine(tmp_path):
    """The five log files and the probe images the tail of ROFT-tracker's main() leaves, against the engine's own log
    (row for row, at the 6 significant digits bfl::Logger writes) and its propagated masks."""
    import util
    from roft_amd import io as rio
    from test_engine_gpu import make_engine
    n = 14
    st = util.stream(35, n, 2)
    dump_stream(str(tmp_path / "s.bin"), st, n)
    exe = build(tmp_path, "tracker_tail_check")
    for d in ("segmentation", "segmentation_refined"):
        os.mkdir(tmp_path / d)
    subprocess.check_call([exe, str(tmp_path / "s.bin"), str(tmp_path)])
    load = lambda name: np.loadtxt(str(tmp_path / (name + ".txt")), ndmin=2)
    pose_est, vel_est, times = load("pose_estimate"), load("velocity_estimate"), load("execution_times")
    pose_meas, vel_meas = load("pose_measurements"), load("velocity_measurements")
    assert pose_est.shape == (n, 13) and vel_est.shape == (n, 6) and times.shape == (n, 2)
    assert pose_meas.shape == (n, 7) and vel_meas.shape == (n, 6)
    assert np.all(times == np.round(times)) and np.all(times[:, 1] >= 0)      # integer milliseconds (ROFTFilter.cpp:463)
    eng = make_engine([st])
    last_pose = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])     # identity until the first pose arrives
    sig6 = lambda a: np.array([float("%.6g" % v) for v in np.ravel(a)])
    for k in range(n):
        depth, flow, mask, pose = util.frame_inputs(st, k)
        eng.submit([dict(depth=depth, flow=flow, mask=mask, pose=pose, dt=st.dt)])
        eng.step()
        p, _, tw, _ = eng.state(0)
        assert np.array_equal(pose_est[k], sig6(rio.pose_log_row(p))), k        # v w x axis angle
        assert np.array_equal(vel_est[k], sig6(tw)), k
        assert np.array_equal(vel_meas[k], sig6(tw)), k
        if pose is not None:
            last_pose = rio.pose_log_row(np.concatenate([np.zeros(6), pose[0], pose[1]]))[6:]
        assert np.array_equal(pose_meas[k], sig6(last_pose)), k
        refined = rio.read_png(str(tmp_path / "segmentation_refined" / ("%d.png" % k)))
        assert refined.shape == (st.camera.height, st.camera.width, 3)
        m = eng.mask(0) > 0
        assert np.array_equal(refined[:, :, 1] == 204, m) and not refined[:, :, 0].any()    # 0.8 * green over a black image
        outline = rio.read_png(str(tmp_path / "segmentation" / ("%d.png" % k)))
        assert outline.shape == refined.shape and outline[:, :, 0].max() == 255 and not outline[:, :, 1].any()
    eng.close()

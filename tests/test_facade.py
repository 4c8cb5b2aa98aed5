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
    want.update({
        "DatasetImageOpticalFlow.h": ["class DatasetImageOpticalFlow : public ImageOpticalFlowSource", "const std::size_t& heading_zeros = 0, const std::size_t& index_offset = 0"],
        "DatasetImageSegmentation.h": ["class DatasetImageSegmentation : public RobotsIO::Utils::Segmentation", "const bool simulate_missing_detections = false"],
        "DatasetImageSegmentationDelayed.h": ["class DatasetImageSegmentationDelayed : public DatasetImageSegmentation", "const float& fps, const float& simulated_fps, const bool simulate_inference_time",
                                              "int get_frames_between_iterations() const override"],
        "OpticalFlowUtilities.h": ["inline bool is_flow_valid(const float& f_x, const float& f_y)", "read_flow(const std::string& file_name)"],
    })
    for name, needles in want.items():
        text = " ".join(open(os.path.join(inc, name)).read().split())
        for needle in needles:
            assert needle in text, (name, needle)
    # every header name ROFT-tracker's main.cpp includes resolves (src/roft/src/main.cpp:8-30)
    for name in ("CameraMeasurement.h", "ImageOpticalFlowSource.h", "ImageOpticalFlowNVOF.h", "ModelParameters.h", "ImageSegmentationMeasurement.h"):
        assert os.path.exists(os.path.join(inc, name)), name
    for name in ("ConfigParser.h", "BayesFilters/FilteringAlgorithm.h", "RobotsIO/Camera/Camera.h", "RobotsIO/Camera/DatasetCamera.h",
                 "RobotsIO/Camera/CameraParameters.h", "RobotsIO/Utils/DatasetTransform.h", "RobotsIO/Utils/DatasetTransformDelayed.h",
                 "RobotsIO/Utils/ImageFileProbe.h", "RobotsIO/Utils/Parameters.h", "RobotsIO/Utils/Segmentation.h"):
        assert os.path.exists(os.path.join(ROOT, "include", "compat", name)), name


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


@pytest.mark.gpu
def test_filter_facade_in_place_inputs_equal_staged_inputs(tmp_path):
    """ROFT::ROFTFilter hands images that live in the library's pinned pool to the engine in place (zero copy) and the engine
    keeps REFERRING to them: a flow for as long as a delayed mask can be chased through it.  cv::Mat is not copy-on-write, so a
    source that rewrites one flow matrix every frame (a live source with a single output buffer; ROFT::ImageOpticalFlowHIP
    before round 5) must not be referred to -- the filter copies such a flow (flow_buffers_are_immutable()).  Masks delayed by
    six frames and chased through six flows; both pooled variants must equal the staged (HOST upload) run bit for bit."""
    import util
    n = 26
    st = util.stream(41, n, 2)
    dump_stream(str(tmp_path / "s.bin"), st, n)
    exe = build(tmp_path, "filter_check")
    rows = {}
    for mode, env in (("staged", {"ROFT_FACADE_STAGED": "1"}), ("pooled", {}), ("rewriter", {})):
        out = str(tmp_path / ("o_%s.bin" % mode))
        r = subprocess.run([exe, str(tmp_path / "s.bin"), out, "pooled" if mode == "staged" else mode],
                           env=dict(os.environ, ROFT_FILTER_TIMING="1", **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ("staged (HOST)" if mode == "staged" else "in-place (pinned)") in r.stdout, r.stdout
        rows[mode] = np.fromfile(out, np.float64).reshape(n, 19)
    assert np.array_equal(rows["pooled"], rows["staged"])
    assert np.array_equal(rows["rewriter"], rows["staged"])


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


# ---- the reference's executable sources against the facade ------------------------------------------------------------------
REF_MAIN = "/root/reference/src/roft/src/main.cpp"
REF_BIN = os.path.join(ROOT, "tests", "cpp", "_ref_build", "ROFT-tracker")


def build_sources_check(tmp_path):
    from roft_amd import _lib
    _lib.build()
    exe = str(tmp_path / "sources_check")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include", "compat"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "sources_check.cpp"), "-o", exe,
                           "-L", CSRC, "-lroft_hip", "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_reference_main_compiles_against_the_facade():
    """src/roft/src/main.cpp of the reference, UNMODIFIED and where it lies, compiles against include/ (+ include/compat for the
    header names of the third-party libraries): the drop-in claim of the class API.  Dev container only."""
    if not os.path.exists(REF_MAIN):
        pytest.skip("the reference checkout is not here")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include", "compat"), "-I", os.path.join(ROOT, "include"), REF_MAIN])


def test_config_parser_cpp_reads_what_the_python_reader_reads(tmp_path):
    """include/compat/ConfigParser.h against roft_amd/config.py on a complete configuration with command-line overrides of
    every type (as test/test.sh passes them), and -- in the dev container -- on the reference's own configuration files."""
    from roft_amd import config as K
    exe = build_sources_check(tmp_path)
    cfg_path = str(tmp_path / "tracker.cfg")
    open(cfg_path, "w").write("# a comment\n" + K.tracker_text(640, 480, 614.7, 615.5, 320.0, 240.0) + "/* trailing\n comment */\n")
    over = ["--camera_dataset::fx", "600.25", "--camera_dataset::width", "320", "--initial_condition::pose::x", "0.1, -0.2,0.7",
            "--initial_condition::pose::axis_angle", "0,0,1,0.5", "--measurement_model::use_pose_resync", "false", "--model::name", "003_cracker_box",
            "--log::path", str(tmp_path), "--optical_flow_dataset::set", "nvof_1_slow/", "--measurement_model::pose::cov_q", "0.01,0.02,0.03",
            "--kinematic_model::pose::sigma_angular", "1e-2,0.01,0.01"]
    cases = [(cfg_path, over), (cfg_path, [])]
    for name in ("config_fast_ycb.cfg", "config_ho3d.cfg"):
        p = os.path.join("/root/reference/config", name)
        if os.path.exists(p):
            cases.append((p, over))
    for path, ov in cases:
        out = str(tmp_path / "cfg_out.txt")
        subprocess.check_call([exe, "cfg", out, "ROFT-tracker", "--from", path] + ov)
        got = dict(line.rstrip("\n").split("=", 1) for line in open(out))
        cfg = K.parse_cfg(open(path).read())
        assert K.apply_overrides(cfg, ["--from", path] + ov) == []
        assert len(got) == 76 == len(list(K.all_keys(cfg)))
        for key in K.all_keys(cfg):
            want = K.lookup(cfg, key)
            if isinstance(want, bool):
                assert got[key] == ("true" if want else "false"), key
            elif isinstance(want, list):
                assert [float(v) for v in got[key].split(",")] == [float(v) for v in want], key
            elif isinstance(want, str):
                assert got[key] == want, key
            else:
                assert float(got[key]) == float(want), key
    # errors are loud: an argument that is not a setting, a malformed value, a missing file
    for bad in (["--no_such::key", "1"], ["--camera_dataset::width", "1.5"], ["--measurement_model::use_pose", "yes"],
                ["--initial_condition::pose::x", "1,2"]):
        r = subprocess.run([exe, "cfg", str(tmp_path / "x.txt"), "ROFT-tracker", "--from", cfg_path] + bad, capture_output=True, text=True)
        assert r.returncode == 3 and "ConfigParser" in r.stdout, (bad, r.stdout)
    r = subprocess.run([exe, "cfg", str(tmp_path / "x.txt"), "ROFT-tracker"], capture_output=True, text=True)
    assert r.returncode == 3 and "--from" in r.stdout


def _png_with_filters(path, img, level=6):
    """PNG with a different filter type per row (0 .. 4 round robin) and real deflate blocks (dynamic Huffman codes)."""
    import zlib
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    rows = img.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft = y % 5
        cur = rows[y]
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if ft == 0:
            pr = 0
        elif ft == 1:
            pr = a
        elif ft == 2:
            pr = prev
        elif ft == 3:
            pr = (a + prev) >> 1
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pr = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        raw += bytes([ft]) + ((cur - pr) & 255).astype(np.uint8).tobytes()
        prev = cur
    chunk = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    z = zlib.compress(bytes(raw), level)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, {1: 0, 2: 4, 3: 2, 4: 6}[ch], 0, 0, 0)) +
                chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))


def test_png_decoder_of_the_file_sources(tmp_path):
    """compat::read_png (its own inflate) against roft_amd.io.read_png (zlib): gray, gray + alpha, RGB, RGBA; all five filter
    types; stored, fixed and dynamic deflate blocks; BGR order and the OpenCV gray conversion."""
    from roft_amd import io
    exe = build_sources_check(tmp_path)
    rng = np.random.default_rng(3)
    smooth = (np.add.outer(np.arange(37), np.arange(53)) * 3 % 256).astype(np.uint8)
    cases = {"gray": smooth, "noise": rng.integers(0, 256, (29, 31), dtype=np.uint8), "mask": (smooth > 128).astype(np.uint8) * 255,
             "rgb": np.stack([smooth, smooth.T[:37, :53] if False else 255 - smooth, rng.integers(0, 256, smooth.shape, dtype=np.uint8)], -1),
             "rgba": rng.integers(0, 256, (17, 19, 4), dtype=np.uint8), "ga": rng.integers(0, 256, (8, 9, 2), dtype=np.uint8)}
    for name, img in cases.items():
        for level in (0, 1, 9):
            p = str(tmp_path / ("%s_%d.png" % (name, level)))
            _png_with_filters(p, img, level)
            ref = io.read_png(p)
            assert np.array_equal(ref, img)
            for mode in ("png", "gray"):
                out = str(tmp_path / "png.bin")
                subprocess.check_call([exe, mode, p, out])
                raw = open(out, "rb").read()
                rows, cols, esz = struct.unpack("3i", raw[:12])
                got = np.frombuffer(raw[12:], np.uint8).reshape(rows, cols, esz) if esz > 1 else np.frombuffer(raw[12:], np.uint8).reshape(rows, cols)
                if img.ndim == 2 or img.shape[2] == 2:
                    want = img if img.ndim == 2 else img[..., 0]
                elif mode == "png":
                    want = img[..., 2::-1]                      # B, G, R as cv::imread orders them
                else:
                    want = io.rgb_to_gray(img)
                assert np.array_equal(got, want), (name, level, mode)
    # written by the stand-in of cv::imwrite (stored blocks) and read back
    io.write_png(str(tmp_path / "w.png"), smooth)
    subprocess.check_call([exe, "png", str(tmp_path / "w.png"), str(tmp_path / "png.bin")])
    assert np.array_equal(np.frombuffer(open(str(tmp_path / "png.bin"), "rb").read()[12:], np.uint8).reshape(smooth.shape), smooth)
    # a missing file and a truncated one come back empty
    open(str(tmp_path / "bad.png"), "wb").write(open(str(tmp_path / "w.png"), "rb").read()[:60])
    for p in ("none.png", "bad.png"):
        subprocess.check_call([exe, "png", str(tmp_path / p), str(tmp_path / "png.bin")])
        assert struct.unpack("3i", open(str(tmp_path / "png.bin"), "rb").read()[:12]) == (0, 0, 0)


def test_file_sources_deliver_on_the_references_schedule(tmp_path):
    """DatasetImageSegmentation(Delayed), DatasetTransform(Delayed) and DatasetCamera over a directory roft_amd.io wrote: masks
    and poses arrive on the frames roft_amd.io.delivery_schedule (the restatement of DatasetImageSegmentationDelayed.cpp:42-63
    the Python sequence reader uses) says, missing detections are skipped, stamps / depth / camera pose are those of the files."""
    from roft_amd import io, synth
    exe = build_sources_check(tmp_path)
    root, n, w, h = str(tmp_path / "seq"), 20, 8, 6
    for d in ("rgb", "depth", "masks/gt", "dope"):
        os.makedirs(os.path.join(root, d))
    io.write_data_txt(os.path.join(root, "data.txt"), n, 30.0)
    pose = np.zeros((n, 7))
    pose[:, 3] = 1.0
    pose[:, 0] = 0.25 + 0.125 * np.arange(n)
    ok = np.ones(n, bool)
    ok[12] = False                                              # a missing detection: its slot stays empty
    io.write_poses(os.path.join(root, "dope", "poses.txt"), pose, ok)
    for k in range(n):
        io.write_png(os.path.join(root, "masks", "gt", "box_%d.png" % k), np.full((h, w), 10 + k, np.uint8))
        io.write_png(os.path.join(root, "rgb", "%d.png" % k), np.full((h, w), 100 + k, np.uint8))
        io.write_depth(os.path.join(root, "depth", "%d.float" % k), np.full((h, w), 0.5 + k, np.float32))
    for fps, sim, delayed in ((30.0, 5.0, 1), (30.0, 5.0, 0), (30.0, 10.0, 1), (30.0, 30.0, 1)):
        r = subprocess.run([exe, "sched", root, "box", "gt", os.path.join(root, "dope", "poses.txt"), str(n), str(w), str(h), str(fps), str(sim), str(delayed)],
                           capture_output=True, text=True, check=True)
        lines = r.stdout.strip().splitlines()
        assert lines[0].split() == ["between", str(int(fps / sim)), str(int(fps / sim)), "-1", "-1"]
        sched = io.delivery_schedule(n, fps, sim, simulate_inference_time=bool(delayed))
        for k, line in enumerate(lines[1:]):
            f = line.split()
            assert int(f[0]) == k
            assert int(f[1]) == (10 + sched[k] if sched[k] >= 0 else -1), (fps, sim, delayed, k)
            want = pose[sched[k], 0] if sched[k] >= 0 and ok[sched[k]] else float("nan")
            assert (np.isnan(want) and f[2] == "nan") or float(f[2]) == want, (fps, sim, delayed, k, f[2], want)
            assert int(f[3]) == 10 + k                          # the plain sources: every frame its own mask and pose
            assert (not ok[k] and f[4] == "nan") or float(f[4]) == pose[k, 0]
    # started in the middle of the sequence as test/test_ho3d.sh:142-160 does (index_offset of camera / flow / mask sources,
    # skip_rows of the pose file): the schedule of roft_amd.io.Sequence(first_frame=...)
    for first in (6, 12):
        r = subprocess.run([exe, "sched", root, "box", "gt", os.path.join(root, "dope", "poses.txt"), str(n), str(w), str(h), "30.0", "5.0", "1", str(first)],
                           capture_output=True, text=True, check=True)
        lines = r.stdout.strip().splitlines()[1:]
        seq = io.Sequence(root, "box", flow_set="none", mask_set="gt", pose_set="dope", width=w, height=h, first_frame=first)
        assert len(lines) == n - first
        for line in lines:
            f = line.split()
            k = int(f[0])
            assert int(f[1]) == (10 + seq.mask_src[k] if seq.mask_src[k] >= 0 else -1), (first, k)
            pi = seq.pose_src[k]
            want = pose[pi, 0] if pi >= 0 and ok[pi] else float("nan")
            assert (np.isnan(want) and f[2] == "nan") or float(f[2]) == want, (first, k, f[2], want)
        assert seq.pose_src[first] == first and seq.pose_src[first + 6] == first and seq.mask_src[first] == first - 6
        r = subprocess.run([exe, "camera", root, str(w), str(h), str(first)], capture_output=True, text=True, check=True)
        rows = [line.split() for line in r.stdout.strip().splitlines() if not line.startswith("Dataset")]
        assert [int(f[0]) for f in rows] == list(range(first, n)) and float(rows[0][4]) == 0.5 + first and int(rows[0][8]) == 100 + first
        assert float(rows[0][1]) == io.read_data_txt(os.path.join(root, "data.txt"))[0][first]
    r = subprocess.run([exe, "camera", root, str(w), str(h)], capture_output=True, text=True, check=True)
    rows = [line.split() for line in r.stdout.strip().splitlines() if not line.startswith("Dataset")]
    assert len(rows) == n
    stamps, _, _ = io.read_data_txt(os.path.join(root, "data.txt"))
    for k, f in enumerate(rows):
        assert int(f[0]) == k and float(f[1]) == stamps[k] == float(f[2]) and int(f[3]) == 1
        assert float(f[4]) == 0.5 + k == float(f[5]) and float(f[6]) == 0.0 and float(f[7]) == 1.0 and int(f[8]) == 100 + k


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["A_f32_grid1", "B_s16_grid4", "A_from_first_detection", "A_internal_mesh_db"])
def test_reference_tracker_runs_on_this_engine(tmp_path, capsys, shape):
    """The reference's own src/roft/src/main.cpp -- unmodified, built by __graft_entry__.build() in the dev container against
    include/ROFT + include/compat and linked with libroft_hip.so -- started the way test/test.sh starts ROFT-tracker on a
    sequence directory: it must write the five log files, and they must say what tools/run_sequence.py (the Python host over
    the same engine, same files, same configuration and overrides) says, to the six digits bfl::Logger prints."""
    import importlib.util
    import json
    import util
    from roft_amd import config as K
    from roft_amd import io, synth
    if not os.path.exists(REF_BIN):
        pytest.skip("tests/cpp/_ref_build/ROFT-tracker is built where the reference checkout is (python __graft_entry__.py)")
    import copy
    if shape != "B_s16_grid4":     # config_ho3d.cfg's shape: 640x480 (here halved), CV_32FC2 flow per pixel
        n = 40
        st = copy.copy(util.stream(703, n, 2, with_gray=True))
    else:                          # config_fast_ycb.cfg's: 1280x720, CV_16SC2 flow on a grid of 4, render divider 4
        n = 20
        st = copy.copy(util.stream(704, n, 1, shape="B", flow_type=synth.FLOW_S16C2, mesh_n=24, with_gray=True, device="cuda"))
    st.pose_meas = st.pose_meas.copy()
    st.pose_meas[0] = st.pose_meas[6]            # one detection per source frame in a pose file (see test_sequence_gpu.py)
    root = str(tmp_path / "seq")
    obj_name, env = "box", dict(os.environ)
    mesh = io.write_sequence(root, st, "box", flow_set="analytic")
    if shape == "A_internal_mesh_db":
        # test/test.sh leaves model.use_internal_db = true / internal_db_name = "DOPE" of the configuration file alone and
        # only names the object: the mesh comes from the library's data base -- here the directory ROFT_MESH_DB names, holding
        # the reference's own 003_cracker_box.obj (copied next to the binary by __graft_entry__.build())
        db = os.path.join(os.path.dirname(REF_BIN), "meshes")
        if not os.path.exists(os.path.join(db, "DOPE", "003_cracker_box.obj")):
            pytest.skip("no mesh data base next to the binary")
        obj_name, env["ROFT_MESH_DB"] = "003_cracker_box", db
        mesh = os.path.join(db, "DOPE", "003_cracker_box.obj")
        for k in range(n):
            os.rename(os.path.join(root, "masks", "gt", "box_%d.png" % k), os.path.join(root, "masks", "gt", "%s_%d.png" % (obj_name, k)))
    c = st.camera
    cfg_path = str(tmp_path / "config.cfg")
    open(cfg_path, "w").write(K.tracker_text(c.width, c.height, 1.0, 1.0, 0.0, 0.0))
    m0 = synth.initial_pose_from_stream(st)
    axis, angle = io.quat_to_axis_angle(m0[9:13])
    first, start_args, py_start = 0, [], []
    if shape == "A_from_first_detection":
        # test/test_ho3d.sh:68-72, 142-160: no detection on frame 0 -> the tracker starts at the frame tools/dataset/
        # dope_pose_finder/pose_finder.py reports, with that detection as its initial pose
        pp = os.path.join(root, "dope", "poses.txt")
        rows = open(pp).read().splitlines()
        rows[0] = "0.0 0.0 0.0 0.0 0.0 0.0 0.0"
        open(pp, "w").write("\n".join(rows) + "\n")
        first, line = io.find_initial_pose(pp, 5.0)
        assert first % 6 == 0 and 12 <= first <= 24
        aa = [float(v) for v in line.split()]
        m0 = np.array(m0)
        m0[6:9], axis, angle = aa[:3], aa[3:6], aa[6]
        start_args = ["--camera_dataset::index_offset", str(first), "--optical_flow_dataset::index_offset", str(first),
                      "--pose_dataset::skip_rows", str(first), "--segmentation_dataset::index_offset", str(first)]
        py_start = ["--start-at-first-detection"]
        n -= first
    out_dir = str(tmp_path / "out")
    os.makedirs(out_dir)
    # test/test.sh:135-157, with the mesh as a file instead of the compiled-in data base
    args = ["--from", cfg_path,
            "--camera_dataset::fx", repr(c.fx), "--camera_dataset::fy", repr(c.fy), "--camera_dataset::cx", repr(c.cx), "--camera_dataset::cy", repr(c.cy),
            "--camera_dataset::path", root,
            "--initial_condition::pose::x", ",".join("%.17g" % v for v in m0[6:9]),
            "--initial_condition::pose::axis_angle", ",".join("%.17g" % v for v in list(axis) + [angle]),
            "--kinematic_model::pose::sigma_angular", "1.0,1.0,1.0",
            "--log::path", out_dir, "--log::enable_segmentation", "true",
            "--measurement_model::pose::cov_q", "0.0001,0.0001,0.0001",
            "--measurement_model::use_pose", "true", "--measurement_model::use_pose_resync", "true", "--measurement_model::use_velocity", "true",
            "--model::name", obj_name] + ([] if shape == "A_internal_mesh_db" else ["--model::use_internal_db", "false", "--model::external_path", mesh]) + [
            "--optical_flow_dataset::path", root, "--optical_flow_dataset::set", "analytic/",
            "--outlier_rejection::enable", "true",
            "--pose_dataset::path", os.path.join(root, "dope", "poses.txt"),
            "--segmentation_dataset::flow_aided", "true", "--segmentation_dataset::path", root, "--segmentation_dataset::set", "gt"] + start_args
    for d in ("segmentation", "segmentation_refined"):
        os.makedirs(os.path.join(out_dir, d))
    r = subprocess.run([REF_BIN] + args, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DatasetImageOpticalFlow::ctor." in r.stdout and "Unscented transform:" in r.stdout
    assert ("grid size: 1" in r.stdout and "CV_32FC2" in r.stdout) if shape != "B_s16_grid4" else ("grid size: 4" in r.stdout and "scaling factor: 32" in r.stdout)
    logs = {name: io.read_log(os.path.join(out_dir, name + ".txt")) for name in
            ("pose_estimate", "velocity_estimate", "execution_times", "pose_measurements", "velocity_measurements")}
    assert logs["pose_estimate"].shape == (n, 13) and logs["velocity_estimate"].shape == (n, 6) and logs["execution_times"].shape == (n, 2)
    assert logs["pose_measurements"].shape == (n, 7) and logs["velocity_measurements"].shape == (n, 6)
    assert len(os.listdir(os.path.join(out_dir, "segmentation"))) == n == len(os.listdir(os.path.join(out_dir, "segmentation_refined")))
    # the Python host over the same engine
    spec = importlib.util.spec_from_file_location("run_sequence", os.path.join(ROOT, "tools", "run_sequence.py"))
    rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rs)
    keep = [a for a in args[2:]]
    drop = {"--camera_dataset::index_offset", "--optical_flow_dataset::index_offset", "--pose_dataset::skip_rows", "--segmentation_dataset::index_offset",
            "--camera_dataset::path", "--log::path", "--log::enable_segmentation", "--model::name", "--model::use_internal_db", "--model::external_path",
            "--optical_flow_dataset::path", "--optical_flow_dataset::set", "--pose_dataset::path", "--segmentation_dataset::path", "--segmentation_dataset::set"}
    over = []
    for i in range(0, len(keep), 2):
        if keep[i] not in drop:
            over += keep[i:i + 2]
    open(str(tmp_path / "filter.cfg"), "w").write(K.default_text(c.width, c.height, 1.0, 1.0, 0.0, 0.0))
    assert rs.main(["--root", root, "--object", obj_name, "--mesh", mesh, "--flow-set", "analytic", "--mask-set", "gt", "--out", str(tmp_path / "py_"),
                    "--from", str(tmp_path / "filter.cfg")] + over + py_start) == 0
    rep = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rep["frames"] == n and rep["first_frame"] == first and rep["adds_auc"] > (80.0 if shape == "A_f32_grid1" else 60.0)
    est = np.loadtxt(str(tmp_path / "py_pose_estimate"))
    vel = np.loadtxt(str(tmp_path / "py_velocity_estimate"))
    # six significant digits per value in the C++ logs
    assert np.allclose(logs["pose_estimate"], est, rtol=2e-5, atol=1e-6), np.abs(logs["pose_estimate"] - est).max()
    assert np.allclose(logs["velocity_estimate"], vel, rtol=2e-5, atol=1e-6), np.abs(logs["velocity_estimate"] - vel).max()
    # ... and against the CPU ORACLE (round 6; the comparison above is engine against engine -- an API test): the same sequence
    # directory, read frame by frame through roft_amd.io, through oracle/ro_tracker.c with the configuration this command line
    # amounts to (test/test.sh's settings = the oracle's defaults; the initial pose and the camera as passed).  What the reference's
    # executable logged over the HIP engine must be the oracle's trajectory to the six digits bfl::Logger prints.
    if shape in ("A_f32_grid1", "B_s16_grid4", "A_from_first_detection"):
        from oracle import binding as ob
        seq = io.Sequence(root, obj_name, flow_set="analytic", mask_set="gt", pose_set="dope", width=c.width, height=c.height, delayed=True, first_frame=first)
        ocfg = ob.default_config(640 if c.width in (640, 320, 160) else 1280, c.height)
        ocfg.cam.width, ocfg.cam.height = c.width, c.height
        ocfg.cam.fx, ocfg.cam.fy, ocfg.cam.cx, ocfg.cam.cy = c.fx, c.fy, c.cx, c.cy
        for i in range(6):
            ocfg.p_mean0[i] = 0.0
        for i in range(3):
            ocfg.p_mean0[6 + i] = float(m0[6 + i])
        q0 = io.axis_angle_to_quat(np.asarray(axis, float), float(angle))
        for i in range(4):
            ocfg.p_mean0[9 + i] = float(q0[i])
        verts_o, tris_o = io.load_obj(mesh)
        trk = ob.Tracker(ocfg, verts_o, tris_o)
        ref_pose, ref_vel = [], []
        for k in range(first, len(seq)):
            fr = seq.frame(k)
            res = trk.step(fr["dt"] if fr["dt"] > 0 else st.dt, fr["depth"], fr["flow"], fr["mask"], fr["pose"])
            ref_pose.append(io.pose_log_row(np.array(res.pose)))
            ref_vel.append(np.array(res.twist))
        trk.close()
        ref_pose, ref_vel = np.array(ref_pose), np.array(ref_vel)
        assert ref_pose.shape == logs["pose_estimate"].shape
        assert np.allclose(logs["pose_estimate"], ref_pose, rtol=2e-5, atol=2e-6), np.abs(logs["pose_estimate"] - ref_pose).max()
        assert np.allclose(logs["velocity_estimate"], ref_vel, rtol=2e-5, atol=2e-6), np.abs(logs["velocity_estimate"] - ref_vel).max()


def test_queue_handler_and_mesh_resource(tmp_path):
    """OpticalFlowQueueHandler (window of the last n flows, the region AFTER the entry with a stamp, 1 ms tolerance, unknown
    stamp -> empty) and MeshResource (external file; the internal data base = $ROFT_MESH_DB/<set>/<name>.obj; loud when absent)."""
    exe = build_sources_check(tmp_path)
    r = subprocess.run([exe, "queue"], capture_output=True, text=True, check=True)
    assert r.stdout.split("\n")[:7] == ["0.133333: 5 6", "0.133833: 5 6", "0.135333:", "0.066667:", "0.200000:", "0.100000: 4 5 6", "0.133333:"]
    db = tmp_path / "db" / "DOPE"
    db.mkdir(parents=True)
    (db / "box.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "ext.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1 2 4\n")
    env = dict(os.environ, ROFT_MESH_DB=str(tmp_path / "db"))
    r = subprocess.run([exe, "mesh", "box", "DOPE", str(tmp_path / "ext.obj")], capture_output=True, text=True, env=env, check=True)
    n_int, n_ext = len((db / "box.obj").read_text()), len((tmp_path / "ext.obj").read_text())
    assert r.stdout.split() == ["external", str(n_ext), "internal", str(n_int), "named", str(n_int)]
    env.pop("ROFT_MESH_DB")
    r = subprocess.run([exe, "mesh", "box", "DOPE", str(tmp_path / "ext.obj")], capture_output=True, text=True, env=env)
    assert r.returncode == 3 and ("external %d" % n_ext) in r.stdout and "Cannot find requested mesh among available resources" in r.stdout
    r = subprocess.run([exe, "mesh", "box", "DOPE", str(tmp_path / "none.obj")], capture_output=True, text=True, env=env)
    assert r.returncode == 3 and "Cannot open model from external path" in r.stdout


@pytest.mark.gpu
def test_stamped_source_facade_equals_the_oracle(tmp_path):
    """ROFT::ImageSegmentationOFAidedSourceStamped<cv::Vec2f> (host bookkeeping + roft_mask_propagate) against the oracle's
    restatement of the same source on an irregular live delivery (tests/test_stamped.py::schedule: late masks, a stamp that is not
    in the queue, a mask older than the frames_between window): the mask held after every frame, bit for bit."""
    import util
    import test_stamped as ts
    from oracle import binding as ob
    n = 48
    st = util.stream(95, n, 4)
    deliver = ts.schedule(n, 5)
    want = ts.run_oracle(ob, st, n, deliver)
    # the recorded stream with the masks on their delivery frames
    import copy
    rec = copy.copy(st)
    rec.mask_delivery = np.full(n, -1, np.int64)
    for k, (src, _stamp) in deliver.items():
        rec.mask_delivery[k] = src
    dump_stream(str(tmp_path / "s.bin"), rec, n)
    with open(str(tmp_path / "stamps.txt"), "w") as f:
        for k in range(n):
            f.write("%.17g %.17g\n" % (k / 30.0, deliver[k][1] if k in deliver else -1.0))
    exe = build(tmp_path, "stamped_check")
    subprocess.check_call([exe, str(tmp_path / "s.bin"), str(tmp_path / "stamps.txt"), "6", str(tmp_path / "out.bin")])
    raw = open(str(tmp_path / "out.bin"), "rb").read()
    H, W = st.mask_gt.shape[1:]
    pos = 0
    for k in range(n):
        have = raw[pos]
        pos += 1
        assert have == 1
        got = np.frombuffer(raw[pos:pos + W * H], np.uint8).reshape(H, W)
        pos += W * H
        assert np.array_equal(np.where(got > 1, 255, 0).astype(np.uint8), want[k]["mask"]), k
    assert pos == len(raw)


REF_DUMPER_MAIN = "/root/reference/tools/nvof/dumper/src/main.cpp"
REF_DUMPER_BIN = os.path.join(ROOT, "tests", "cpp", "_ref_build", "ROFT-of-dumper")


def test_reference_flow_dumper_compiles_against_the_facade():
    """tools/nvof/dumper/src/main.cpp of the reference (ROFT-of-dumper), UNMODIFIED and where it lies, against include/: the
    executable of the step before the path.  Dev container only."""
    if not os.path.exists(REF_DUMPER_MAIN):
        pytest.skip("the reference checkout is not here")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include", "compat"), "-I", os.path.join(ROOT, "include"), REF_DUMPER_MAIN])


@pytest.mark.gpu
@pytest.mark.parametrize("version", ["nvof1", "nvof2"])
def test_reference_flow_dumper_runs_on_the_hip_producer(tmp_path, version):
    """The built ROFT-of-dumper, started with the reference's nine arguments on a sequence directory: `<index>.float` for every
    frame but the first, each file what roft_optical_flow gives for the two RGB frames (nvof1: CV_16SC2 on a grid of 4, nvof2:
    CV_32FC2 per pixel), readable by the file-backed flow source; wrong argument counts end with the synopsis."""
    import util
    from roft_amd import io, ops, synth
    if not os.path.exists(REF_DUMPER_BIN):
        pytest.skip("tests/cpp/_ref_build/ROFT-of-dumper is built where the reference checkout is (python __graft_entry__.py)")
    n = 5
    st = util.stream(712, n, 2, with_gray=True)
    root = str(tmp_path / "seq")
    io.write_sequence(root, st, "box")
    H, W = st.mask_gt.shape[1:]
    out = tmp_path / "flow_out"
    out.mkdir()
    r = subprocess.run([REF_DUMPER_BIN, root, "txt", "png", "0", "0", str(W), str(H), version, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Processing completed." in r.stdout
    assert sorted(os.listdir(out)) == ["%d.float" % k for k in range(1, n)]
    gray = [io.rgb_to_gray(io.read_png(os.path.join(root, "rgb", "%d.png" % k))) for k in range(n)]
    for k in range(1, n):
        want = ops.optical_flow(gray[k - 1], gray[k], flow_type=synth.FLOW_S16C2 if version == "nvof1" else synth.FLOW_F32C2)
        valid, got = io.read_flow(str(out / ("%d.float" % k)))
        assert valid and got.dtype == want.dtype and got.shape == want.shape, k
        assert np.array_equal(got, want), k
    bad = subprocess.run([REF_DUMPER_BIN, root], capture_output=True, text=True)
    assert bad.returncode != 0 and "Synopsis: ROFT-of-dumper" in bad.stderr
    bad = subprocess.run([REF_DUMPER_BIN, root, "txt", "png", "0", "0", str(W), str(H), "nvof3", str(out)], capture_output=True, text=True)
    assert bad.returncode != 0 and "Invalid <nvof_version>" in bad.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("product", [1, 2])
def test_live_flow_source_with_the_references_constructor(tmp_path, product):
    """ROFT::ImageOpticalFlowNVOF(camera_measurement, performance, hints) -- the reference's constructor (ImageOpticalFlowNVOF.h:43-46)
    over the HIP flow producer -- fed by CameraMeasurement(DatasetCamera) from a sequence directory: no flow on the first frame,
    then frame after frame what roft_optical_flow gives for the two images, in the format of the product (1: CV_16SC2 on a grid of
    4, scale 32; 2: CV_32FC2 per pixel)."""
    import util
    from roft_amd import io, ops, synth
    n = 5
    st = util.stream(710, n, 2, with_gray=True)
    root = str(tmp_path / "seq")
    io.write_sequence(root, st, "box")
    H, W = st.mask_gt.shape[1:]
    exe = build_sources_check(tmp_path)
    out = str(tmp_path / "flow.bin")
    subprocess.check_call([exe, "nvof", root, str(W), str(H), str(product), out])
    raw = open(out, "rb").read()
    gray = [io.rgb_to_gray(io.read_png(os.path.join(root, "rgb", "%d.png" % k))) for k in range(n)]
    pos = 0
    for k in range(n):
        valid, rows, cols, typ = struct.unpack("4i", raw[pos:pos + 16])
        pos += 16
        assert valid == (k > 0)
        if not valid:
            continue
        want = ops.optical_flow(gray[k - 1], gray[k], flow_type=synth.FLOW_S16C2 if product == 1 else synth.FLOW_F32C2)
        assert (rows, cols) == want.shape[:2] and typ == (11 if product == 1 else 13)
        nbytes = want.size * want.itemsize
        got = np.frombuffer(raw[pos:pos + nbytes], want.dtype).reshape(want.shape)
        pos += nbytes
        assert np.array_equal(got, want), k
    assert pos == len(raw)


def test_the_references_meshes_load(tmp_path):
    """Dev container only: the seven meshes the reference compiles into its library (src/roft-lib/meshes/DOPE/*.obj, Meshlab
    output with `vn` lines and `f a//n b//n c//n` faces) through the facade's OBJ reader and through roft_amd.io.load_obj: same
    vertices and triangles; and the directory IS a valid ROFT_MESH_DB for MeshResource's internal data base."""
    from roft_amd import io
    db = "/root/reference/src/roft-lib/meshes"
    if not os.path.isdir(db):
        pytest.skip("the reference checkout is not here")
    exe = build_sources_check(tmp_path)
    names = sorted(f for f in os.listdir(os.path.join(db, "DOPE")) if f.endswith(".obj"))
    assert len(names) >= 6
    for name in names:
        path = os.path.join(db, "DOPE", name)
        verts, tris = io.load_obj(path)
        r = subprocess.run([exe, "obj", path], capture_output=True, text=True, check=True)
        f = r.stdout.split()
        assert int(f[0]) == len(verts) and int(f[1]) == len(tris) and len(verts) > 5000 and len(tris) > 10000
        assert tris.min() == 0 and tris.max() == len(verts) - 1
        flat = tris.reshape(-1).astype(np.int64)
        assert int(f[2]) == int((flat * (np.arange(flat.size) % 7 + 1)).sum())
        lo, hi = verts.min(0), verts.max(0)
        assert np.allclose([float(v) for v in f[3:]], [lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]], rtol=0, atol=1e-7)
    # SURVEY 8d quotes the extents of the cracker box
    verts, _ = io.load_obj(os.path.join(db, "DOPE", "003_cracker_box.obj"))
    assert np.allclose(verts.min(0), [-0.080, -0.102, -0.036], atol=2e-3) and np.allclose(verts.max(0), [0.084, 0.112, 0.035], atol=2e-3)
    env = dict(os.environ, ROFT_MESH_DB=db)
    (tmp_path / "e.obj").write_text("v 0 0 0\n")
    r = subprocess.run([exe, "mesh", "003_cracker_box", "DOPE", str(tmp_path / "e.obj")], capture_output=True, text=True, env=env, check=True)
    size = os.path.getsize(os.path.join(db, "DOPE", "003_cracker_box.obj"))
    assert r.stdout.split() == ["external", "8", "internal", str(size), "named", str(size)]


@pytest.mark.gpu
def test_batched_tracker_front_end_equals_one_tracker_per_object(tmp_path, capsys):
    """tools/track_many.cpp (ROFT-tracker-batch): ROFT-tracker's configuration file, overrides and file-backed sources, but three
    objects -- three sequence directories -- advanced by ONE engine in batches of six frames.  Objects do not interact in
    ROFTFilter::filtering_step, so every object's logs must say what a tracker of its own says (tools/run_sequence.py over
    the same engine code, frame by frame), to the digits bfl::Logger prints."""
    import copy
    import importlib.util
    import json
    import util
    from roft_amd import _lib
    from roft_amd import config as K
    from roft_amd import io
    _lib.build()
    exe = str(tmp_path / "ROFT-tracker-batch")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include", "compat"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "track_many.cpp"), "-o", exe, "-L", CSRC, "-lroft_hip", "-Wl,-rpath," + CSRC,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    n = 32
    seqs = []
    for i in range(3):
        st = copy.copy(util.stream(720 + i, n, 2, with_gray=True))
        st.pose_meas = st.pose_meas.copy()
        st.pose_meas[0] = st.pose_meas[6]
        name = "box%d" % i
        root = str(tmp_path / "dataset" / name)            # the reference's layout: one directory per object / sequence
        os.makedirs(root)
        mesh = io.write_sequence(root, st, name, flow_set="analytic")
        seqs.append((root, name, mesh, st))
    c = seqs[0][3].camera
    cfg_path = str(tmp_path / "config.cfg")
    open(cfg_path, "w").write(K.tracker_text(c.width, c.height, c.fx, c.fy, c.cx, c.cy))
    args = ["--from", cfg_path, "--pose_dataset::path", "dope/poses.txt", "--optical_flow_dataset::set", "analytic", "--segmentation_dataset::set", "gt",
            "--model::use_internal_db", "false", "--measurement_model::pose::cov_q", "0.0002,0.0002,0.0002",
            "--log_root", str(tmp_path / "out"), "--batch_frames", "6"]
    for root, name, mesh, _ in seqs:
        args += ["--object", root, name, mesh]
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "tracked 3 objects over %d frames" % n in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    spec = importlib.util.spec_from_file_location("run_sequence", os.path.join(ROOT, "tools", "run_sequence.py"))
    rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rs)
    open(str(tmp_path / "filter.cfg"), "w").write(K.default_text(c.width, c.height, c.fx, c.fy, c.cx, c.cy))
    for i, (root, name, mesh, _) in enumerate(seqs):
        assert rs.main(["--root", root, "--object", name, "--mesh", mesh, "--flow-set", "analytic", "--mask-set", "gt", "--out", str(tmp_path / ("py%d_" % i)),
                        "--from", str(tmp_path / "filter.cfg"), "--measurement_model::pose::cov_q", "0.0002,0.0002,0.0002"]) == 0
        rep = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        assert rep["frames"] == n
        est = np.loadtxt(str(tmp_path / ("py%d_pose_estimate" % i)))
        vel = np.loadtxt(str(tmp_path / ("py%d_velocity_estimate" % i)))
        got_p = io.read_log(str(tmp_path / "out" / name / "pose_estimate.txt"))
        got_v = io.read_log(str(tmp_path / "out" / name / "velocity_estimate.txt"))
        assert got_p.shape == (n, 13) and got_v.shape == (n, 6)
        assert np.allclose(got_p, est, rtol=2e-5, atol=1e-6), (i, np.abs(got_p - est).max())
        assert np.allclose(got_v, vel, rtol=2e-5, atol=1e-6), (i, np.abs(got_v - vel).max())
    # two processes with a shard each (one per GPU on a node: --device r --shard r G) write the same files
    for rank in range(2):
        sh = [a if a != str(tmp_path / "out") else str(tmp_path / "out_sharded") for a in args] + ["--device", "0", "--shard", str(rank), "2"]
        r = subprocess.run([exe] + sh, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and ("tracked %d objects" % (2 if rank == 0 else 1)) in r.stdout, r.stdout[-800:] + r.stderr[-800:]
    for _, name, _, _ in seqs:
        for f in ("pose_estimate.txt", "velocity_estimate.txt"):
            assert open(str(tmp_path / "out" / name / f)).read() == open(str(tmp_path / "out_sharded" / name / f)).read(), (name, f)
    # the native exchange of a multi-GPU job (--gather, built with rccl.h): every process's result rows through an ncclAllGather,
    # process 0 writes the logs of ALL objects.  One process (a communicator of one rank) always works; two processes on the ONE
    # GPU of this box are what RCCL may refuse ("Duplicate GPU detected"): then only the refusal is checked -- on a node every
    # process has a GPU of its own (--device r --shard r G).
    exe_rccl = str(tmp_path / "ROFT-tracker-batch-rccl")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-DROFT_WITH_RCCL", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I", os.path.join(ROOT, "include", "compat"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "track_many.cpp"), "-o", exe_rccl, "-L", CSRC, "-lroft_hip", "-Wl,-rpath," + CSRC,
                           "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    g1 = [a if a != str(tmp_path / "out") else str(tmp_path / "out_gather1") for a in args] + ["--gather", str(tmp_path / "id1")]
    r = subprocess.run([exe_rccl] + g1, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rows all-gathered over RCCL, logs of 3 objects written" in r.stdout, r.stdout[-800:] + r.stderr[-800:]
    for _, name, _, _ in seqs:
        for f in ("pose_estimate.txt", "velocity_estimate.txt"):
            assert open(str(tmp_path / "out" / name / f)).read() == open(str(tmp_path / "out_gather1" / name / f)).read(), (name, f)
    procs = []
    for rank in range(2):
        g2 = [a if a != str(tmp_path / "out") else str(tmp_path / "out_gather2") for a in args] + ["--device", "0", "--shard", str(rank), "2", "--gather", str(tmp_path / "id2")]
        procs.append(subprocess.Popen([exe_rccl] + g2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    import time as _time
    t_end = _time.time() + 240
    while _time.time() < t_end and any(p_.poll() is None for p_ in procs):
        if any(p_.poll() not in (None, 0) for p_ in procs):      # one process gave up: the other one waits for it in vain
            _time.sleep(2.0)
            break
        _time.sleep(0.2)
    for p_ in procs:
        if p_.poll() is None:
            p_.kill()
    outs = [p_.communicate() for p_ in procs]
    if all(p_.returncode == 0 for p_ in procs):
        assert "logs of 3 objects written" in outs[0][0]
        for _, name, _, _ in seqs:
            for f in ("pose_estimate.txt", "velocity_estimate.txt"):
                assert open(str(tmp_path / "out" / name / f)).read() == open(str(tmp_path / "out_gather2" / name / f)).read(), (name, f)
        print("two processes on one GPU: RCCL all-gather ok")
    else:
        msg = " ".join(o[1] for o in outs)
        assert "ncclCommInitRank" in msg or "ncclAllGather" in msg, msg[-1500:]     # refused by RCCL itself, with its reason
        print("two processes on one GPU refused by RCCL: " + msg.strip().splitlines()[-1][:200])
    # ... and the step after the path: the metrics table of that results tree against the sequences' ground truth
    spec = importlib.util.spec_from_file_location("evaluate_results", os.path.join(ROOT, "tools", "evaluate_results.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    assert ev.main(["--results", str(tmp_path / "out"), "--dataset", str(tmp_path / "dataset"), "--metrics", "rmse_cartesian_3d,rmse_angular,add,adi",
                    "--json", str(tmp_path / "table.json")]) == 0
    table = json.load(open(str(tmp_path / "table.json")))
    lines = capsys.readouterr().out.strip().splitlines()
    assert [ln.split("|")[1].strip() for ln in lines[-4:]] == ["box0", "box1", "box2", "ALL"]
    assert table["rmse_cartesian_3d"]["ALL"] < 3.0 and table["rmse_angular"]["ALL"] < 8.0 and table["adi"]["ALL"] > 80.0 and table["add"]["ALL"] > 60.0
    # the three trajectories differ from each other (the objects were not mixed up)
    a = io.read_log(str(tmp_path / "out" / "box0" / "pose_estimate.txt"))
    b = io.read_log(str(tmp_path / "out" / "box1" / "pose_estimate.txt"))
    assert np.abs(a - b).max() > 1e-2


def test_batched_tracker_front_end_compiles_and_explains_itself(tmp_path):
    from roft_amd import _lib
    _lib.build()
    exe = str(tmp_path / "ROFT-tracker-batch")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include", "compat"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "track_many.cpp"), "-o", exe, "-L", CSRC, "-lroft_hip", "-Wl,-rpath," + CSRC,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "usage: ROFT-tracker-batch --from config.cfg" in r.stderr
    r = subprocess.run([exe, "--log_root", str(tmp_path), "--object", str(tmp_path), "box"], capture_output=True, text=True)
    assert r.returncode == 1 and "--from" in r.stderr


def test_every_facade_header_compiles_on_its_own(tmp_path):
    """Each header of include/ROFT and include/compat is self-contained (a translation unit that includes only it compiles), with
    warnings as errors."""
    units = []
    for sub in ("ROFT", "compat", "compat/BayesFilters", "compat/RobotsIO/Camera", "compat/RobotsIO/Utils", "compat/Eigen"):
        d = os.path.join(ROOT, "include", sub)
        for name in sorted(os.listdir(d)):
            if os.path.isfile(os.path.join(d, name)) and (name.endswith((".h", ".hpp")) or name == "Dense"):
                units.append(os.path.join(sub, name))
    assert len(units) >= 40
    src = tmp_path / "all.cpp"
    # one translation unit per header would be 45 compiler runs; including all of them twice in every order is not needed
    # either: each header is the FIRST include of one unit, compiled in batches
    for k, u in enumerate(units):
        (tmp_path / ("u%d.cpp" % k)).write_text('#include "%s"\nint unit_%d() { return 0; }\n' % (u[len("compat/"):] if u.startswith("compat/") else u, k))
    procs = []
    for k in range(len(units)):
        procs.append(subprocess.Popen(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include", "compat"),
                                       "-I", os.path.join(ROOT, "include"), str(tmp_path / ("u%d.cpp" % k))], stderr=subprocess.PIPE, text=True))
        if len(procs) == 8 or k == len(units) - 1:
            for j, pr in enumerate(procs):
                err = pr.communicate()[1]
                assert pr.returncode == 0, (units[k - len(procs) + 1 + j], err[-1500:])
            procs = []
    del src


KIT_BIN = os.path.join(ROOT, "tests", "cpp", "_kit_build", "replay")


@pytest.mark.gpu
def test_conformance_kit_replays_the_vectors_through_the_class_api():
    """tests/ref_kit/replay.cpp, FACADE build (-DROFT_KIT_FACADE, __graft_entry__.build_conformance_kit): the committed oracle
    vectors replayed through bfl::UKFPrediction over ROFT::CartesianQuaternionModel, ROFT::UKFCorrection over
    ROFT::CartesianQuaternionMeasurement (three measurement types, sigma rotations beyond pi) and ROFT::SKFCorrection over a
    recorded linear model -- the reference's class API, constructor for constructor -- running on the HIP engine.  This pins
    NOTHING (the vectors are the oracle's own): it is the kit's compiler, its I/O check and a third consumer of the vectors; with
    real bfl only the include and link lines change."""
    if not os.path.exists(KIT_BIN):
        pytest.skip("tests/cpp/_kit_build/replay is built by python __graft_entry__.py")
    r = subprocess.run([KIT_BIN, os.path.join(ROOT, "tests", "golden", "oracle_vectors")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert "DIFFERS" not in r.stdout
    # three predictions, seven corrections of the pose filter, two of the velocity filter: mean + covariance each
    assert sum(1 for ln in lines if ln.startswith("ukf_predict_")) == 6
    assert sum(1 for ln in lines if ln.startswith("ukf_correct_")) == 14
    assert sum(1 for ln in lines if ln.startswith("skf_correct")) == 4
    assert "pins nothing" in lines[-1]

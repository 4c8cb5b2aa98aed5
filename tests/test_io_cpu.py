"""File formats either side of the path (roft_amd/io.py): byte-level known answers written from the reference's
reader/writer code, round trips, and -- in the dev container only -- the reference's own mesh file."""
import os
import struct
import zlib

import numpy as np
import pytest

from roft_amd import io, synth


def test_flow_float_byte_layout_and_roundtrip(tmp_path):
    # OpticalFlowUtilities.cpp:99-119: int type | size_t cols | size_t rows | interleaved elements
    f32 = np.arange(2 * 3 * 2, dtype=np.float32).reshape(2, 3, 2) * 0.5
    p = str(tmp_path / "7.float")
    io.save_flow(f32, p)
    raw = open(p, "rb").read()
    assert raw[:4] == struct.pack("=i", 13) and raw[4:20] == struct.pack("=QQ", 3, 2)
    assert raw[20:] == f32.tobytes() and len(raw) == 20 + 48
    ok, back = io.read_flow(p)
    assert ok and back.dtype == np.float32 and np.array_equal(back, f32)
    s16 = (np.arange(4 * 5 * 2, dtype=np.int16).reshape(4, 5, 2) - 17)
    io.save_flow(s16, p)
    assert open(p, "rb").read()[:4] == struct.pack("=i", 11)
    ok, back = io.read_flow(p)
    assert ok and back.dtype == np.int16 and np.array_equal(back, s16)
    assert io.flow_format(s16, 20) == (11, 4, 32.0) and io.flow_format(f32, 3) == (13, 1, 1.0)
    # missing / truncated files are "not available", not errors (read_flow returns (false, Mat()))
    assert io.read_flow(str(tmp_path / "nope.float"))[0] is False
    open(p, "wb").write(raw[:30])
    assert io.read_flow(p)[0] is False
    with pytest.raises(ValueError):
        io.save_flow(np.zeros((2, 2, 2), np.float64), p)


def test_depth_float(tmp_path):
    d = np.random.default_rng(0).random((5, 7)).astype(np.float32)
    p = str(tmp_path / "0.float")
    io.write_depth(p, d)
    raw = open(p, "rb").read()
    assert raw[:16] == struct.pack("=QQ", 7, 5) and raw[16:] == d.tobytes()      # ho3d_utils.py:74-79
    assert np.array_equal(io.read_depth(p), d)


def test_poses_txt_and_logs(tmp_path):
    p = str(tmp_path / "poses.txt")
    open(p, "w").write("0.1 0.2 0.7 0 0 1 1.5707963267948966\n0 0 0 0 0 0 0\n")
    pose, valid = io.read_poses(p)
    assert list(valid) == [True, False] and len(pose) == 2
    assert np.allclose(pose[0], [0.1, 0.2, 0.7, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    open(p, "w").write("header\nNaN NaN NaN NaN NaN NaN 0.3 -0.1 0.9 1 0 0 0.2\n")
    pose, valid = io.read_poses(p, skip_rows=1, skip_cols=6)  # prediction rows carry 6 velocity columns first
    assert valid[0] and np.allclose(pose[0, :3], [0.3, -0.1, 0.9]) and np.allclose(pose[0, 3:], [np.cos(0.1), np.sin(0.1), 0, 0])
    # log rows: v w x axis angle
    q = io.axis_angle_to_quat([0, 1, 0], 0.4)
    io.write_estimate_logs(str(tmp_path) + "/", np.concatenate([np.zeros(6), [1, 2, 3], q])[None], np.arange(6.0)[None])
    row = np.loadtxt(str(tmp_path / "pose_estimate"))
    assert np.allclose(row, [0] * 6 + [1, 2, 3, 0, 1, 0, 0.4])
    assert np.allclose(np.loadtxt(str(tmp_path / "velocity_estimate")), np.arange(6.0))
    axis, angle = io.quat_to_axis_angle(-q)                    # w < 0: Eigen flips the axis, angle stays in [0, pi]
    assert np.allclose(axis, [0, 1, 0]) and abs(angle - 0.4) < 1e-12
    assert io.quat_to_axis_angle([1, 0, 0, 0])[0].tolist() == [1.0, 0.0, 0.0]


def test_delivery_schedule_matches_generator():
    st = synth.make_stream(3, 20, synth.Camera.shape_a().scaled(8))
    assert list(io.delivery_schedule(20)) == list(st.mask_delivery)
    assert list(io.delivery_schedule(8, 30.0, 10.0)) == [0, -1, -1, 0, -1, -1, 3, -1]
    assert list(io.delivery_schedule(4, simulate_inference_time=False, desired_fps=15.0)) == [0, -1, 2, -1]


def test_obj_loader(tmp_path):
    p = str(tmp_path / "m.obj")
    open(p, "w").write("# mesh\nv 0 0 0 0.5 0.5 0.5\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 1/1/1 3/2/1 4/3/1\nf 1 2 3 4\nf -4 -3 -2\n")
    v, t = io.load_obj(p)
    assert v.shape == (4, 3) and v.dtype == np.float32
    assert t.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 2, 3], [0, 1, 2]]


def test_obj_loader_on_the_reference_mesh():
    ref = "/root/reference/src/roft-lib/meshes/DOPE/003_cracker_box.obj"
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present (GPU box)")
    v, t = io.load_obj(ref)
    assert 7000 < len(v) < 9000 and 15000 < len(t) < 17000
    assert np.allclose(v.min(0), [-0.080, -0.102, -0.036], atol=2e-3) and np.allclose(v.max(0), [0.084, 0.112, 0.035], atol=2e-3)
    assert t.min() == 0 and t.max() == len(v) - 1


def _png(img, filters):
    h = img.shape[0]
    ch = 1 if img.ndim == 2 else img.shape[2]
    rows = img.reshape(h, -1).astype(np.int32)
    out = bytearray()
    prev = np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        ft = filters[y % len(filters)]
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
            pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
            pr = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        out.append(ft)
        out += bytes(((cur - pr) & 255).astype(np.uint8))
        prev = cur
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))

    comp = zlib.compress(bytes(out))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", img.shape[1], h, 8, ctype, 0, 0, 0)) +
            chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b""))


def test_png_decoder_all_filters(tmp_path):
    rng = np.random.default_rng(2)
    gray = (rng.random((13, 17)) < 0.4).astype(np.uint8) * 255
    rgb = rng.integers(0, 256, size=(9, 11, 3), dtype=np.uint8)
    for img in (gray, rgb, rng.integers(0, 256, size=(6, 5, 4), dtype=np.uint8)):
        for filters in ([0], [1], [2], [3], [4], [0, 1, 2, 3, 4]):
            p = str(tmp_path / "m.png")
            open(p, "wb").write(_png(img, filters))
            assert np.array_equal(io.read_png(p), img), filters
    p = str(tmp_path / "c.png")
    open(p, "wb").write(_png(rgb, [4]))
    m = io.read_mask_png(p)
    assert m.shape == rgb.shape[:2] and m.dtype == np.uint8
    with pytest.raises(ValueError):
        open(p, "wb").write(b"not a png")
        io.read_png(p)


def test_sequence_reader_feeds_the_engine_format(tmp_path):
    """Write a tiny Fast-YCB style directory from a synthetic stream and read it back frame by frame."""
    st = synth.make_stream(5, 8, synth.Camera.shape_a().scaled(8), flow_invalid=0.0)
    root = tmp_path
    for d in ("optical_flow/nvof", "masks/gt", "depth", "dope"):
        os.makedirs(root / d)
    with open(root / "data.txt", "w") as f:
        for k in range(8):
            f.write("%f %f 0 0 0 1 0 0 0\n" % (k / 30.0, k / 30.0))
    with open(root / "dope" / "poses.txt", "w") as f:
        for k in range(8):
            axis, angle = io.quat_to_axis_angle(st.gt.q[k])
            f.write(" ".join("%.17g" % v for v in list(st.gt.x[k]) + list(axis) + [angle]) + "\n")
    for k in range(8):
        io.write_depth(str(root / "depth" / ("%d.float" % k)), st.depth[k].numpy())
        if k > 0:
            io.save_flow(st.flow[k].numpy(), str(root / "optical_flow" / "nvof" / ("%d.float" % k)))
        open(root / "masks" / "gt" / ("obj_%d.png" % k), "wb").write(_png(st.mask_gt[k].numpy(), [0, 2]))
    seq = io.Sequence(str(root), "obj", flow_set="nvof", mask_set="gt", pose_set="dope", width=st.camera.width,
                      height=st.camera.height)
    assert len(seq) == 8
    f0, f3, f6 = seq.frame(0), seq.frame(3), seq.frame(6)
    assert f0["flow"] is None and f0["mask"] is not None and f0["pose"] is not None    # frame 0: no flow file
    assert f3["mask"] is None and f3["pose"] is None and np.array_equal(f3["flow"], st.flow[3].numpy())
    assert np.array_equal(f6["mask"], st.mask_gt[0].numpy()) and np.array_equal(f6["depth"], st.depth[6].numpy())
    assert np.allclose(f6["pose"][0], st.gt.x[0]) and abs(abs(np.dot(f6["pose"][1], st.gt.q[0])) - 1) < 1e-12
    assert abs(f6["dt"] - 1 / 30.0) < 1e-6


def test_png_writer_and_gray_conversion(tmp_path):
    rng = np.random.default_rng(3)
    for shape in ((7, 9), (5, 6, 3), (4, 4, 4)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        p = str(tmp_path / "w.png")
        io.write_png(p, img)
        assert np.array_equal(io.read_png(p), img)
    # cv::cvtColor BGR2GRAY fixed-point known answers
    rgb = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 77]]], np.uint8)
    assert io.rgb_to_gray(rgb).tolist() == [[255, 0, 76, 150, 29, 129]]
    g = rng.integers(0, 256, (3, 3), dtype=np.uint8)
    assert io.rgb_to_gray(g) is g


def test_flow_dumper_argument_errors(capsys):
    import importlib.util
    spec = importlib.util.spec_from_file_location("flow_dumper", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "flow_dumper.py"))
    fd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fd)
    assert fd.main(["x"]) == 1
    assert "Synopsis: ROFT-of-dumper" in capsys.readouterr().err
    base = ["x", "/nonexistent", "txt", "png", "6", "0", "640", "480"]
    assert fd.main(base + ["nvof3", "/tmp/out"]) == 1
    assert 'Invalid <nvof_version> "nvof3"' in capsys.readouterr().err
    bad = list(base)
    bad[4] = "six"
    assert fd.main(bad + ["nvof1", "/tmp/out"]) == 1
    assert "Invalid value six for parameter <heading_zeros>." in capsys.readouterr().err


def test_log_files_as_the_references_evaluation_reads_them(tmp_path):
    """tests/golden/log_fixtures.json: the five log files written by the facade's bfl::Logger stand-in, and the arrays the
    reference's evaluation/data_loader.py (imported in the dev container by make_log_fixtures.py) parses out of them.  The
    writer must still produce those texts byte for byte, and roft_amd.io.read_log must read what the reference reads."""
    import json
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "log_fixtures.json")))
    exe = str(tmp_path / "log_writer")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "golden", "log_writer.cpp"), "-o", exe])
    subprocess.check_call([exe, str(tmp_path)])
    ours = fx.pop("load_ours")
    for name, want in fx.items():
        path = str(tmp_path / (name + ".txt"))
        assert open(path).read() == want["text"], name
        got = io.read_log(path, skip_cols=6 if name == "pose_estimate" else 0)
        assert np.array_equal(got, np.array(want["parsed"])), name
    assert np.array(fx["pose_estimate"]["parsed"]).shape == (4, 7) and np.array(fx["execution_times"]["parsed"]).shape == (4, 2)
    # the whole results tree through the reference's DataLoader("ours").load(): the five contents under the names and the
    # layout it expects, cam_K.json as roft_amd.io writes it
    from roft_amd import synth
    io.write_cam_k(str(tmp_path / "cam_K.json"), synth.Camera.shape_b())
    assert open(str(tmp_path / "cam_K.json")).read() == ours["cam_k_json"]
    cam = synth.Camera.shape_b()
    assert ours["cam_intrinsics"] == dict(fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy)
    assert ours["shapes"] == dict(time=[4, 2], pose=[4, 7], velocity=[4, 6], pose_meas=[4, 7], vel_meas=[4, 6])
    assert np.array_equal(np.array(ours["pose"]), np.array(fx["pose_estimate"]["parsed"]))


def test_dataset_scripts_of_the_reference(tmp_path):
    """tests/golden/script_fixtures.json holds what the reference's own scripts printed / wrote when run (in the dev container)
    on files written by roft_amd.io: the initial-pose finder of test/test_ho3d.sh:68 and the data.txt generator."""
    import json
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "script_fixtures.json")))
    for i, c in enumerate(fx["pose_finder"]):
        p = str(tmp_path / ("p%d.txt" % i))
        open(p, "w").write(c["poses_txt"])
        got = io.find_initial_pose(p, c["fps"])
        want = c["stdout"].strip()
        assert (want == "" and got is None) or (got is not None and "%d %s" % got == want), (i, got, want)
        # the rows the finder skipped are the ones read_poses reports as invalid detections
        pose, ok = io.read_poses(p)
        if got is not None:
            row = got[0] - 6 if got[0] else 0
            assert ok[row] and not ok[:row][::6].any()
    assert any(c["stdout"].strip() == "" for c in fx["pose_finder"]) and any(c["stdout"].startswith("0 ") for c in fx["pose_finder"])
    for c in fx["data_txt"]:
        p = str(tmp_path / ("d%d.txt" % c["frames"]))
        io.write_data_txt(p, c["frames"], 30.0)
        assert open(p).read() == c["text"]
        rgb, depth, cam = io.read_data_txt(p)
        assert len(rgb) == c["frames"] and np.array_equal(rgb, depth) and np.all(cam[:, 3] == 1.0)

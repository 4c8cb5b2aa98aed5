"""C++ facade (include/ROFT/Filters.h): compiles against the C ABI, fails loudly without a device,
and on the GPU returns what the operator-level ABI returns."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "roft_amd", "csrc")


def build_exe(tmp_path):
    from roft_amd import _lib
    _lib.build()
    exe = str(tmp_path / "facade_check")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "facade_check.cpp"), "-o", exe, "-L", CSRC, "-lroft_hip",
                           "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
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
    path = str(tmp_path / "in.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("i", n))
        for a in (x, P, q, y, H, pm, pP):
            f.write(np.ascontiguousarray(a, np.float64).tobytes())
    return path, (x, P, q, y, H, pm, pP)


def test_facade_compiles_and_fails_loudly_without_device(tmp_path):
    from roft_amd import _lib
    exe = build_exe(tmp_path)
    if _lib.lib().roft_device_count() > 0:
        pytest.skip("a HIP device is present")
    inp, _ = problem(tmp_path)
    r = subprocess.run([exe, inp, str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout


@pytest.mark.gpu
def test_facade_matches_operator_abi(tmp_path):
    from roft_amd import ops
    exe = build_exe(tmp_path)
    inp, (x, P, q, y, H, pm, pP) = problem(tmp_path)
    out = str(tmp_path / "out.bin")
    subprocess.check_call([exe, inp, out])
    raw = open(out, "rb").read()
    got = np.frombuffer(raw[:8 * (6 + 36 + 13 + 144)], np.float64)
    flow = np.frombuffer(raw[8 * (6 + 36 + 13 + 144):], np.float32).reshape(48, 64, 2)
    xp, Pp = ops.kf_predict(x, P, q)
    rc, xc, Pc = ops.skf_correct(xp, Pp, y, H, (1.0, 1.0), True)
    Q = ops.process_noise([1.0] * 3, [1.0] * 3, 1.0 / 30.0)
    m1, P1 = ops.ukf_predict(pm, pP, Q, 1.0 / 30.0)
    assert np.array_equal(got[:6], xc) and np.array_equal(got[6:42].reshape(6, 6), Pc)
    assert np.array_equal(got[42:55], m1) and np.array_equal(got[55:].reshape(12, 12), P1)
    # optical-flow source facade == operator ABI on the same pattern, and it sees the (2, 1) shift
    yy, xx = np.mgrid[0:48, 0:64]
    g0 = (128 + (60.0 * np.sin(0.35 * xx) * np.cos(0.27 * yy)).astype(np.int64)).astype(np.uint8)
    g1 = (128 + (60.0 * np.sin(0.35 * (xx - 2)) * np.cos(0.27 * (yy - 1))).astype(np.int64)).astype(np.uint8)
    assert np.array_equal(flow, ops.optical_flow(g0, g1, levels=2, det_min=1.0))
    inner = flow[12:36, 16:48]
    assert abs(np.median(inner[..., 0]) - 2.0) < 0.3 and abs(np.median(inner[..., 1]) - 1.0) < 0.3


def build_filter_exe(tmp_path):
    from roft_amd import _lib
    _lib.build()
    exe = str(tmp_path / "filter_check")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "filter_check.cpp"), "-o", exe, "-L", CSRC, "-lroft_hip",
                           "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def dump_stream(path, st, n):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
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


def test_whole_filter_facade_compiles_and_fails_loudly_without_device(tmp_path):
    from roft_amd import _lib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    exe = build_filter_exe(tmp_path)
    if _lib.lib().roft_device_count() > 0:
        pytest.skip("a HIP device is present")
    st = util.stream(33, 3, 4)
    dump_stream(str(tmp_path / "s.bin"), st, 3)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout


@pytest.mark.gpu
def test_whole_filter_facade_equals_the_python_engine(tmp_path):
    """ROFT::ROFTFilter (C++ facade) fed frame by frame == roft_amd.engine.ROFTFilterBatch on the same stream, and both
    == the oracle."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    from oracle import binding as ob
    from test_engine_gpu import make_engine
    n = 20
    st = util.stream(34, n, 2)
    dump_stream(str(tmp_path / "s.bin"), st, n)
    exe = build_filter_exe(tmp_path)
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

"""GPU parity of the batched engine (roft_engine_* C ABI) against the oracle's ROFTFilter restatement
on identical synthetic Fast-YCB-shaped streams: same flow point sets, same propagated masks, same
outlier decisions, trajectories within the stated tolerance."""
import numpy as np
import pytest

from roft_amd import _lib as L
from roft_amd import engine as E
from roft_amd import synth

import util

pytestmark = pytest.mark.gpu

POS_TOL = 1e-6    # m
ROT_TOL = 1e-6    # rad
TWIST_TOL = 1e-6  # m/s, rad/s
# likelihoods of the outlier test inside the engine: the alternatives' poses differ from the oracle's by ~1e-11, which
# moves a render depth by an ulp here and there (the kernel itself is bit exact on identical poses:
# test_parity_gpu.py::test_outlier_test_hot_path_kernel)
LIK_ENGINE_RTOL = 1e-9


def make_engine(streams, **over):
    st0 = streams[0]
    cfg = E.default_config(st0.camera.width, st0.camera.height, st0.flow_type, max_objects=len(streams))
    c = st0.camera
    cfg.cam.fx, cfg.cam.fy, cfg.cam.cx, cfg.cam.cy = c.fx, c.fy, c.cx, c.cy
    cfg.flow_grid, cfg.flow_scale = st0.flow_grid, st0.flow_scale
    for k, v in over.items():
        setattr(cfg, k, v)
    eng = E.ROFTFilterBatch(cfg)
    for st in streams:
        d = E.default_object()
        m0 = synth.initial_pose_from_stream(st)
        for i in range(13):
            d.p_mean0[i] = m0[i]
        eng.add_object(d, *st.mesh)
    return eng


def rot_err(qa, qb):
    return 2.0 * np.arccos(min(1.0, abs(float(np.dot(qa, qb)))))


def compare(streams, n_frames, check_masks=True, **over):
    from oracle import binding as ob
    ref = [util.run_oracle_tracker(ob, st, n_frames, **over) for st in streams]
    eng = make_engine(streams, **over)
    n_tests = 0
    for k in range(n_frames):
        frames = []
        for st in streams:
            depth, flow, mask, pose = util.frame_inputs(st, k)
            frames.append(dict(depth=depth, flow=flow, mask=mask, pose=pose, dt=st.dt))
        eng.submit(frames)
        eng.step()
        outs = eng.outputs()
        for o, r in enumerate(ref):
            got = outs[o]
            exp = r[k]
            assert got.n_flow_points == exp["n"], (k, o)
            assert got.outlier_selected == exp["sel"], (k, o, list(got.outlier_L), exp["L"])
            if exp["sel"] >= 0:
                n_tests += 1
                np.testing.assert_allclose(np.array(got.outlier_L), exp["L"], rtol=LIK_ENGINE_RTOL)
            pose = np.array(got.pose)
            np.testing.assert_allclose(pose[:9], exp["pose"][:9], rtol=0, atol=POS_TOL, err_msg="frame %d obj %d" % (k, o))
            assert rot_err(pose[9:], exp["pose"][9:]) < ROT_TOL, (k, o)
            np.testing.assert_allclose(np.array(got.twist), exp["twist"], rtol=0, atol=TWIST_TOL)
            if check_masks:
                assert np.array_equal(eng.mask(o), exp["mask"]), (k, o)
    eng.close()
    return n_tests


def test_engine_matches_oracle_default_config():
    streams = [util.stream(100 + i, 20, scale=2) for i in range(3)]
    n_tests = compare(streams, 20)
    assert n_tests >= 6   # outlier rejection ran at every pose arrival


def test_engine_matches_oracle_s16_flow():
    streams = [util.stream(200 + i, 14, scale=2, flow_type=synth.FLOW_S16C2) for i in range(2)]
    compare(streams, 14)


@pytest.mark.parametrize("over", [
    dict(use_pose_resync=0),
    dict(outlier_rejection=0),
    dict(use_pose_resync=0, outlier_rejection=0),
    dict(use_pose=0, use_pose_resync=0, outlier_rejection=0),
    dict(flow_weighting=0),
    dict(flow_aided_segmentation=0),
])
def test_engine_matches_oracle_ablations(over):
    """The ablation matrix of the reference's test/test.sh:62-118."""
    streams = [util.stream(300, 14, scale=2)]
    compare(streams, 14, **over)


def test_engine_full_resolution_batch():
    """640x480, the BASELINE metric shape, device-resident inputs (zero copy)."""
    import torch
    streams = [util.stream(400 + i, 8, scale=1, mesh_n=36) for i in range(2)]
    from oracle import binding as ob
    ref = [util.run_oracle_tracker(ob, st, 8) for st in streams]
    eng = make_engine(streams)
    dev = [dict(depth=st.depth.cuda(), flow=st.flow.cuda(), mask=st.mask_gt.cuda()) for st in streams]
    for k in range(8):
        frames = []
        for st, d in zip(streams, dev):
            _, _, _, pose = util.frame_inputs(st, k)
            mi = st.mask_delivery[k]
            frames.append(dict(depth=d["depth"][k].data_ptr(), flow=d["flow"][k].data_ptr() if st.flow_valid[k] else None,
                               mask=d["mask"][mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt,
                               mem_kind=L.MEM_DEVICE))
        eng.submit(frames)
        eng.step()
        outs = eng.outputs()
        for o, r in enumerate(ref):
            assert outs[o].n_flow_points == r[k]["n"]
            assert outs[o].outlier_selected == r[k]["sel"]
            np.testing.assert_allclose(np.array(outs[o].pose)[:9], r[k]["pose"][:9], rtol=0, atol=POS_TOL)
            assert np.array_equal(eng.mask(o), r[k]["mask"])
    eng.close()


def test_engine_rejects_bad_calls():
    cfg = E.default_config(320, 240)
    eng = E.ROFTFilterBatch(cfg)
    with pytest.raises(L.RoftError):
        eng.step()                      # nothing submitted
    with pytest.raises(L.RoftError):
        E.ROFTFilterBatch(E.default_config(333, 240))   # width not a multiple of 32
    eng.close()


def test_prediction_cholesky_guard_changes_nothing_measurable():
    """roft_config::ukf_cholesky_guard: drawing the prediction's sigma points from the Cholesky factor (default,
    while the rotational variances are small) or from the eigen-decomposition (guard = 0) gives the same
    trajectories to ~1e-12, and both match the oracle."""
    from oracle import binding as ob
    st = util.stream(31, 60, 2)
    ref = util.run_oracle_tracker(ob, st, 60)
    out = {}
    for guard in (2e-4, 0.0):
        eng = make_engine([st], ukf_cholesky_guard=guard)   # (the correction's guard is void when this one is 0)
        traj = []
        for k in range(60):
            depth, flow, mask, pose = util.frame_inputs(st, k)
            eng.submit([dict(depth=depth, flow=flow, mask=mask, pose=pose, dt=st.dt)])
            eng.step()
            traj.append(eng.state(0)[0])
        eng.close()
        out[guard] = np.array(traj)
        assert np.abs(out[guard] - np.array([r["pose"] for r in ref])).max() < 1e-8
    assert 0.0 < np.abs(out[2e-4] - out[0.0]).max() < 1e-10


def test_square_root_switches_between_cholesky_and_eigen_within_a_run():
    """A run that starts with a large pose covariance (var(theta) = 0.05 >> the guard 2e-4) and large angular
    process noise: the prediction and correction square roots fall back to the eigen-decomposition at first and switch to
    the Cholesky factor as the filter converges.  The trajectory follows the oracle (always eigen) throughout."""
    import ctypes as C
    from oracle import binding as ob
    n = 40
    st = util.stream(41, n, 2)
    cfg = util.oracle_config(ob, st)
    for i in range(12):
        cfg.p_cov0_diag[i] = 5e-2
    verts, tris = st.mesh
    trk = ob.Tracker(cfg, verts, tris)
    ref = []
    for k in range(n):
        depth, flow, mask, pose = util.frame_inputs(st, k)
        r = trk.step(st.dt, depth, flow, mask, pose)
        ref.append((np.array(r.pose), r.outlier_selected, r.n_flow_points, np.array(r.pose_cov).reshape(12, 12)))
    trk.close()
    c = st.camera
    ecfg = E.default_config(c.width, c.height, st.flow_type, max_objects=1)
    ecfg.cam.fx, ecfg.cam.fy, ecfg.cam.cx, ecfg.cam.cy = c.fx, c.fy, c.cx, c.cy
    eng = E.ROFTFilterBatch(ecfg)
    d = E.default_object()
    m0 = synth.initial_pose_from_stream(st)
    for i in range(13):
        d.p_mean0[i] = m0[i]
    for i in range(12):
        d.p_cov0_diag[i] = 5e-2
    eng.add_object(d, verts, tris)
    var_theta = []
    for k in range(n):
        depth, flow, mask, pose = util.frame_inputs(st, k)
        eng.submit([dict(depth=depth, flow=flow, mask=mask, pose=pose, dt=st.dt)])
        eng.step()
        p, P, _, _ = eng.state(0)
        o = eng.outputs()[0]
        assert np.abs(p - ref[k][0]).max() < 1e-7, k
        assert np.abs(P - ref[k][3]).max() <= 1e-6 * np.abs(ref[k][3]).max(), k
        assert o.outlier_selected == ref[k][1] and o.n_flow_points == ref[k][2], k
        var_theta.append(P[9:, 9:].diagonal().max())
    eng.close()
    # frame 0 predicts and corrects from var(theta) = 5e-2 (eigen path); the pose measurement of that frame brings it
    # below the guard, so every later step draws from the Cholesky factor
    assert 5e-2 > 2e-4 > max(var_theta)

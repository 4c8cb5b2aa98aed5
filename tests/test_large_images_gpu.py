"""Images larger than one CU's LDS holds as a bit plane (1920 x 1080: 259 KB per plane; rounds 1 - 4 refused everything beyond
~1.1 Mpixel).  The reference has no size limit -- ImageOpticalFlowMeasurement<T>::freeze scans any cv::Mat
(include/ROFT/ImageOpticalFlowMeasurement.hpp:231-256), so does ImageSegmentationOFAidedSource::map (hpp:234-281) --: the flow
measurement, the mask propagation (binary and three-valued), the feature buffering and the whole engine at 1920 x 1080, both
flow types, against the oracle; bit exact where the small images are."""
import numpy as np
import pytest

from roft_amd import _lib as L
from roft_amd import engine as E
from roft_amd import ops, synth

import util

pytestmark = pytest.mark.gpu

W, H = 1920, 1080


def full_hd():
    return synth.Camera(W, H, 1400.0, 1400.0, W / 2.0, H / 2.0)


@pytest.mark.parametrize("flow_type", [synth.FLOW_F32C2, synth.FLOW_S16C2])
def test_flow_measurement_and_mask_propagation_at_1920x1080(oracle, flow_type):
    cam = full_hd()
    st = synth.make_stream(77, 4, cam, flow_type=flow_type, mesh_n=6, device="cuda")
    st.depth, st.flow, st.mask_gt = st.depth.cpu(), st.flow.cpu(), st.mask_gt.cpu()
    ocam = util.oracle_camera(oracle, cam)
    lcam = L.Camera(W, H, cam.fx, cam.fy, cam.cx, cam.cy)
    rng = np.random.default_rng(5)
    for k in (1, 3):
        mask, depth, flow = st.mask_gt[k - 1].numpy(), st.depth[k - 1].numpy(), st.flow[k].numpy()
        for m, radius in ((mask, 35.0), (mask, 1.0), ((rng.random((H, W)) < 0.25).astype(np.uint8) * 255, 35.0)):
            n0, uv0, y0, H0 = oracle.flow_measurement(ocam, m, depth, flow, st.dt, radius=radius)
            n1, uv1, y1, H1 = ops.flow_measurement(lcam, m, depth, flow, st.dt, radius=radius)
            assert n0 > 100 and n1 == n0
            assert np.array_equal(uv0, uv1) and np.array_equal(y0, y1) and np.array_equal(H0, H1)
    # mask propagation through 1 and 3 flows: binary mask, and a three-valued one (the map-based general path, listed in pieces)
    flows = [st.flow[k].numpy() for k in (1, 2, 3)]
    m255 = st.mask_gt[0].numpy()
    m3 = m255.copy()
    vs, us = np.nonzero(m3)
    m3[vs[::3], us[::3]] = 1
    for m in (m255, m3):
        for fl in (flows[:1], flows):
            want = oracle.mask_propagate(m, fl)
            got = ops.mask_propagate(m, fl)
            assert got.any() and np.array_equal(want, got)


@pytest.mark.parametrize("flow_type", [synth.FLOW_F32C2, synth.FLOW_S16C2])
def test_engine_tracks_at_1920x1080_like_the_oracle(oracle, flow_type):
    """One object, 14 frames through two pose arrivals (re-sync replay, outlier test on features buffered at full size), single
    frames and a batch of 6: flow point counts, outlier decisions and the final mask equal the oracle's, poses to 1e-6."""
    n = 14
    cam = full_hd()
    st = synth.make_stream(78, n, cam, flow_type=flow_type, mesh_n=8, device="cuda")
    host = util.to_device(st)   # (device copies for the engine)
    st.depth, st.flow, st.mask_gt = st.depth.cpu(), st.flow.cpu(), st.mask_gt.cpu()
    ref = util.run_oracle_tracker(oracle, st, n)
    for T in (1, 6):
        cfg = E.default_config(W, H, flow_type, max_objects=1, max_batch_frames=T)
        cfg.cam.fx, cfg.cam.fy, cfg.cam.cx, cfg.cam.cy = cam.fx, cam.fy, cam.cx, cam.cy
        cfg.flow_grid, cfg.flow_scale = st.flow_grid, st.flow_scale
        eng = E.ROFTFilterBatch(cfg)
        d = E.default_object()
        m0 = synth.initial_pose_from_stream(st)
        for i in range(13):
            d.p_mean0[i] = m0[i]
        eng.add_object(d, *st.mesh)
        eng.enable_log(n)
        k = 0
        while k < n:
            t = min(T, n - k)
            eng.submit_batch([[util.device_frame(host, k + j)] for j in range(t)])
            eng.step()
            k += t
        pose, twist, npts, sel = eng.get_log(0, n)
        mask = eng.mask(0)
        eng.close()
        assert np.array_equal(npts[:, 0], np.array([r["n"] for r in ref])), T
        assert np.array_equal(sel[:, 0], np.array([r["sel"] for r in ref])), T
        assert (sel[:, 0] >= 0).sum() >= 1
        assert np.abs(pose[:, 0] - np.array([r["pose"] for r in ref])).max() < 1e-6, T
        assert np.array_equal(mask, ref[n - 1]["mask"]), T

"""Optical-flow producer: HIP kernels vs the C restatement (oracle/ro_opticalflow.c), the
product formats of ImageOpticalFlowNVOF.cpp:19-80, accuracy against the analytic flow of the synthetic stream, and
the tracker fed by the produced flow."""
import numpy as np
import pytest
import torch

from roft_amd import _lib as L
from roft_amd import ops, synth

import util
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def gray_stream(seed, n, scale=2, **kw):
    st = util.stream(seed, n, scale, with_gray=True, device="cuda", **kw)
    return st, st.gray.cpu().numpy()


@pytest.mark.parametrize("levels,radius,iterations", [(3, 3, 3), (1, 2, 2), (4, 4, 1), (2, 1, 5)])
def test_dense_field_matches_the_oracle(levels, radius, iterations):
    st, gray = gray_stream(71, 4)
    for k in (1, 3):
        ref = ob.optical_flow(gray[k - 1], gray[k], levels, radius, iterations, 100.0)
        got = ops.optical_flow(gray[k - 1], gray[k], levels=levels, radius=radius, iterations=iterations, det_min=100.0)
        # same float expression tree on both sides (no contraction): bit-exact
        assert np.array_equal(ref, got), float(np.abs(ref - got).max())


def test_noise_images_match_the_oracle():
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    b = np.roll(a, (1, 2), (0, 1))
    ref = ob.optical_flow(a, b, 2, 2, 3, 1.0)
    got = ops.optical_flow(a, b, levels=2, radius=2, iterations=3, det_min=1.0)
    assert np.array_equal(ref, got)


def test_s16_grid4_product():
    st, gray = gray_stream(72, 3)
    ref = ob.flow_to_s16_grid4(ob.optical_flow(gray[1], gray[2]))
    got = ops.optical_flow(gray[1], gray[2], flow_type=L.FLOW_S16C2)
    assert got.dtype == np.int16 and got.shape == (st.camera.height // 4, st.camera.width // 4, 2)
    assert np.array_equal(ref, got)


def test_accuracy_against_analytic_flow():
    st, gray = gray_stream(73, 6, scale=1)
    flow_gt = st.flow.cpu().numpy()
    masks = st.mask_gt.cpu().numpy()
    for k in (2, 5):
        got = ops.optical_flow(gray[k - 1], gray[k])
        # interior of the object in frame k-1 (the pixels the velocity stage samples)
        m = torch.from_numpy(masks[k - 1].astype(np.float32))[None, None]
        inner = (-torch.nn.functional.max_pool2d(-m, 9, 1, 4))[0, 0].numpy() > 0
        inner &= np.isfinite(flow_gt[k]).all(axis=2) & (np.abs(np.nan_to_num(flow_gt[k])) < 1e3).all(axis=2)   # invalid markers (NaN, 1e10) of the stream
        epe = np.linalg.norm(got - flow_gt[k], axis=2)
        assert inner.sum() > 500
        assert epe[inner].mean() < 0.15 and np.percentile(epe[inner], 95) < 0.4
        near = torch.nn.functional.max_pool2d(m, 21, 1, 10)[0, 0].numpy() > 0
        far = torch.nn.functional.max_pool2d(m, 65, 1, 32)[0, 0].numpy() > 0
        assert np.abs(got[~near]).max() < 0.01        # the coarse levels leak a little around the silhouette
        assert (~far).any() and np.abs(got[~far]).max() == 0.0   # static background: zero flow


def test_batched_device_producer_equals_single_calls():
    st, gray = gray_stream(74, 5)
    H, W = gray.shape[1:]
    g = torch.from_numpy(gray).cuda()
    n = 4
    for ft, shape, dt in ((L.FLOW_F32C2, (n, H, W, 2), torch.float32), (L.FLOW_S16C2, (n, H // 4, W // 4, 2), torch.int16)):
        out = torch.zeros(shape, dtype=dt, device="cuda")
        torch.cuda.synchronize()
        fp = ops.FlowProducer(W, H, n, ft)
        fp.run([g[k].data_ptr() for k in range(n)], [g[k + 1].data_ptr() for k in range(n)], [out[k].data_ptr() for k in range(n)])
        fp.sync()
        # second run on the same workspaces gives the same result
        out2 = torch.zeros_like(out)
        torch.cuda.synchronize()
        fp.run([g[k].data_ptr() for k in range(n)], [g[k + 1].data_ptr() for k in range(n)], [out2[k].data_ptr() for k in range(n)])
        fp.sync()
        fp.close()
        assert torch.equal(out, out2)
        for k in range(n):
            single = ops.optical_flow(gray[k], gray[k + 1], flow_type=ft)
            assert np.array_equal(single, out[k].cpu().numpy())


def test_argument_errors():
    a = np.zeros((60, 80), np.uint8)
    with pytest.raises(L.RoftError):
        ops.optical_flow(a, a, levels=4)           # 80 is not a multiple of 4 * 8
    with pytest.raises(L.RoftError):
        ops.optical_flow(a, a, levels=0)
    with pytest.raises(L.RoftError):
        ops.optical_flow(a, a, radius=9)
    with pytest.raises(ValueError):
        ops.optical_flow(a, a[:30])


def test_tracker_fed_by_the_produced_flow():
    """End to end: gray images -> HIP flow producer -> engine; tracks like with the analytic flow."""
    from test_engine_gpu import make_engine
    st, gray = gray_stream(75, 48, scale=1)
    H, W = gray.shape[1:]
    g = torch.from_numpy(gray).cuda()
    n = st.n_frames
    flow = torch.zeros((n, H, W, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    fp = ops.FlowProducer(W, H, n - 1)
    fp.run([g[k - 1].data_ptr() for k in range(1, n)], [g[k].data_ptr() for k in range(1, n)], [flow[k].data_ptr() for k in range(1, n)])
    fp.sync()
    fp.close()
    errs = {}
    for name, fl in (("analytic", st.flow.cuda()), ("produced", flow)):
        eng = make_engine([st])
        depth, masks = st.depth.cuda(), st.mask_gt.cuda()
        torch.cuda.synchronize()
        for k in range(n):
            mi = st.mask_delivery[k]
            pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
            eng.submit([dict(depth=depth[k].data_ptr(), flow=fl[k].data_ptr() if st.flow_valid[k] else None,
                             mask=masks[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE)])
            eng.step()
        eng.sync()
        pose, _, _, _ = eng.state(0)
        errs[name] = float(np.linalg.norm(pose[6:9] - st.gt.x[n - 1]))
        eng.close()
    assert errs["produced"] < 0.02, errs
    assert errs["produced"] < errs["analytic"] + 0.01, errs


def test_flow_dumper_writes_the_reference_file_layout(tmp_path):
    """tools/flow_dumper.py (the ROFT-of-dumper counterpart): RGB PNG frames in, `<index>.float` files out."""
    import importlib.util
    import os

    from roft_amd import io
    st, gray = gray_stream(76, 5)
    H, W = gray.shape[1:]
    root = tmp_path / "seq"
    (root / "rgb").mkdir(parents=True)
    rgbs = []
    for k in range(5):
        rgb = np.stack([gray[k], np.roll(gray[k], 1, 0), 255 - gray[k]], -1)   # three different channels
        rgbs.append(io.rgb_to_gray(rgb))
        io.write_png(str(root / "rgb" / ("%06d.png" % (k + 2))), rgb)
    np.savetxt(str(root / "data.txt"), np.zeros((5, 9)))
    spec = importlib.util.spec_from_file_location("flow_dumper", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                              "tools", "flow_dumper.py"))
    fd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fd)
    for nvof, ft in (("nvof1", L.FLOW_S16C2), ("nvof2", L.FLOW_F32C2)):
        out = tmp_path / nvof
        assert fd.main(["x", str(root), "txt", "png", "6", "2", str(W), str(H), nvof, str(out)]) == 0
        assert sorted(os.listdir(out)) == ["%06d.float" % i for i in range(3, 7)]
        for k in range(1, 5):
            ok, fl = io.read_flow(str(out / ("%06d.float" % (k + 2))))
            assert ok and np.array_equal(fl, ops.optical_flow(rgbs[k - 1], rgbs[k], flow_type=ft))
            assert io.flow_format(fl, W) == ((11, 4, 32.0) if nvof == "nvof1" else (13, 1, 1.0))

"""Frame batches (roft_frames_submit): the three per-object chains of the engine walk a batch of frames in one
persistent kernel each.  A batch must give bit for bit what the same frames give one at a time -- same kernels, same
arithmetic, only the launch structure differs -- for every split of the sequence into batches, and both must match
the oracle's ROFTFilter restatement (ROFTFilter::filtering_step, src/roft-lib/src/ROFTFilter.cpp:255-452)."""
import copy

import numpy as np
import pytest
import torch

from roft_amd import _lib as L
from roft_amd import synth

import util
from test_engine_gpu import POS_TOL, make_engine, rot_err

pytestmark = pytest.mark.gpu


def clone(st):
    c = copy.copy(st)
    c.depth, c.flow, c.mask_gt = st.depth.clone(), st.flow.clone(), st.mask_gt.clone()
    c.mask_delivery, c.pose_valid, c.pose_meas, c.flow_valid = (st.mask_delivery.copy(), st.pose_valid.copy(),
                                                                st.pose_meas.copy(), st.flow_valid.copy())
    return c


def awkward_streams(n):
    """Three objects that exercise every branch of the chains: (0) the default schedule, (1) dropped poses, a missing
    flow frame and an empty delivered mask, (2) a three-valued mask {0, 1, 255} (the general, map-based propagation)
    and a pose on every frame (an outlier test per frame: several pose chain segments per batch)."""
    a = clone(util.stream(700, n, scale=2, device="cuda"))
    b = clone(util.stream(701, n, scale=2, pose_drop_prob=0.4, device="cuda"))
    b.flow_valid[[7, 8]] = False
    b.mask_gt[6] = 0
    c = clone(util.stream(702, n, scale=2, device="cuda"))
    m = c.mask_gt.numpy()
    for k in range(m.shape[0]):
        vs, us = np.nonzero(m[k])
        m[k, vs[::3], us[::3]] = 1
    c.pose_valid[:] = True
    for k in range(n):
        c.pose_meas[k, :3] = c.gt.x[k] + 0.002
        c.pose_meas[k, 3:] = c.gt.q[k]
    return [a, b, c]


def check_against_oracle(streams, n, log, masks, **over):
    from oracle import binding as ob
    pose, twist, npts, sel = log
    for o, st in enumerate(streams):
        ref = util.run_oracle_tracker(ob, st, n, **over)
        assert np.array_equal(npts[:, o], np.array([r["n"] for r in ref])), o
        assert np.array_equal(sel[:, o], np.array([r["sel"] for r in ref])), o
        want = np.array([r["pose"] for r in ref])
        assert np.abs(pose[:, o, :9] - want[:, :9]).max() < POS_TOL, o
        assert max(rot_err(pose[k, o, 9:], want[k, 9:]) for k in range(n)) < 1e-6, o
        assert np.abs(twist[:, o] - np.array([r["twist"] for r in ref])).max() < 1e-6, o
        assert np.array_equal(masks[o], ref[n - 1]["mask"]), o


def test_batches_equal_single_frames_bit_for_bit():
    n = 40
    streams = awkward_streams(n)
    dev = [util.to_device(st) for st in streams]
    one, masks1, stats1 = util.run_engine_logged(make_engine, dev, n, T=1)
    check_against_oracle(streams, n, one, masks1)
    assert (one[3][:, 2] >= 0).sum() >= n - 2            # the every-frame pose object tests (nearly) every frame
    for kw in (dict(T=2), dict(T=6), dict(T=8), dict(splits=[3, 1, 8, 5])):
        got, masks, stats = util.run_engine_logged(make_engine, dev, n, **kw)
        for x, y in zip(one, got):
            assert np.array_equal(x, y), kw
        for x, y in zip(masks1, masks):
            assert np.array_equal(x, y), kw
        assert stats["launches"] < stats1["launches"]
    # launches per frame fall with the batch length
    assert stats1["launches"] / n > 5.0


def test_batches_one_stream_equals_three_streams(monkeypatch):
    n = 30
    dev = [util.to_device(st) for st in awkward_streams(n)]
    a, ma, _ = util.run_engine_logged(make_engine, dev, n, T=6)
    monkeypatch.setenv("ROFT_ONE_STREAM", "1")
    b, mb, _ = util.run_engine_logged(make_engine, dev, n, T=6)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for x, y in zip(ma, mb):
        assert np.array_equal(x, y)


def test_batch_ablations_match_oracle():
    """No re-sync (the outlier test reads the features buffered in the same frame) and no flow-aided segmentation,
    in batches."""
    n = 26
    streams = [util.stream(710, n, scale=2, device="cuda")]
    dev = [util.to_device(st) for st in streams]
    for over in (dict(use_pose_resync=0), dict(flow_aided_segmentation=0), dict(outlier_rejection=0, flow_weighting=0)):
        log, masks, _ = util.run_engine_logged(make_engine, dev, n, T=5, **over)
        check_against_oracle(streams, n, log, masks, **over)


def test_batch_size_is_checked_and_a_failed_submit_consumes_nothing():
    n = 12
    st = util.stream(720, n, scale=2, device="cuda")
    dev = util.to_device(st)
    ref, masks_ref, _ = util.run_engine_logged(make_engine, [dev], n, T=4)
    eng = make_engine([dev], max_batch_frames=4)
    eng.enable_log(n)
    with pytest.raises(L.RoftError):
        eng.submit_batch([[util.device_frame(dev, k)] for k in range(5)])      # longer than max_batch_frames
    for k0 in range(0, n, 4):
        frames = [[util.device_frame(dev, k0 + j)] for j in range(4)]
        bad = [[dict(f[0])] for f in frames]
        bad[2][0]["depth"] = None                                               # third frame lacks its depth image
        with pytest.raises(L.RoftError):
            eng.submit_batch(bad)
        with pytest.raises(L.RoftError):
            eng.step()                                                          # nothing was submitted
        eng.submit_batch(frames)                                                # the same frames, corrected
        eng.step()
    got = eng.get_log(0, n)
    assert np.array_equal(eng.mask(0), masks_ref[0])
    eng.close()
    for x, y in zip(ref, got):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("prep", ["1", "2"])
def test_host_buffers_may_be_reused_when_submit_returns(prep, monkeypatch):
    """roft_frame_input: HOST buffers are copied before the submit call returns -- a live caller that refills its one
    pinned capture buffer right after the call must get the same trajectory.  (prep = 2: every batch prepared on the upload
    stream, behind the copies of its own inputs -- ROFT_PREP_AHEAD.)"""
    monkeypatch.setenv("ROFT_PREP_AHEAD", prep)
    n = 20
    st = util.stream(730, n, scale=2, device="cuda")
    dev = util.to_device(st)
    ref, masks_ref, _ = util.run_engine_logged(make_engine, [dev], n, T=1)
    depth = torch.zeros_like(st.depth[0]).pin_memory()
    flow = torch.zeros_like(st.flow[0]).pin_memory()
    mask = torch.zeros_like(st.mask_gt[0]).pin_memory()
    for T in (1, 4):
        eng = make_engine([st], max_batch_frames=T)
        eng.enable_log(n)
        for k0 in range(0, n, T):
            bufs = []
            frames = []
            for k in range(k0, min(k0 + T, n)):
                # one set of pinned buffers per frame of the batch, overwritten with garbage right after the call
                d, f, m = (depth, flow, mask) if T == 1 else (depth.clone().pin_memory(), flow.clone().pin_memory(), mask.clone().pin_memory())
                d.copy_(st.depth[k]); f.copy_(st.flow[k])
                mi = st.mask_delivery[k]
                if mi >= 0:
                    m.copy_(st.mask_gt[mi])
                pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
                frames.append([dict(depth=d.data_ptr(), flow=f.data_ptr() if st.flow_valid[k] else None,
                                    mask=m.data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_HOST)])
                bufs.append((d, f, m))
            eng.submit_batch(frames)
            for d, f, m in bufs:
                d.fill_(float("nan")); f.fill_(1e10); m.fill_(255)
            eng.step()
        got = eng.get_log(0, n)
        assert np.array_equal(eng.mask(0), masks_ref[0])
        eng.close()
        for x, y in zip(ref, got):
            assert np.array_equal(x, y), T


def test_shared_scene_host_inputs_are_uploaded_once():
    """Config #4's shared-scene form: every object points at the same HOST depth and flow image; the engine uploads
    each distinct buffer once per frame."""
    n, n_obj = 6, 4
    st = util.stream(740, n, scale=2, device="cuda")
    eng = make_engine([st] * n_obj)
    depth, flow, masks = st.depth.numpy(), st.flow.numpy(), st.mask_gt.numpy()
    for k in range(n):
        mi = st.mask_delivery[k]
        pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
        f = dict(depth=depth[k].ctypes.data, flow=flow[k].ctypes.data if st.flow_valid[k] else None,
                 mask=masks[mi].ctypes.data if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_HOST)
        eng.submit([f] * n_obj)
        eng.step()
    s = eng.stats()
    poses = [eng.state(o)[0] for o in range(n_obj)]
    eng.close()
    per_frame = depth[0].nbytes + flow[0].nbytes
    n_masks = int((st.mask_delivery[:n] >= 0).sum())
    n_flows = int(st.flow_valid[:n].sum())
    assert s["h2d_bytes"] == n * depth[0].nbytes + n_flows * flow[0].nbytes + n_masks * masks[0].nbytes
    assert s["h2d_bytes"] < 0.5 * n_obj * n * per_frame
    for p in poses[1:]:
        assert np.array_equal(p, poses[0])


def test_mask_chain_result_independent_of_workgroups_per_object():
    """The mask frames of 8, 64 and 200 objects (grids of 24 x n_obj small workgroups per frame, several rounds of the device at
    200): the same eight streams tracked alone and as the first eight of 64 / 200 objects give the same rows and masks bit
    for bit."""
    n = 22
    base = [util.to_device(clone(util.stream(740 + i, n, scale=2, pose_drop_prob=0.2 if i == 3 else 0.03, device="cuda")))
            for i in range(8)]
    ref, ref_masks, _ = util.run_engine_logged(make_engine, base, n, T=8)
    for n_obj in (64, 200):
        streams = [base[i % 8] for i in range(n_obj)]
        got, masks, _ = util.run_engine_logged(make_engine, streams, n, T=8)
        for x, y in zip(ref, got):
            assert np.array_equal(x[:, :8], y[:, :8]), n_obj
            assert np.array_equal(x[:, :8], y[:, n_obj - 8:]), n_obj
        for o in range(8):
            assert np.array_equal(ref_masks[o], masks[o]), (n_obj, o)
            assert np.array_equal(ref_masks[o], masks[n_obj - 8 + o]), (n_obj, o)


def test_mask_workgroups_per_object_changes_nothing():
    """roft_config::mask_workgroups_per_object: one band (workgroup) per object and frame, a few, or the automatic choice (bands
    of ~20 image rows) -- the propagation is an order-free OR, so the whole trajectory is bit for bit the same."""
    n = 20
    streams = [util.to_device(st) for st in awkward_streams(n)]
    ref = None
    for wgs in (0, 1, 2, 5, 8):
        log, masks, _ = util.run_engine_logged(make_engine, streams, n, T=8, mask_workgroups_per_object=wgs)
        if ref is None:
            ref = (log, masks)
            continue
        for a, b in zip(ref[0], log):
            assert np.array_equal(a, b), wgs
        for a, b in zip(ref[1], masks):
            assert np.array_equal(a, b), wgs


def test_outlier_bands_per_alternative_change_nothing():
    """roft_config::outlier_bands_per_alternative: 1 .. 8 horizontal bands per rendered alternative, or the engine's choice
    (which follows the load).  The rendered depths are the same pixels whatever the bands (tests/test_parity_gpu.py) and the
    likelihood's sums are exact integers (LikelihoodSum: tests/test_parity_gpu.py compares the likelihoods themselves across bands,
    strips and vertex-cache settings bit for bit), so the whole log is bit for bit the same."""
    n = 26
    streams = [util.to_device(st) for st in awkward_streams(n)]
    ref = None
    for bands in (0, 1, 2, 3, 8):
        log, masks, _ = util.run_engine_logged(make_engine, streams, n, T=8, outlier_bands_per_alternative=bands)
        if ref is None:
            ref = (log, masks)
            assert (log[3] >= 0).any()
            continue
        for a, b in zip(ref[0], log):
            assert np.array_equal(a, b), bands
        for a, b in zip(ref[1], masks):
            assert np.array_equal(a, b), bands
    with pytest.raises(L.RoftError):
        make_engine(streams, outlier_bands_per_alternative=9)
    # ... and whether the workgroups of an alternative share its triangles (windows merged through memory by the last one to
    # arrive: the default) or only the rows of its window
    try:
        for split in (0, 1):
            L.check(L.lib().roft_debug_outlier_split(split))
            for bands in (0, 4):
                log, masks, _ = util.run_engine_logged(make_engine, streams, n, T=8, outlier_bands_per_alternative=bands)
                for a, b in zip(ref[0], log):
                    assert np.array_equal(a, b), (split, bands)
    finally:
        L.lib().roft_debug_outlier_split(-1)


def test_preparation_ahead_and_early_velocity_gate_change_nothing(monkeypatch):
    """ROFT_PREP_AHEAD (control blocks, counter reset and ingest of a batch on the upload stream, behind the mask chain two batches
    back) and ROFT_MASK_PART_GATE (velocity chain released behind the batch's last-but-one mask frame; the general-mask kernel runs
    twice per batch then): the engine applies them by batch index and object count -- here every setting is forced, over enough
    batches for the steady state (the sixth batch on), on streams with a three-valued mask, dropped poses, a missing flow frame and
    an empty delivered mask -- and rows and masks stay bit for bit the same."""
    n = 66
    dev = [util.to_device(st) for st in awkward_streams(n)]
    monkeypatch.setenv("ROFT_PREP_AHEAD", "0")
    monkeypatch.setenv("ROFT_MASK_PART_GATE", "0")
    ref, ref_masks, _ = util.run_engine_logged(make_engine, dev, n, T=6)
    for prep, part in (("2", "0"), ("0", "2"), ("2", "2"), ("3", "3"), ("1", "1")):
        monkeypatch.setenv("ROFT_PREP_AHEAD", prep)
        monkeypatch.setenv("ROFT_MASK_PART_GATE", part)
        for kw in (dict(T=6), dict(splits=[3, 1, 8, 5, 2])):
            got, masks, _ = util.run_engine_logged(make_engine, dev, n, **kw)
            for x, y in zip(ref, got):
                assert np.array_equal(x, y), (prep, part, kw)
            for x, y in zip(ref_masks, masks):
                assert np.array_equal(x, y), (prep, part, kw)


def test_where_the_feature_kernel_runs_and_what_the_lanes_wait_for_change_nothing(monkeypatch):
    """Round 6: the features kernel of a batch runs behind the velocity filter (ROFT_FEAT_ON_MASK=0) or behind the batch's mask frames
    (2; the default chooses by object count), and pose lanes that are not handed their twists frame by frame (ROFT_HANDOFF=0) wait for
    the velocity filter alone or for the features behind it as well (ROFT_LANES_WAIT_SKF=1 / 0).  Engines read the switches when they
    are created; every combination gives the rows and masks of the first one, bit for bit, on the awkward streams -- incl. an object
    with a pose (and so an outlier test and a feature set) on every frame."""
    n = 42
    dev = [util.to_device(st) for st in awkward_streams(n)]
    ref = None
    combos = [(feat, handoff, skf, "1") for feat in ("0", "2") for handoff in ("0", "2") for skf in ("0", "1")]
    # ... and with every batch prepared ahead on the upload stream, which rewrites the control blocks a feature kernel two batches back
    # has read: behind the mask frames that kernel ends after the event the preparation used to wait for
    combos += [("2", "0", "1", "2"), ("2", "2", "1", "2")]
    for feat, handoff, skf, prep in combos:
        monkeypatch.setenv("ROFT_FEAT_ON_MASK", feat)
        monkeypatch.setenv("ROFT_HANDOFF", handoff)
        monkeypatch.setenv("ROFT_LANES_WAIT_SKF", skf)
        monkeypatch.setenv("ROFT_PREP_AHEAD", prep)
        for kw in (dict(T=6), dict(splits=[3, 1, 8, 5, 2])):
            got = util.run_engine_logged(make_engine, dev, n, **kw)
            if ref is None:
                ref = got
                continue
            for x, y in zip(ref[0], got[0]):
                assert np.array_equal(x, y), (feat, handoff, skf, prep, kw)
            for x, y in zip(ref[1], got[1]):
                assert np.array_equal(x, y), (feat, handoff, skf, prep, kw)


@pytest.mark.parametrize("shared", [False, True])
def test_host_frames_that_are_consecutive_in_memory_go_up_in_one_copy(shared):
    """Round 6: a recorded sequence held as one [frames, H, W] host array -- frame t + 1 starts where frame t ends.  A batch's
    depth images (and its flow images) then cross the bus in ONE copy per run instead of one per frame; the bytes are the same,
    the trajectory is the one DEVICE inputs give, and objects that show the same host pointers (a shared scene) share the run."""
    n, T = 16, 8
    n_obj = 3 if shared else 2
    sts = [util.stream(760, n, scale=2, device="cuda")] * n_obj if shared else [util.stream(760 + o, n, scale=2, device="cuda") for o in range(n_obj)]
    devs = [util.to_device(s) for s in sts]
    ref, masks_ref, _ = util.run_engine_logged(make_engine, devs, n, T=T)
    host = {}
    for s in sts:
        if id(s) not in host:
            host[id(s)] = (s.depth.cpu().pin_memory(), s.flow.cpu().pin_memory(), s.mask_gt.cpu().pin_memory())
    eng = make_engine(sts, max_batch_frames=T)
    eng.enable_log(n)
    for k0 in range(0, n, T):
        frames = []
        for k in range(k0, k0 + T):
            row = []
            for s in sts:
                d, f, m = host[id(s)]
                mi = s.mask_delivery[k]
                pose = (s.pose_meas[k, :3], s.pose_meas[k, 3:]) if s.pose_valid[k] else None
                row.append(dict(depth=d[k].data_ptr(), flow=f[k].data_ptr() if s.flow_valid[k] else None,
                                mask=m[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=s.dt, mem_kind=L.MEM_HOST))
            frames.append(row)
        eng.submit_batch(frames)
        eng.step()
    got = eng.get_log(0, n)
    s = eng.stats()
    eng.close()
    for x, y in zip(ref, got):
        assert np.array_equal(x, y)
    distinct = 1 if shared else n_obj
    st0 = sts[0]
    d0, f0, m0 = host[id(st0)]
    n_flows = int(sum(int(s_.flow_valid[:n].sum()) for s_ in (sts[:1] if shared else sts)))
    n_masks = int(sum(int((s_.mask_delivery[:n] >= 0).sum()) for s_ in sts)) if not shared else None
    assert s["h2d_bytes"] >= distinct * n * d0[0].numel() * 4 + n_flows * f0[0].numel() * f0.element_size()
    # two batches: per distinct stream one depth run per batch, one flow run per batch (frame 0 has no flow: the first batch's
    # run starts at frame 1), + the delivered masks one by one -- not 2 x 16 image copies per stream
    masks_total = sum(int((s_.mask_delivery[:n] >= 0).sum()) for s_ in sts)
    assert s["h2d_copies"] <= distinct * 2 * 2 + masks_total, s
    assert s["h2d_copies"] < distinct * 2 * n

"""the time-stamped live mask source (ImageSegmentationOFAidedSourceStamped.hpp:153-268 with
the 30-flow OpticalFlowQueueHandler): a new mask is propagated through the flows stored after the flow whose stamp
matches the mask's stamp; without a match it goes through the current flow only with mask(0,0) forced to 0."""
import numpy as np
import pytest

from roft_amd import synth

import util


def schedule(n, seed):
    """Irregular live delivery: every 4-7 frames a mask computed on an image 2-6 frames old; one delivery carries a stamp
    that is not in the queue, one is older than the frames_between window."""
    rng = np.random.default_rng(seed)
    deliver = {0: (0, 0.0)}      # frame -> (source frame, stamp); frame 0 initialises
    k = 5
    j = 0
    while k < n:
        lat = int(rng.integers(2, 7))
        src = k - lat
        stamp = src / 30.0
        if j == 2:
            stamp += 0.5          # unknown stamp -> fall-back path
        if j == 4:
            src, stamp = k - 9, (k - 9) / 30.0   # 9 flows in the region, only the last 6 are used
        deliver[k] = (max(src, 0), stamp)
        k += int(rng.integers(4, 8))
        j += 1
    return deliver


def run_oracle(ob, st, n, deliver):
    cfg = util.oracle_config(ob, st, stamped_masks=1)
    trk = ob.Tracker(cfg, *st.mesh)
    out = []
    for k in range(n):
        depth, flow, _, pose = util.frame_inputs(st, k)
        mask, mstamp = None, 0.0
        if k in deliver:
            mask, mstamp = st.mask_gt[deliver[k][0]].cpu().numpy(), deliver[k][1]
        r = trk.step(st.dt, depth, flow, mask, pose, stamp=k / 30.0, mask_stamp=mstamp)
        out.append(dict(pose=np.array(r.pose), n=r.n_flow_points, mask=trk.mask()))
    trk.close()
    return out


def test_oracle_stamped_source_known_answers():
    from oracle import binding as ob
    st = util.stream(90, 12, 4)
    H, W = st.mask_gt.shape[1:]
    n = 12
    flows = st.flow.numpy()
    # (a) a mask of frame 3 delivered at frame 7 with the right stamp == manual propagation through flows 4..7
    deliver = {0: (0, 0.0), 7: (3, 3 / 30.0)}
    got = run_oracle(ob, st, n, deliver)
    manual = ob.mask_propagate(st.mask_gt[3].numpy(), [flows[4], flows[5], flows[6], flows[7]], 6)
    assert np.array_equal(got[7]["mask"], np.where(manual > 1, 255, 0).astype(np.uint8))
    # (b) unknown stamp: the new mask goes through the current flow only, (0,0) forced to 0
    deliver = {0: (0, 0.0), 7: (3, 99.0)}
    got = run_oracle(ob, st, n, deliver)
    m = st.mask_gt[3].numpy().copy()
    m[0, 0] = 0
    manual = ob.mask_propagate(m, [flows[7]], 6)
    assert np.array_equal(got[7]["mask"], np.where(manual > 1, 255, 0).astype(np.uint8))
    # (c) an empty delivered mask is ignored: same as no delivery at all
    a = run_oracle(ob, st, n, {0: (0, 0.0)})
    st2 = type("S", (), {})()
    st2.__dict__.update(st.__dict__)
    st2.mask_gt = st.mask_gt.clone()
    st2.mask_gt[3] = 0
    b = run_oracle(ob, st2, n, {0: (0, 0.0), 7: (3, 3 / 30.0)})
    assert all(np.array_equal(x["mask"], y["mask"]) for x, y in zip(a, b))


@pytest.mark.gpu
def test_engine_stamped_source_matches_oracle():
    from oracle import binding as ob
    from test_engine_gpu import make_engine
    n = 48
    st = util.stream(91, n, 2)
    deliver = schedule(n, 3)
    assert len(deliver) >= 7
    ref = run_oracle(ob, st, n, deliver)
    eng = make_engine([st], stamped_masks=1)
    for k in range(n):
        depth, flow, _, pose = util.frame_inputs(st, k)
        mask, mstamp = None, 0.0
        if k in deliver:
            mask, mstamp = st.mask_gt[deliver[k][0]].cpu().numpy(), deliver[k][1]
        eng.submit([dict(depth=depth, flow=flow, mask=mask, pose=pose, dt=st.dt, stamp=k / 30.0, mask_stamp=mstamp)])
        eng.step()
        assert np.array_equal(eng.mask(0), ref[k]["mask"]), k
        assert eng.outputs()[0].n_flow_points == ref[k]["n"], k
        assert np.abs(eng.state(0)[0] - ref[k]["pose"]).max() < 1e-8, k
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fb", [0, 12])
def test_engine_stamped_source_long_latency(fb):
    """Masks that arrive 20 and 29 frames late are chased through every flow stored after their stamp (up to the 29
    that follow the matching entry of the 30-flow queue, OpticalFlowQueueHandler.cpp:18-57) -- or through the last
    mask_frames_between of them when that number is known; a mask older than the queue takes the fall-back path."""
    from oracle import binding as ob
    from test_engine_gpu import make_engine
    n = 46
    st = util.stream(92, n, 2)
    deliver = {0: (0, 0.0), 25: (5, 5 / 30.0), 33: (4, 4 / 30.0), 41: (9, 9 / 30.0)}
    cfg = util.oracle_config(ob, st, stamped_masks=1, mask_frames_between=fb)
    trk = ob.Tracker(cfg, *st.mesh)
    eng = make_engine([st], stamped_masks=1, mask_frames_between=fb)
    for k in range(n):
        depth, flow, _, pose = util.frame_inputs(st, k)
        mask, mstamp = None, 0.0
        if k in deliver:
            mask, mstamp = st.mask_gt[deliver[k][0]].cpu().numpy(), deliver[k][1]
        r = trk.step(st.dt, depth, flow, mask, pose, stamp=k / 30.0, mask_stamp=mstamp)
        eng.submit([dict(depth=depth, flow=flow, mask=mask, pose=pose, dt=st.dt, stamp=k / 30.0, mask_stamp=mstamp)])
        eng.step()
        assert np.array_equal(eng.mask(0), trk.mask()), k
        assert eng.outputs()[0].n_flow_points == r.n_flow_points, k
        assert np.abs(eng.state(0)[0] - np.array(r.pose)).max() < 1e-8, k
    trk.close()
    eng.close()

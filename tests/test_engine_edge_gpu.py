"""GPU parity of the batched engine on the awkward paths of ROFTFilter::filtering_step: dropped and
irregular pose deliveries (velocity deque of varying length), empty delivered masks, missing flow frames,
unobservable velocity steps (N < 3), the 1280x720 / CV_16SC2 shape, and a longer many-object run."""
import copy

import numpy as np
import pytest
import torch

from roft_amd import _lib as L
from roft_amd import synth

import util
from test_engine_gpu import compare, make_engine

pytestmark = pytest.mark.gpu


def clone(st):
    c = copy.copy(st)
    c.depth, c.flow, c.mask_gt = st.depth.clone(), st.flow.clone(), st.mask_gt.clone()
    c.mask_delivery, c.pose_valid, c.pose_meas, c.flow_valid = (st.mask_delivery.copy(), st.pose_valid.copy(),
                                                                st.pose_meas.copy(), st.flow_valid.copy())
    return c


def test_dropped_and_irregular_poses():
    """40 % of the pose deliveries are dropped: the re-sync replays deques of 7, 13, 19 ... velocities
    trimmed to the last D + 1 (CartesianQuaternionMeasurement.cpp:100-104)."""
    streams = [util.stream(500 + i, 44, scale=2, pose_drop_prob=0.4, device="cuda") for i in range(3)]
    assert any((~st.pose_valid[::6]).any() for st in streams)
    compare(streams, 44)


def test_pose_on_every_frame_and_never():
    st = clone(util.stream(510, 20, scale=2, device="cuda"))
    # a pose measurement on every frame (un-delayed source): re-sync with a one-element deque
    every = clone(st)
    every.pose_valid[:] = True
    for k in range(20):
        every.pose_meas[k, :3] = st.gt.x[k] + 0.002
        every.pose_meas[k, 3:] = st.gt.q[k]
    compare([every], 20)
    never = clone(st)
    never.pose_valid[1:] = False
    compare([never], 20)


def test_empty_delivered_mask_is_ignored():
    """A delivered but empty mask must not replace the propagated one (hpp:186-198)."""
    st = clone(util.stream(520, 20, scale=2, device="cuda"))
    st.mask_gt[6] = 0            # delivered at frame 12
    assert st.mask_delivery[12] == 6
    compare([st], 20)


def test_missing_flow_frames():
    """Flow absent on some frames: no velocity stage, no mask propagation, depth/mask still latched
    (hpp:217-229); the flow buffer simply skips those frames."""
    st = clone(util.stream(530, 24, scale=2, device="cuda"))
    st.flow_valid[[5, 6, 13]] = False
    compare([st], 24)


def test_unobservable_velocity_step():
    """A mask of one or two pixels gives N < 3: the twist belief is restored (ROFTFilter.cpp:294-301)."""
    st = clone(util.stream(540, 14, scale=2, device="cuda"))
    tiny = torch.zeros_like(st.mask_gt[0])
    vs, us = np.nonzero(st.mask_gt[0].numpy())
    tiny[vs[len(vs) // 2], us[len(us) // 2]] = 255
    st.mask_gt[0] = tiny          # delivered at frames 0 and 6
    compare([st], 14)


def test_engine_shape_b_s16():
    """1280x720, CV_16SC2 grid 4 (config_fast_ycb.cfg + nvof_1_slow): render divider 4."""
    streams = [util.stream(550 + i, 9, scale=1, shape="B", flow_type=synth.FLOW_S16C2, mesh_n=24, device="cuda")
               for i in range(2)]
    compare(streams, 9)


def test_long_many_object_run():
    """BASELINE config #5 in small: 16 objects, re-sync + outlier rejection on, 90 frames."""
    streams = [util.stream(600 + i, 90, scale=2, device="cuda") for i in range(16)]
    n_tests = compare(streams, 90, check_masks=False)
    assert n_tests >= 16 * 10


def test_device_buffers_recycled_after_retain_frames():
    """The zero-copy contract of roft_frame_input: a DEVICE buffer handed over for frame k may be overwritten once
    roft_frame_submit() for frame k + ROFT_RETAIN_FRAMES has returned, although several frames are in flight on three
    streams.  The caller keeps exactly ROFT_RETAIN_FRAMES input slots and refills them round-robin."""
    from oracle import binding as ob
    from roft_amd import _lib as L
    from test_engine_gpu import make_engine
    n = 70
    st = util.stream(540, n, scale=2, device="cuda")
    ref = util.run_oracle_tracker(ob, st, n)
    eng = make_engine([st])
    R = eng.retain_frames()
    assert R <= L.RETAIN_FRAMES       # ROFT_RETAIN_FRAMES covers the default configuration
    depth = torch.zeros((R,) + tuple(st.depth.shape[1:]), dtype=st.depth.dtype, device="cuda")
    flow = torch.zeros((R,) + tuple(st.flow.shape[1:]), dtype=st.flow.dtype, device="cuda")
    mask = torch.zeros((R,) + tuple(st.mask_gt.shape[1:]), dtype=st.mask_gt.dtype, device="cuda")
    eng.enable_log(n)
    for k in range(n):
        s = k % R
        # (the copies are synchronous with respect to the host: the data is in place before the submit)
        depth[s].copy_(st.depth[k])
        flow[s].copy_(st.flow[k])
        mi = st.mask_delivery[k]
        if mi >= 0:
            mask[s].copy_(st.mask_gt[mi])
        torch.cuda.synchronize()
        pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
        eng.submit([dict(depth=depth[s].data_ptr(), flow=flow[s].data_ptr() if st.flow_valid[k] else None,
                         mask=mask[s].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE)])
        eng.step()
    pose_log, twist_log, npts, sel = eng.get_log(0, n)
    eng.close()
    want = np.array([r["pose"] for r in ref])
    assert np.abs(pose_log[:, 0] - want).max() < 1e-8
    assert np.array_equal(npts[:, 0], np.array([r["n"] for r in ref]))
    assert np.array_equal(sel[:, 0], np.array([r["sel"] for r in ref]))


def test_unknown_frames_between_masks_chases_all_buffered_flows():
    """mask_frames_between <= 0 (`segm_frames_between_iterations_` unknown): a new mask is chased through ALL flows
    buffered since the last one -- 12 after a skipped delivery -- and an EMPTY delivered mask drops the buffer before
    that frame's flow is stored, so the next mask goes through 7 flows, not 6 (hpp:186-198, 239-245)."""
    st = clone(util.stream(525, 34, scale=2, device="cuda"))
    assert st.mask_delivery[12] == 6 and st.mask_delivery[18] == 12
    st.mask_delivery[12] = -1     # no delivery at frame 12: frame 18's mask goes through the 12 flows 7..18
    st.mask_gt[18] = 0            # delivered (empty) at frame 24: buffer dropped, frame 30's mask goes through 7 flows
    compare([st], 34, mask_frames_between=0)
    compare([st], 34, mask_frames_between=-1)
    compare([st], 34, mask_frames_between=9)


def test_sparse_flows_and_a_mask_outage_over_recycled_buffers():
    """Two of three flow frames are missing and no mask arrives for 36 frames: the six flows the next mask is chased
    through span 18 frames, more than the retention window of the caller's recycled DEVICE buffers.  The reference
    clones every buffered flow; the engine copies a flow into its own memory before the window closes."""
    from oracle import binding as ob
    from roft_amd import _lib as L
    from test_engine_gpu import make_engine
    n = 64
    st = clone(util.stream(545, n, scale=2, device="cuda"))
    st.flow_valid[:] = False
    st.flow_valid[1::3] = True
    st.mask_delivery[12:48] = -1
    ref = util.run_oracle_tracker(ob, st, n)
    eng = make_engine([st])
    R = eng.retain_frames()
    assert R <= L.RETAIN_FRAMES
    depth = torch.zeros((R,) + tuple(st.depth.shape[1:]), dtype=st.depth.dtype, device="cuda")
    flow = torch.zeros((R,) + tuple(st.flow.shape[1:]), dtype=st.flow.dtype, device="cuda")
    mask = torch.zeros((R,) + tuple(st.mask_gt.shape[1:]), dtype=st.mask_gt.dtype, device="cuda")
    eng.enable_log(n)
    for k in range(n):
        s = k % R
        depth[s].copy_(st.depth[k])
        flow[s].fill_(float("nan"))          # a recycled slot never keeps an old flow
        if st.flow_valid[k]:
            flow[s].copy_(st.flow[k])
        mi = st.mask_delivery[k]
        if mi >= 0:
            mask[s].copy_(st.mask_gt[mi])
        torch.cuda.synchronize()
        pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
        eng.submit([dict(depth=depth[s].data_ptr(), flow=flow[s].data_ptr() if st.flow_valid[k] else None,
                         mask=mask[s].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE)])
        eng.step()
        if k in (47, 48, 49, n - 1):
            assert np.array_equal(eng.mask(0), ref[k]["mask"]), k
    pose_log, twist_log, npts, sel = eng.get_log(0, n)
    eng.close()
    assert np.abs(pose_log[:, 0] - np.array([r["pose"] for r in ref])).max() < 1e-8
    assert np.array_equal(npts[:, 0], np.array([r["n"] for r in ref]))


def test_width_not_a_multiple_of_64():
    """160x120: a 64-pixel group straddles image rows, which takes the generic path of the mask gather (and the
    row-remainder handling of every rank query)."""
    streams = [util.stream(560 + i, 30, scale=4, device="cuda") for i in range(2)]
    assert streams[0].camera.width % 64 == 32
    compare(streams, 30)


def _run_logged(streams, n, order=None):
    from roft_amd import _lib as L
    from test_engine_gpu import make_engine
    order = list(range(len(streams))) if order is None else order
    sel = [streams[i] for i in order]
    eng = make_engine(sel)
    eng.enable_log(n)
    for k in range(n):
        frames = []
        for st in sel:
            mi = st.mask_delivery[k]
            pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
            frames.append(dict(depth=st.depth[k].data_ptr(), flow=st.flow[k].data_ptr() if st.flow_valid[k] else None,
                               mask=st.mask_gt[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt,
                               mem_kind=L.MEM_DEVICE))
        eng.submit(frames)
        eng.step()
    out = eng.get_log(0, n)
    eng.close()
    return out


def test_full_size_batch_is_deterministic_and_order_independent(monkeypatch):
    """BASELINE config #4 size (64 objects, 640x480), size-independent properties of the batched, pipelined engine:
    the same inputs give bit-identical trajectories run after run (no race between the three chains), object i's
    result does not depend on which other objects share the launch or on its position in it, and the three-stream
    pipeline equals the serial single-stream execution bit for bit."""
    n, n_obj = 26, 64
    cam = synth.Camera.shape_a()
    streams = [synth.make_stream(8000 + i, n, cam, device="cuda") for i in range(n_obj)]
    a = _run_logged(streams, n)
    b = _run_logged(streams, n)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    perm = list(np.random.default_rng(1).permutation(n_obj))
    c = _run_logged(streams, n, perm)
    for x, y in zip(a, c):
        assert np.array_equal(x[:, perm], y)
    half = _run_logged(streams, n, list(range(0, n_obj, 2)))
    for x, y in zip(a, half):
        assert np.array_equal(x[:, ::2], y)
    monkeypatch.setenv("ROFT_ONE_STREAM", "1")
    d = _run_logged(streams, n)
    for x, y in zip(a, d):
        assert np.array_equal(x, y)
    assert np.isfinite(a[0]).all() and (a[2] >= 0).any() and (a[3] >= 0).any()


def test_two_engines_interleaved_in_one_process():
    """Two engines (different image shapes and flow types) stepped alternately give what each gives alone: no shared
    mutable state between engine handles (streams, rings and device state are per engine)."""
    a = [util.stream(580, 24, scale=2, device="cuda")]
    b = [util.stream(581, 24, scale=2, flow_type=synth.FLOW_S16C2, shape="B", device="cuda")]
    for st in a + b:   # (util.stream leaves the images on the host; this test hands device pointers over)
        st.depth, st.flow, st.mask_gt = st.depth.cuda(), st.flow.cuda(), st.mask_gt.cuda()
    alone_a = _run_logged(a, 24)
    alone_b = _run_logged(b, 24)
    from roft_amd import _lib as L
    from test_engine_gpu import make_engine
    ea, eb = make_engine(a), make_engine(b)
    ea.enable_log(24)
    eb.enable_log(24)
    for k in range(24):
        for eng, sts in ((ea, a), (eb, b)):
            st = sts[0]
            mi = st.mask_delivery[k]
            pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
            eng.submit([dict(depth=st.depth[k].data_ptr(), flow=st.flow[k].data_ptr() if st.flow_valid[k] else None,
                             mask=st.mask_gt[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE)])
            eng.step()
    for x, y in zip(alone_a, ea.get_log(0, 24)):
        assert np.array_equal(x, y)
    for x, y in zip(alone_b, eb.get_log(0, 24)):
        assert np.array_equal(x, y)
    ea.close()
    eb.close()


def test_busy_streams_of_every_engine_have_hardware_queues_of_their_own():
    """The runtime maps HIP streams round robin onto four hardware queues in creation order; two busy chains on one queue
    serialise at the dispatch level.  A new stream set is probed and re-created until its four busy streams (pose lanes,
    velocity, mask) are independent -- also for the third engine of a process and behind streams the application created."""
    import ctypes as C
    import torch
    keep = [torch.cuda.Stream() for _ in range(3)]          # streams of the "application"
    st = util.stream(41, 4, scale=2)
    engines = []
    for _ in range(3):
        eng = make_engine([st])
        # (a shared queue shows in every repetition: hundreds of microseconds; a hiccup of the box in one of them does not count)
        m = None
        for _rep in range(3):
            out = (C.c_double * 25)()
            L.check(L.lib().roft_debug_probe_streams(eng._h, out))
            m1 = np.array(out).reshape(5, 5)
            m = m1 if m is None else np.minimum(m, m1)
        busy = m[:4, :4]
        # the first engine of the process gets an independent set; a later one, created while the others are alive, the best of up
        # to four probed sets -- which on some boxes still has ONE pair on a shared queue (the runtime deals its queues round robin
        # over every stream of the process)
        shared_pairs = int((np.triu(np.maximum(busy, busy.T), 1) >= 50.0).sum())
        assert shared_pairs <= (0 if not engines else 1), np.round(m)
        engines.append(eng)
    for eng in engines:
        eng.close()
    del keep


def test_more_objects_than_the_device_has_room_for_mask_workgroups():
    """200 objects (not a multiple of eight, more workgroups per launch than the device holds at once): several rounds of every
    kernel's grid -- every object against the oracle, masks included, through a pose arrival with its re-sync replay and
    outlier test."""
    n_obj, n = 200, 8
    streams = [util.stream(3000 + i, n, scale=4, mesh_n=6, device="cuda") for i in range(n_obj)]
    n_tests = compare(streams, n)
    assert n_tests >= n_obj          # an outlier test per object at the arrival on frame 6


def test_roofline_kernel_is_timed_twice_and_timing_changes_nothing():
    """roft_engine_enable_timing: the flow measurement's launches come back with their HIP event pair ("flow_measure") and on
    the device's own clock ("flow_measure_span": first workgroup in -> last workgroup out, what bench.py's roofline.kernel_span
    quotes) -- as many launches, a span that is positive and not longer than the event pair's figure plus the clock's
    resolution --, the stamps change no result, and a second collection starts from zero."""
    n = 12
    streams = [util.stream(560 + i, n, scale=2, device="cuda") for i in range(3)]

    def run(timing):
        eng = make_engine(streams)
        eng.enable_log(n)
        if timing:
            eng.enable_timing(1)
        got = []
        for rep in range(2 if timing else 1):
            for k in range(rep * n // 2, (rep + 1) * n // 2) if timing else range(n):
                frames = []
                for st in streams:
                    depth, flow, mask, pose = util.frame_inputs(st, k)
                    frames.append(dict(depth=depth, flow=flow, mask=mask, pose=pose, dt=st.dt))
                eng.submit(frames)
                eng.step()
            if timing:
                got.append(eng.timing())
        out = eng.get_log(0, n)
        eng.close()
        return out, got

    plain, _ = run(False)
    timed, marks = run(True)
    for x, y in zip(plain, timed):
        assert np.array_equal(x, y)
    for tm in marks:
        ms_ev, n_ev = tm["flow_measure"]
        ms_sp, n_sp = tm["flow_measure_span"]
        assert n_ev == n_sp == n // 2
        assert 0.0 < ms_sp <= ms_ev + 1e-4 * n_sp, (ms_sp, ms_ev)


def test_host_pointer_declared_as_device_memory_is_refused():
    """The first submit of an engine looks its ROFT_MEM_DEVICE pointers up: a host buffer handed over as device memory is an
    error with its reason, not a GPU page fault at the first kernel; the engine is usable afterwards."""
    st = util.stream(570, 4, scale=2, device="cuda")   # (util.stream leaves the images on the host)
    eng = make_engine([st])
    depth, flow, mask, pose = util.frame_inputs(st, 0)
    bad = dict(depth=depth.ctypes.data, flow=None, mask=mask.ctypes.data if mask is not None else None, pose=pose, dt=st.dt,
               mem_kind=L.MEM_DEVICE)
    with pytest.raises(L.RoftError) as err:
        eng.submit([bad])
    assert "neither device memory nor pinned" in str(err.value) and "depth" in str(err.value)
    dev, dev_mask = torch.from_numpy(depth).cuda(), torch.from_numpy(mask).cuda()
    ok = dict(depth=dev.data_ptr(), flow=None, mask=dev_mask.data_ptr(), pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE)
    eng.submit([ok])
    eng.step()
    eng.sync()
    eng.close()


def test_pinned_host_memory_is_read_in_place_as_device_input():
    """Pinned, mapped host memory (hipHostMalloc / hipHostRegister: what torch's pin_memory() allocates) handed over as
    ROFT_MEM_DEVICE is read in place over the bus -- only the sectors the kernels touch cross it, nothing is uploaded -- under the
    retention contract of DEVICE inputs; the trajectory equals the one tracked from HBM-resident copies bit for bit."""
    n = 14
    st = util.stream(571, n, scale=2, device="cuda")
    dev = util.to_device(st)
    ref, ref_masks, _ = util.run_engine_logged(make_engine, [dev], n, T=1)
    import copy
    pin = copy.copy(st)
    pin.depth, pin.flow, pin.mask_gt = st.depth.pin_memory(), st.flow.pin_memory(), st.mask_gt.pin_memory()
    assert pin.depth.is_pinned() and not pin.depth.is_cuda
    eng = make_engine([pin])
    eng.enable_log(n)
    h2d0 = eng.stats()["h2d_bytes"]
    for k in range(n):
        eng.submit([util.device_frame(pin, k)])       # mem_kind = MEM_DEVICE, pointers into the pinned host tensors
        eng.step()
    log = eng.get_log(0, n)
    assert eng.stats()["h2d_bytes"] == h2d0           # nothing was staged
    masks = [eng.mask(0)]
    eng.close()
    for a, b in zip(ref, log):
        assert np.array_equal(a, b)
    assert np.array_equal(ref_masks[0], masks[0])


def test_batch_trace_reports_a_schedule_that_depends_on_the_batch_index_only():
    """roft_engine_get_batch_trace: the scheduling mode of a batch is a function of the number of batches stepped since the engine
    was last idle (five: the in-flight bound) -- not of whether the submit call happened to wait --, so two runs of the same
    sequence take the same decisions batch for batch and log the same results; roft_sync starts a new burst; the host's times and
    the completion marks are filled in."""
    n = 60
    streams = [util.to_device(util.stream(580 + i, n, scale=2, device="cuda")) for i in range(3)]

    def run(sync_at=None):
        eng = make_engine(streams, max_batch_frames=6)
        eng.enable_log(n)
        k = 0
        while k < n:
            t = min(6, n - k)
            eng.submit_batch([[util.device_frame(st, k + j) for st in streams] for j in range(t)])
            eng.step()
            k += t
            if sync_at is not None and k == sync_at:
                eng.sync()
        log = eng.get_log(0, n)
        tr = eng.batch_trace()
        eng.close()
        return log, tr

    log_a, tr_a = run()
    log_b, tr_b = run()
    assert len(tr_a) == 10 and [b["batch"] for b in tr_a] == list(range(10)) and all(b["frames"] == 6 for b in tr_a)
    assert [b["steady"] for b in tr_a] == [0] * 5 + [1] * 5
    keys = ("steady", "handoff", "early_lanes", "outlier_parts_halved", "launches", "event_ops")
    assert [[b[k] for k in keys] for b in tr_a] == [[b[k] for k in keys] for b in tr_b]
    for x, y in zip(log_a, log_b):
        assert np.array_equal(x, y)
    for b in tr_a:
        assert b["submit_us"] > 0 and b["step_us"] > 0 and b["wait_us"] >= 0 and b["t_done_us"] > b["t_submit_us"] and b["launches"] >= 8
    # a synchronisation in the middle: the batches behind it are a burst again, and no result changes
    log_c, tr_c = run(sync_at=36)
    assert [b["steady"] for b in tr_c] == [0] * 5 + [1] + [0] * 4
    for x, y in zip(log_a, log_c):
        assert np.array_equal(x, y)


def test_lanes_are_released_early_only_by_the_only_engine_of_the_device():
    """whether a burst batch releases its pose lanes early (they then spin for twists whose producer is not even
    enqueued) is decided from a COUNT -- the stream sets this process holds on the device when the batch is submitted --, never
    from whether another engine happens to be busy at that instant.  One engine alone: early lanes in its burst batches.  Two
    engines in one process, stepped alternately in batches: none in either, and both log what each logs alone, bit for bit."""
    n = 36
    sa = [util.to_device(util.stream(590 + i, n, scale=2, device="cuda")) for i in range(2)]
    sb = [util.to_device(util.stream(595 + i, n, scale=2, device="cuda")) for i in range(3)]

    def batches(eng, streams, k0, t):
        eng.submit_batch([[util.device_frame(st, k0 + j) for st in streams] for j in range(t)])
        eng.step()

    def alone(streams):
        eng = make_engine(streams, max_batch_frames=6)
        eng.enable_log(n)
        for k0 in range(0, n, 6):
            batches(eng, streams, k0, 6)
        log, tr = eng.get_log(0, n), eng.batch_trace()
        eng.close()
        return log, tr

    log_a, tr_a = alone(sa)
    log_b, tr_b = alone(sb)
    # (bursts of the only engine release lanes early -- on a stream set whose four chains were probed onto hardware queues of their
    #  own, which is what a fresh process gets; a set with conflicts never does, and then there is nothing to compare)
    early_alone = any(b["early_lanes"] for b in tr_a) and any(b["early_lanes"] for b in tr_b)
    ea, eb = make_engine(sa, max_batch_frames=6), make_engine(sb, max_batch_frames=6)
    ea.enable_log(n)
    eb.enable_log(n)
    for k0 in range(0, n, 6):
        batches(ea, sa, k0, 6)
        batches(eb, sb, k0, 6)
    got_a, got_b = ea.get_log(0, n), eb.get_log(0, n)
    tr2_a, tr2_b = ea.batch_trace(), eb.batch_trace()
    eb.close()
    # ... and once the other engine is gone, the remaining one is alone again: a new burst releases its lanes early
    ea.sync()
    more = [util.to_device(util.stream(590 + i, n + 12, scale=2, device="cuda")) for i in range(2)]
    for k0 in range(n, n + 12, 6):
        batches(ea, more, k0, 6)
    tr3_a = ea.batch_trace()
    ea.close()
    assert not any(b["early_lanes"] for b in tr2_a) and not any(b["early_lanes"] for b in tr2_b)
    if early_alone:
        assert any(b["early_lanes"] for b in tr3_a if b["batch"] >= n // 6)
    for x, y in zip(log_a, got_a):
        assert np.array_equal(x, y)
    for x, y in zip(log_b, got_b):
        assert np.array_equal(x, y)


def test_a_mesh_whose_triangles_point_outside_its_vertices_is_refused():
    """roft_object_add / roft_render_depth: the rasteriser indexes the vertex array with the triangles' entries -- an index outside
    of it is refused at the boundary (ROFT_ERR_INVALID with the triangle named), not read."""
    from roft_amd import engine as E
    from roft_amd import ops
    st = util.stream(597, 2, scale=2)
    verts, tris = st.mesh
    bad = tris.copy()
    bad[5, 1] = len(verts)
    cfg = E.default_config(st.camera.width, st.camera.height, st.flow_type, max_objects=1)
    eng = E.ROFTFilterBatch(cfg)
    with pytest.raises(L.RoftError) as ei:
        eng.add_object(E.default_object(), verts, bad)
    assert "triangle 5" in str(ei.value)
    bad[5, 1] = -1
    with pytest.raises(L.RoftError):
        eng.add_object(E.default_object(), verts, bad)
    eng.add_object(E.default_object(), verts, tris)     # the engine is usable afterwards
    eng.close()
    cam = L.Camera(st.camera.width, st.camera.height, st.camera.fx, st.camera.fy, st.camera.cx, st.camera.cy)
    with pytest.raises(L.RoftError):
        ops.render_depth(ops.make_mesh(verts, bad), st.gt.x[0], st.gt.q[0], cam, 2)

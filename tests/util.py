"""Shared helpers for the tests: cached synthetic streams and oracle/engine drivers."""
import functools

import numpy as np

from roft_amd import synth


@functools.lru_cache(maxsize=16)
def stream(seed, n_frames, scale=2, flow_type=synth.FLOW_F32C2, shape="A", device="cpu", **kw):
    """Seeded synthetic stream.  device="cuda" generates on the GPU (fast) and moves the tensors to the
    host so that the oracle and the engine see the same bytes."""
    cam = synth.Camera.shape_a() if shape == "A" else synth.Camera.shape_b()
    if scale > 1:
        cam = cam.scaled(scale)
    st = synth.make_stream(seed, n_frames, cam, flow_type=flow_type, mesh_n=kw.pop("mesh_n", 12), device=device, **kw)
    if device != "cpu":
        st.depth, st.flow, st.mask_gt = st.depth.cpu(), st.flow.cpu(), st.mask_gt.cpu()
    return st


def oracle_camera(ob, cam):
    return ob.camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy)


def frame_inputs(st, k):
    """What the Dataset* sources would hand to the tracker at frame k."""
    mask = st.mask_gt[st.mask_delivery[k]].cpu().numpy() if st.mask_delivery[k] >= 0 else None
    pose = (st.pose_meas[k, :3].copy(), st.pose_meas[k, 3:].copy()) if st.pose_valid[k] else None
    i = st.image(k) if hasattr(st, "image") else k
    flow = st.flow[i].cpu().numpy() if st.flow_valid[k] else None
    depth = st.depth[i].cpu().numpy()
    return depth, flow, mask, pose


def oracle_config(ob, st, **over):
    cam = st.camera
    cfg = ob.default_config(640 if cam.width in (640, 320, 160) else 1280, cam.height)
    cfg.cam.width, cfg.cam.height = cam.width, cam.height
    cfg.cam.fx, cfg.cam.fy, cfg.cam.cx, cfg.cam.cy = cam.fx, cam.fy, cam.cx, cam.cy
    m0 = synth.initial_pose_from_stream(st)
    for i in range(13):
        cfg.p_mean0[i] = m0[i]
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def run_oracle_tracker(ob, st, n_frames=None, **over):
    cfg = oracle_config(ob, st, **over)
    verts, tris = st.mesh
    trk = ob.Tracker(cfg, verts, tris)
    # the oracle picks the render divider from width == 640 like the reference; scaled test
    # cameras use the same rule (ROFTFilter.cpp:191-193)
    out = []
    for k in range(n_frames or st.n_frames):
        depth, flow, mask, pose = frame_inputs(st, k)
        r = trk.step(st.dt, depth, flow, mask, pose)
        out.append(dict(pose=np.array(r.pose), twist=np.array(r.twist), n=r.n_flow_points,
                        sel=r.outlier_selected, L=np.array(r.outlier_L), mask=trk.mask()))
    trk.close()
    return out


def to_device(st):
    """Copy of a stream whose image tensors live in HBM (zero-copy DEVICE inputs of the engine)."""
    import copy
    c = copy.copy(st)
    c.depth, c.flow, c.mask_gt = st.depth.cuda(), st.flow.cuda(), st.mask_gt.cuda()
    return c


def device_frame(st, k):
    """roft_frame_input of frame k for a stream returned by to_device()."""
    from roft_amd import _lib as L
    mi = st.mask_delivery[k]
    pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
    i = st.image(k)
    return dict(depth=st.depth[i].data_ptr(), flow=st.flow[i].data_ptr() if st.flow_valid[k] else None,
                mask=st.mask_gt[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE)


def run_engine_logged(make_engine, streams, n, T=1, splits=None, **over):
    """Runs device-resident streams through the engine in batches of T frames (or the explicit list `splits` of batch
    sizes) and returns (pose, twist, n_points, outlier decision) logs plus the final masks."""
    eng = make_engine(streams, max_batch_frames=max(T, max(splits) if splits else 1), **over)
    eng.enable_log(n)
    k = 0
    i = 0
    while k < n:
        t = min(splits[i % len(splits)] if splits else T, n - k)
        i += 1
        if t == 1 and not splits:
            eng.submit([device_frame(st, k) for st in streams])
        else:
            eng.submit_batch([[device_frame(st, k + j) for st in streams] for j in range(t)])
        eng.step()
        k += t
    log = eng.get_log(0, n)
    masks = [eng.mask(o) for o in range(len(streams))]
    stats = eng.stats()
    eng.close()
    return log, masks, stats

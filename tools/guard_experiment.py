import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, torch
from roft_amd import _lib as L, synth, engine as E
import run_baseline_configs as rb
n_obj, n = 64, 49
dev = torch.device("cuda", 0)
cam = synth.Camera.shape_a()
streams = []
for gid in range(n_obj):
    scale = 0.8 + 0.4 * (((gid % 64) * 7) % 10) / 9.0
    half = tuple(h * scale for h in synth.CRACKER_BOX_HALF_EXTENTS)
    streams.append(synth.make_stream(4000 + gid, n, cam, flow_type=synth.FLOW_F32C2, half_extents=half, device=dev))
def run(guard, guard_bil):
    st0 = streams[0]
    cfg = E.default_config(cam.width, cam.height, st0.flow_type, max_objects=n_obj, max_batch_frames=8)
    cfg.cam.fx, cfg.cam.fy, cfg.cam.cx, cfg.cam.cy = cam.fx, cam.fy, cam.cx, cam.cy
    cfg.ukf_cholesky_guard, cfg.ukf_cholesky_guard_bilinear = guard, guard_bil
    eng = E.ROFTFilterBatch(cfg)
    for st in streams:
        d = E.default_object(); m0 = synth.initial_pose_from_stream(st)
        for i in range(13): d.p_mean0[i] = m0[i]
        eng.add_object(d, *st.mesh)
    eng.enable_log(n)
    batches = []
    for k0, t in E.aligned_batches(0, n, 8, 6):
        fl = [[dict(depth=st.depth[k].data_ptr(), flow=st.flow[k].data_ptr() if st.flow_valid[k] else None,
                    mask=st.mask_gt[st.mask_delivery[k]].data_ptr() if st.mask_delivery[k] >= 0 else None,
                    pose=(st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None, dt=st.dt, mem_kind=L.MEM_DEVICE) for st in streams] for k in range(k0, k0 + t)]
        batches.append(eng.build_batch(fl))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for arr, keep, t in batches:
        eng.submit_batch_raw(arr, t); eng.step()
    eng.sync(); dt = time.perf_counter() - t0
    log = eng.get_log(0, n); eng.close()
    return dt, log
base = None
for g, gb in ((0.0, 0.0), (4e-4, 8e-3), (1.6e-3, 3.2e-2), (6.4e-3, 0.128), (1.0, 10.0)):
    run(g, gb)
    dt, log = run(g, gb)
    if base is None: base = log
    dev_pose = np.abs(log[0] - base[0]); 
    print("guard %.1e: %.2f ms for %d frames (%.3g obj-frames/s) | max |pose - eigen| all frames %.2e, frames<25 %.2e, decisions equal %s, npts equal %s"
          % (g, dt * 1e3, n, n_obj * n / dt, dev_pose.max(), dev_pose[:25].max(), np.array_equal(log[3], base[3]), np.array_equal(log[2], base[2])))

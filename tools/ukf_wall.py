#!/usr/bin/env python3
"""Wall time per pose step INSIDE ukf_chain_kernel (library built with -DROFT_UKF_WALL), pipelined batches vs one
batch at a time: python tools/ukf_wall.py [n_obj] [frames]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from roft_amd import _lib as L, synth
import run_baseline_configs as rb

n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
first = int(sys.argv[4]) if len(sys.argv) > 4 else T   # frames of the first batch (1 with T = 6: batches end with the pose-arrival frame, like bench.py's)
dev = torch.device("cuda", 0)
streams = [synth.make_stream(4000 + i, n, synth.Camera.shape_a(), device=dev) for i in range(n_obj)]
for mode in ("pipelined", "one batch at a time"):
    eng = rb.make_engine(streams, T)
    batches = []
    cuts = [0] + list(range(first, n, T)) + [n]
    for k0, k1 in zip(cuts[:-1], cuts[1:]):
        fl = []
        for k in range(k0, k1):
            frames = []
            for st in streams:
                mi = st.mask_delivery[k]
                pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
                frames.append(dict(depth=st.depth[k].data_ptr(), flow=st.flow[k].data_ptr() if st.flow_valid[k] else None,
                                   mask=st.mask_gt[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE))
            fl.append(frames)
        batches.append(eng.build_batch(fl))
    import time
    t0 = time.perf_counter()
    for arr, _keep, t in batches:
        eng.submit_batch_raw(arr, t)
        eng.step()
        if mode != "pipelined":
            eng.sync()
    eng.sync()
    dt = time.perf_counter() - t0
    tot = steps = 0
    recs = []
    for o in range(n_obj):
        buf = (C.c_longlong * 32)()
        L.lib().roft_debug_get_dbg(eng._h, o, buf)
        tot += buf[28]
        steps += buf[29]
        fs = globals().setdefault("first_acc", [0, 0, 0])
        fs[0] += buf[26]; fs[1] += buf[27]; fs[2] += buf[19]
        hb = globals().setdefault("hist_acc", [0, 0, 0, 0, 0, 0])
        for i in range(6): hb[i] += buf[20 + i]
        for lane in range(2):
            r = [buf[lane * 8 + i] for i in range(8)]
            if r[5]:
                first = (1 << 62) - r[0]
                recs.append((first, lane, (r[1] - first) / 100.0, (r[2] - first) / 100.0, r[3], r[4] / max(r[5], 1), r[5], r[6] / 100.0, r[7] / 100.0))
    recs.sort()
    for first, lane, skew, dur, smax, smean, nwg, ps_max, wg_max in recs[:48]:
        print("  lane %d  t %9.1f  last workgroup starts +%6.1f us, launch lasts %6.1f us, steps max %2d mean %5.2f, %d workgroups walked; slowest workgroup %.1f us, slowest us/step %.1f" % (
            lane, (first - recs[0][0]) / 100.0, skew, dur, smax, smean, nwg, wg_max, ps_max))
    print("%-20s %.1f us per frame; %d steps (%.2f per object-frame), %.2f us wall per step inside the kernel" % (
        mode, 1e6 * dt / n, steps, steps / (n * n_obj), tot / 100.0 / max(steps, 1)))
    fs = globals().pop("first_acc")
    print("   launches that walked a step: %d; kernel entry -> first step %.1f us, first step %.1f us (mean)" % (fs[2], fs[0] / 100.0 / max(fs[2], 1), fs[1] / 100.0 / max(fs[2], 1)))
    hb = globals().pop("hist_acc")
    print("   steps (not the first of a launch) < 18 us: %d, 18-22: %d, 22-30: %d, > 30: %d (mean %.1f us; sum of n_corr*100+type: %d)" % (hb[0], hb[1], hb[2], hb[3], hb[4] / 100.0 / max(hb[3], 1), hb[5]))
    eng.close()

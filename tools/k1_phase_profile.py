#!/usr/bin/env python3
"""Wall-clock profile (100 MHz ticks -> microseconds) of the phases of one flow_measure (K1) launch at the metric shape.
Needs a library built with -DROFT_K1_PROFILE:
  make -C roft_amd/csrc clean && make -C roft_amd/csrc CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DROFT_K1_PROFILE"
Slots: 0 control block read, 1 plane -> LDS, 2 chunk popcounts + block scan, 3 candidate list, 4 depth + flow gathers,
5 validity scan + record writes."""
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
n = 14 if os.environ.get("PHASES") == "fused" else (32 if os.environ.get("PHASES") == "mask" else 12)
dev = torch.device("cuda", 0)
# SHAPE=B FLOW=s16: 1280x720 with CV_16SC2 grid-4 flow (config_fast_ycb.cfg) instead of the metric shape
cam = synth.Camera.shape_b() if os.environ.get("SHAPE") == "B" else synth.Camera.shape_a()
ftype = synth.FLOW_S16C2 if os.environ.get("FLOW") == "s16" else synth.FLOW_F32C2
streams = [synth.make_stream(4000 + i, n, cam, flow_type=ftype, device=dev) for i in range(n_obj)]
eng = rb.make_engine(streams)
names = ["ctrl", "plane->LDS", "popc+scan", "cand list", "gathers", "scan+write"]
if os.environ.get("PHASES") == "feat":
    names = ["count", "scan+ranks", "expand", "gathers"]
if os.environ.get("PHASES") == "mask":   # -DROFT_MASK_PROFILE: workgroup 0 of each object, mask_step_kernel
    names = ["ctrl + decide", "zero next + LDS plane", "group list", "walks", "flush"]
if os.environ.get("PHASES") == "fused":   # -DROFT_FUSED_PROFILE: alternative 0 of each object, outlier_fused_kernel
    names = ["vertices + box", "clear", "triangles", "features", "strips", "window w", "window h"]
if os.environ.get("PHASES") == "skf":
    names = ["load", "innovations", "norms", "median", "mean abs dev", "max weight", "accumulate", "reduce", "solve"]
if os.environ.get("PHASES") == "mask":
    # the persistent chain kernel: batches of 8 frames, stamps summed over the frames of a batch (phase 5: the barrier
    # among the object's workgroups between two frames)
    names.append("object barrier")
    T = 8
    eng.close()
    eng = rb.make_engine(streams, T)
    for k0 in range(0, 32, T):
        fl = []
        for k in range(k0, k0 + T):
            frames = []
            for st in streams:
                mi = st.mask_delivery[k]
                pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
                kf = 1 if os.environ.get("FLOWFIX") else k   # FLOWFIX=1: every frame reads the same flow image (warm caches / TLB)
                frames.append(dict(depth=st.depth[k].data_ptr(), flow=st.flow[kf].data_ptr() if st.flow_valid[k] else None,
                                   mask=st.mask_gt[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE))
            fl.append(frames)
        arr, _keep, t = eng.build_batch(fl)
        eng.submit_batch_raw(arr, t)
        eng.step()
        eng.sync()
        rows = []
        for o in range(n_obj):
            buf = (C.c_longlong * 32)()
            L.lib().roft_debug_get_dbg(eng._h, o, buf)
            rows.append([max(buf[q * 8 + i] for q in range(4)) / 100.0 / T for i in range(len(names))])
        r = np.array(rows)
        print("batch at %d  mean us per frame and phase: " % k0 + ", ".join("%s %.2f" % (nm, v) for nm, v in zip(names, r.mean(0))) +
              "  | sum %.2f (max over objects %.2f)" % (r.sum(1).mean(), r.sum(1).max()))
    eng.close()
    sys.exit(0)
for k in range(n):
    frames = []
    for st in streams:
        mi = st.mask_delivery[k]
        pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
        frames.append(dict(depth=st.depth[k].data_ptr(), flow=st.flow[k].data_ptr() if st.flow_valid[k] else None,
                           mask=st.mask_gt[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE))
    eng.submit_raw(eng.build_inputs(frames)[0])
    eng.step()
    eng.sync()
    if k >= 8 and (os.environ.get("PHASES") != "fused" or k in (6, 12)):
        rows = []
        for o in range(n_obj):
            buf = (C.c_longlong * 32)()
            L.lib().roft_debug_get_dbg(eng._h, o, buf)
            if os.environ.get("PHASES") == "mask":   # slowest of the object's first four workgroups, phase by phase
                rows.append([max(buf[q * 8 + i] for q in range(4)) / 100.0 for i in range(len(names))])
            else:
                rows.append([buf[i] / 100.0 for i in range(len(names))])
        r = np.array(rows)
        print("frame %d  mean us per phase: " % k + ", ".join("%s %.2f" % (nm, v) for nm, v in zip(names, r.mean(0))) +
              "  | sum %.2f (max over objects %.2f)" % (r.sum(1).mean(), r.sum(1).max()))
eng.close()

#!/usr/bin/env python3
"""Splits the host-side cost of one frame into roft_frame_submit and roft_step (64 objects, inputs resident in HBM)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from roft_amd import _lib as L
from roft_amd import synth
from test_engine_gpu import make_engine

n_obj, n = int(os.environ.get("N_OBJ", 64)), 72
cam = synth.Camera.shape_a()
streams = [synth.make_stream(7000 + i, n, cam, device="cuda") for i in range(n_obj)]
eng = make_engine(streams)
inputs = []
for k in range(n):
    frames = []
    for st in streams:
        mi = st.mask_delivery[k]
        pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
        frames.append(dict(depth=st.depth[k].data_ptr(), flow=st.flow[k].data_ptr() if st.flow_valid[k] else None,
                           mask=st.mask_gt[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE))
    inputs.append(eng.build_inputs(frames))
torch.cuda.synchronize()
ts = tp = 0.0
t00 = time.perf_counter()
for k in range(n):
    t0 = time.perf_counter()
    eng.submit_raw(inputs[k][0])
    t1 = time.perf_counter()
    eng.step()
    t2 = time.perf_counter()
    if k >= 12:
        ts += t1 - t0
        tp += t2 - t1
eng.sync()
tot = time.perf_counter() - t00
print("objects %d: submit %.1f us/frame, step %.1f us/frame, wall %.1f us/frame" % (n_obj, 1e6 * ts / (n - 12), 1e6 * tp / (n - 12), 1e6 * tot / n))

#!/usr/bin/env python3
"""Phase stamps of mask_frame_kernel (library built with -DROFT_MASK_PROFILE: bash tools/build_variant.sh maskprof -DROFT_MASK_PROFILE,
ROFT_LIB_SO=build_ab/maskprof.so): one-frame submits of 64 objects at the metric shape, per frame the kernel's phases on the
device's 100 MHz clock, first workgroup in -> last workgroup through each phase (max over all objects)."""
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
n = 20
dev = torch.device("cuda", 0)
cam = synth.Camera.shape_a()
streams = [synth.make_stream(4000 + i, n, cam, flow_type=synth.FLOW_F32C2, device=dev) for i in range(n_obj)]
eng = rb.make_engine(streams)
names = {4: "ctrl+words landed", 0: "decided", 5: "zero issued", 6: "branches", 7: "loop", 1: "listed", 2: "walked", 3: "flushed"}
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
    rows = []
    for o in range(n_obj):
        buf = (C.c_longlong * 32)()
        L.lib().roft_debug_get_dbg(eng._h, o, buf)
        rows.append(list(buf))
    r = np.array(rows, dtype=np.int64)
    t0 = ((1 << 62) - r[:, 8]).min()
    if k >= 2:
        print("frame %2d%s: " % (k, " (new mask)" if streams[0].mask_delivery[k] >= 0 else "") +
              ", ".join("%s %.1f" % (names[i], (r[:, i].max() - t0) / 100.0) for i in (4, 0, 5, 6, 7, 1, 2, 3)) +
              " | first wg in -> last start %.1f us" % ((((1 << 62) - r[:, 8]).max() - t0) / 100.0))
    # reset the stamps

eng.close()

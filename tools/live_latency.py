#!/usr/bin/env python3
"""Per-frame latency of the tracker used the way a live robot uses it (ROFTFilter::filtering_step followed by the
logger / viewer reading the estimate, ROFTFilter.cpp:255-452): every frame is submitted, stepped and its state read
back before the next one arrives, so nothing overlaps across frames.

Measured for one object (BASELINE config #2 shape: 1280x720, CV_16SC2 grid 4) and for the five-object scene of config #3,
with the frame handed over in device memory and in pinned host memory (PCIe-inclusive).

usage: python tools/live_latency.py [--frames 240] [--out profiles/r01_live_latency.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

from roft_amd import _lib as L
from roft_amd import synth

import run_baseline_configs as rb


def run_live(streams, n_frames, host):
    eng = rb.make_engine(streams)
    bufs = []
    for st in streams:
        if host:
            bufs.append(dict(depth=st.depth.cpu().pin_memory(), flow=st.flow.cpu().pin_memory(), mask=st.mask_gt.cpu().pin_memory()))
        else:
            bufs.append(dict(depth=st.depth, flow=st.flow, mask=st.mask_gt))
    inputs = []
    for k in range(n_frames):
        frames = []
        for st, b in zip(streams, bufs):
            mi = st.mask_delivery[k]
            pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
            frames.append(dict(depth=b["depth"][k].data_ptr(), flow=b["flow"][k].data_ptr() if st.flow_valid[k] else None,
                               mask=b["mask"][mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt,
                               mem_kind=L.MEM_HOST if host else L.MEM_DEVICE))
        inputs.append(eng.build_inputs(frames))
    eng.sync()
    torch.cuda.synchronize()
    lat = np.zeros(n_frames)
    pose_frames = np.array([bool(streams[0].pose_valid[k]) for k in range(n_frames)])
    for k in range(n_frames):
        t0 = time.perf_counter()
        eng.submit_raw(inputs[k][0])
        eng.step()
        eng.state(0)                       # waits for the frame: pose + twist of the first object on the host
        lat[k] = time.perf_counter() - t0
    eng.close()
    w = lat[12:] * 1e6                     # skip the start-up frames (first launches, first mask ingest)
    pf = pose_frames[12:]
    return dict(median_us=float(np.median(w)), mean_us=float(w.mean()), p99_us=float(np.percentile(w, 99)), max_us=float(w.max()),
                median_us_pose_frames=float(np.median(w[pf])) if pf.any() else None,
                median_us_other_frames=float(np.median(w[~pf])) if (~pf).any() else None, frames=int(len(w)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r01_live_latency.json"))
    args = ap.parse_args()
    L.require_device()
    dev = torch.device("cuda", 0)
    cam_b = synth.Camera.shape_b()
    n = args.frames
    report = {"device": torch.cuda.get_device_name(0), "what": "submit + step + get_state per frame, nothing in flight across frames"}
    one = [synth.make_stream(1000, n, cam_b, flow_type=synth.FLOW_S16C2, device=dev)]
    report["one_object_1280x720_device_inputs"] = run_live(one, n, False)
    report["one_object_1280x720_host_inputs"] = run_live(one, n, True)
    five = [synth.make_stream(3000 + i, n, cam_b, flow_type=synth.FLOW_S16C2, half_extents=rb.FAST_YCB_HALF_EXTENTS[i], device=dev)
            for i in range(5)]
    report["five_objects_1280x720_device_inputs"] = run_live(five, n, False)
    report["five_objects_1280x720_host_inputs"] = run_live(five, n, True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Runs the five workloads of BASELINE.json `configs` on one MI355X and writes a JSON report
(committed under profiles/):

  #1 config_fast_ycb shape (1280x720, CV_16SC2 grid 4), one cracker-box object, CPU reference path end to end
  #2 the same stream on the GPU engine, parity vs #1
  #3 the five Fast-YCB-sized objects batched on one GPU
  #4 64 objects at 640x480 (the metric shape; `bench.py` is the full measurement of this one)
  #5 1280x720, 16 objects, outlier rejection + pose re-sync on a long sequence, tolerance vs the CPU path

usage: python tools/run_baseline_configs.py [--frames5 600] [--out profiles/r02_baseline_configs.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from roft_amd import _lib as L
from roft_amd import engine as E
from roft_amd import metrics, synth

import util
from oracle import binding as ob

FAST_YCB_HALF_EXTENTS = synth.FAST_YCB_HALF_EXTENTS


BATCH = 8   # frames per roft_frames_submit: the configs are recorded sequences


def make_engine(streams, batch=BATCH):
    st0 = streams[0]
    cfg = E.default_config(st0.camera.width, st0.camera.height, st0.flow_type, max_objects=len(streams), max_batch_frames=batch)
    c = st0.camera
    cfg.cam.fx, cfg.cam.fy, cfg.cam.cx, cfg.cam.cy = c.fx, c.fy, c.cx, c.cy
    cfg.flow_grid, cfg.flow_scale = st0.flow_grid, st0.flow_scale
    eng = E.ROFTFilterBatch(cfg)
    for st in streams:
        d = E.default_object()
        m0 = synth.initial_pose_from_stream(st)
        for i in range(13):
            d.p_mean0[i] = m0[i]
        eng.add_object(d, *st.mesh)
    return eng


def run_engine(streams, n_frames, batch=BATCH):
    """The whole sequence through the engine, `batch` frames per submit.  The clock starts after the first two batches
    (code objects loaded, kernel attributes set, pipeline filled once): process start-up is not tracker throughput;
    `seconds` is scaled to the whole sequence."""
    eng = make_engine(streams, batch)
    eng.enable_log(n_frames)
    batches = []
    # batches end with a pose-arrival frame (every 6th frame of the synthetic streams): E.aligned_batches
    for k0, tb in (E.aligned_batches(0, n_frames, batch, 6) if batch >= 6 else [(k, min(batch, n_frames - k)) for k in range(0, n_frames, batch)]):
        frames_list = []
        for k in range(k0, k0 + tb):
            frames = []
            for st in streams:
                mi = st.mask_delivery[k]
                pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
                i = st.image(k)   # (looping streams show their images over and over)
                frames.append(dict(depth=st.depth[i].data_ptr(), flow=st.flow[i].data_ptr() if st.flow_valid[k] else None,
                                   mask=st.mask_gt[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt,
                                   mem_kind=L.MEM_DEVICE))
            frames_list.append(frames)
        batches.append(eng.build_batch(frames_list))
    n_warm = 2 if len(batches) > 4 else 0
    for arr, _keep, t in batches[:n_warm]:
        eng.submit_batch_raw(arr, t)
        eng.step()
    eng.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for arr, _keep, t in batches[n_warm:]:
        eng.submit_batch_raw(arr, t)
        eng.step()
    eng.sync()
    timed = sum(t for _a, _k, t in batches[n_warm:])
    dt = (time.perf_counter() - t0) * n_frames / timed
    pose, twist, npts, sel = eng.get_log(0, n_frames)
    eng.close()
    return dict(pose=pose, twist=twist, n=npts, sel=sel, seconds=dt)


def run_cpu(st, n_frames):
    cfg = util.oracle_config(ob, st)
    trk = ob.Tracker(cfg, *st.mesh)
    depth, flow, masks = st.depth.cpu().numpy(), st.flow.cpu().numpy(), st.mask_gt.cpu().numpy()
    pose = np.zeros((n_frames, 13))
    twist = np.zeros((n_frames, 6))
    sel = np.zeros(n_frames, int)
    npts = np.zeros(n_frames, int)
    t = 0.0
    for k in range(n_frames):
        mi = st.mask_delivery[k]
        pm = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
        i = st.image(k)
        t0 = time.perf_counter()
        r = trk.step(st.dt, depth[i], flow[i] if st.flow_valid[k] else None, masks[mi] if mi >= 0 else None, pm)
        t += time.perf_counter() - t0
        pose[k], twist[k], sel[k], npts[k] = r.pose, r.twist, r.outlier_selected, r.n_flow_points
    trk.close()
    return dict(pose=pose, twist=twist, sel=sel, n=npts, seconds=t)


def compare(eng_all, cpu_list, streams, objects=None):
    """Max deviations of the engine trajectories from the CPU reference path and ADD-S between the two (over the frames
    the CPU path ran; `objects`: engine object index of each CPU trajectory)."""
    out = dict(max_pos_m=0.0, max_rot_rad=0.0, max_twist=0.0, flow_point_sets_equal=True, outlier_decisions_equal=True,
               adds_vs_cpu_mm_mean=0.0, adds_vs_cpu_mm_max=0.0)
    dists = []
    for o, (cpu, st) in enumerate(zip(cpu_list, streams)):
        nc = len(cpu["pose"])
        o = objects[o] if objects else o
        eng = dict(pose=eng_all["pose"][:nc], twist=eng_all["twist"][:nc], n=eng_all["n"][:nc], sel=eng_all["sel"][:nc])
        p = eng["pose"][:, o]
        out["max_pos_m"] = max(out["max_pos_m"], float(np.abs(p[:, :9] - cpu["pose"][:, :9]).max()))
        dq = np.abs(np.sum(p[:, 9:] * cpu["pose"][:, 9:], axis=1)).clip(0, 1)
        out["max_rot_rad"] = max(out["max_rot_rad"], float((2 * np.arccos(dq)).max()))
        out["max_twist"] = max(out["max_twist"], float(np.abs(eng["twist"][:, o] - cpu["twist"]).max()))
        out["flow_point_sets_equal"] &= bool(np.array_equal(eng["n"][:, o], cpu["n"]))
        out["outlier_decisions_equal"] &= bool(np.array_equal(eng["sel"][:, o], cpu["sel"]))
        pts = st.mesh[0].astype(np.float64)[::16]
        est = np.concatenate([p[:, 6:9], p[:, 9:13]], 1)
        ref = np.concatenate([cpu["pose"][:, 6:9], cpu["pose"][:, 9:13]], 1)
        dists.append(metrics.trajectory_adds(est[::5], ref[::5], pts))
    d = np.concatenate(dists)
    out["adds_vs_cpu_mm_mean"] = 1e3 * float(d.mean())
    out["adds_vs_cpu_mm_max"] = 1e3 * float(d.max())
    return out


def accuracy(eng, streams):
    d = []
    for o, st in enumerate(streams):
        est = np.concatenate([eng["pose"][:, o, 6:9], eng["pose"][:, o, 9:13]], 1)
        img = np.array([st.image(k) for k in range(len(est))])
        gt = np.concatenate([st.gt.x, st.gt.q], 1)[img]
        d.append(metrics.trajectory_adds(est[12::5], gt[12::5], st.mesh[0].astype(np.float64)[::16]))
    d = np.concatenate(d)
    return dict(adds_vs_gt_mm_mean=1e3 * float(d.mean()), adds_auc=metrics.auc(d))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--frames5", type=int, default=3000, help="frames of config #5 on the GPU (a looping stream of 60 images)")
    ap.add_argument("--frames5-cpu", type=int, default=600, help="frames of config #5 the CPU path tracks for all 16 objects")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_baseline_configs.json"))
    args = ap.parse_args()
    L.require_device()
    dev = torch.device("cuda", 0)
    cam_b, cam_a = synth.Camera.shape_b(), synth.Camera.shape_a()
    report = {"host_cores": os.cpu_count(), "device": torch.cuda.get_device_name(0), "frames_per_submit": BATCH}

    # ---- #1 / #2: single cracker-box object, shape B, CV_16SC2
    n = args.frames
    st = synth.make_stream(1000, n, cam_b, flow_type=synth.FLOW_S16C2, device=dev)
    cpu = run_cpu(st, n)
    eng = run_engine([st], n)
    report["config1_cpu_reference_path"] = dict(
        workload="1280x720 CV_16SC2 grid 4, 1 object, %d frames" % n, fps=n / cpu["seconds"], ms_per_frame=1e3 * cpu["seconds"] / n,
        frames_over_33ms_budget=0 if 1e3 * cpu["seconds"] / n < 33 else None,
        **accuracy(dict(pose=cpu["pose"][:, None], twist=cpu["twist"][:, None]), [st]))
    report["config2_single_object_gpu"] = dict(fps=n / eng["seconds"], ms_per_frame=1e3 * eng["seconds"] / n,
                                               speedup_vs_cpu_1core=cpu["seconds"] / eng["seconds"],
                                               parity=compare(eng, [cpu], [st]), **accuracy(eng, [st]))
    del st

    # ---- #3: five objects batched, shape B
    streams = [synth.make_stream(3000 + i, n, cam_b, flow_type=synth.FLOW_S16C2, half_extents=FAST_YCB_HALF_EXTENTS[i], device=dev)
               for i in range(5)]
    eng = run_engine(streams, n)
    cpus = [run_cpu(s, n) for s in streams]
    report["config3_five_objects_batched"] = dict(object_frames_per_s=5 * n / eng["seconds"], ms_per_frame=1e3 * eng["seconds"] / n,
                                                  cpu_object_frames_per_s=5 * n / sum(c["seconds"] for c in cpus),
                                                  parity=compare(eng, cpus, streams), **accuracy(eng, streams))
    del streams

    # ---- #4: 64 objects, shape A (short version; bench.py is the reference measurement)
    n4 = 72
    streams = [synth.make_stream(4000 + i, n4, cam_a, device=dev) for i in range(64)]
    eng = run_engine(streams, n4)
    report["config4_64_objects_640x480"] = dict(object_frames_per_s=64 * n4 / eng["seconds"], ms_per_frame=1e3 * eng["seconds"] / n4,
                                                note="see bench.py for the timed measurement with warm-up", **accuracy(eng, streams[:8]))
    del streams

    # ---- #5: 16 objects, shape B, long sequence, tolerance vs the CPU path.  3 000 frames of 16 objects at 1280x720 do not
    #      fit HBM as distinct images (230 GB): the streams loop over 60 images of a closed motion while the delivery
    #      schedules (masks, poses, outliers, drops) run on for the whole length (synth.make_stream(period=...)).
    n5, n5c = args.frames5, min(args.frames5_cpu, args.frames5)
    streams = [synth.make_stream(5000 + i, 0, cam_b, flow_type=synth.FLOW_S16C2, device=dev, half_extents=FAST_YCB_HALF_EXTENTS[i % 5],
                                 period=60, n_schedule=n5) for i in range(16)]
    eng = run_engine(streams, n5)
    cpus = [run_cpu(s, n5c) for s in streams]
    cpus_full = [run_cpu(streams[o], n5) for o in (0, 7)]
    report["config5_16_objects_1280x720_long"] = dict(
        frames=n5, images="60 per object, looping", object_frames_per_s=16 * n5 / eng["seconds"],
        cpu_object_frames_per_s=16 * n5c / sum(c["seconds"] for c in cpus),
        outlier_tests=int((eng["sel"] >= 0).sum()), outliers_rejected=int((eng["sel"] == 1).sum()),
        parity_first_frames_all_objects=dict(frames=n5c, **compare(eng, cpus, streams)),
        parity_whole_sequence_two_objects=dict(frames=n5, objects=[0, 7], **compare(eng, cpus_full, [streams[0], streams[7]], objects=[0, 7])),
        **accuracy(eng, streams))

    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()

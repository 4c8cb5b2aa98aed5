#!/usr/bin/env python3
"""CU x microseconds per object-frame of every kernel of the pipeline, measured: a library built with -DROFT_RESIDENCY
(bash tools/build_variant.sh resid -DROFT_RESIDENCY; ROFT_LIB_SO=build_ab/resid.so) makes every workgroup add the time it was
resident (first instruction -> end of its thread 0, early exits included) to its kernel's counter; x the share of a CU one
workgroup occupies (wave slots, registers, LDS: profiles/r06_kernel_resources.csv + the launch shapes of the 640x480 workload) =
CU x us.  Workload: BASELINE config #4 as bench.py tracks it (64 objects, six-frame batches ending with the pose arrival), all
chains running.     python tools/residency_budget.py [frames] [--objects N] > profiles/r05_residency_budget.csv"""
import csv
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from roft_amd import _lib as L, engine as E, synth
import run_baseline_configs as rb

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 66
n_obj = int(sys.argv[sys.argv.index("--objects") + 1]) if "--objects" in sys.argv else 64
dev = torch.device("cuda", 0)
cam = synth.Camera.shape_a()
streams = []
for gid in range(n_obj):
    scale = 0.8 + 0.4 * (((gid % 64) * 7) % 10) / 9.0
    half = tuple(h * scale for h in synth.CRACKER_BOX_HALF_EXTENTS)
    streams.append(synth.make_stream(4000 + gid, n_frames, cam, flow_type=synth.FLOW_F32C2, half_extents=half, device=dev))
eng = rb.make_engine(streams, 6)


def run(k0, k1):
    for b0, t in E.aligned_batches(k0, k1, 6, 6):
        fl = []
        for k in range(b0, b0 + t):
            frames = []
            for st in streams:
                mi = st.mask_delivery[k]
                pose = (st.pose_meas[k, :3], st.pose_meas[k, 3:]) if st.pose_valid[k] else None
                frames.append(dict(depth=st.depth[k].data_ptr(), flow=st.flow[k].data_ptr() if st.flow_valid[k] else None,
                                   mask=st.mask_gt[mi].data_ptr() if mi >= 0 else None, pose=pose, dt=st.dt, mem_kind=L.MEM_DEVICE))
            fl.append(frames)
        arr, _keep, tt = eng.build_batch(fl)
        eng.submit_batch_raw(arr, tt)
        eng.step()
    eng.sync()


buf = (C.c_ulonglong * 32)()
warm = 12
run(0, warm)
L.check(L.lib().roft_debug_get_residency(eng._h, buf))     # (read and clear: the warm-up does not count)
run(warm, n_frames)
L.check(L.lib().roft_debug_get_residency(eng._h, buf))
eng.close()
obj_frames = float(n_obj * (n_frames - warm))
res = {r["kernel"].split("::")[-1]: r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_kernel_resources.csv")))}
plane = cam.width * cam.height // 8
# (kernel, threads per workgroup, dynamic LDS bytes of the launch at 640x480)
shapes = [("mask_frame_kernel<13, 256>", 256, 4240 + 3072), ("mask_ingest_kernel", 256, 0), ("mask_general_kernel<13>", 256, 19200),
          ("flow_measure_kernel<3>", 1024, 0), ("skf_chain_kernel", 512, 0), ("features_kernel", 1024, plane), ("ukf_chain_kernel", 256, 0),
          ("outlier_fused_kernel", 1024, 98576 + 4 * (320 * 240 // 1 + 320)),
          ("mask_frame_kernel<13, 256> (bands without a pixel)", 256, 4240 + 3072),
          # the frames that deliver a mask: two-wave workgroups, 6-row bands with a 48-row margin (window (6 + 1 + 96) x 80 bytes)
          ("mask_frame_kernel<13, 128>", 128, 8240 + 1536), ("mask_frame_kernel<13, 128> (bands without a pixel)", 128, 8240 + 1536)]
w = csv.writer(sys.stdout)
w.writerow(["kernel", "workgroups", "resident_us_total", "mean_resident_us_per_workgroup", "cu_share_of_one_workgroup", "cu_us_per_object_frame"])
total = 0.0
for kid, (name, threads, dyn_lds) in enumerate(shapes):
    ticks, wgs = buf[2 * kid], buf[2 * kid + 1]
    k = res.get(name.split(" (")[0]) or res.get(name.split("<")[0]) or res.get(name + "<true>")   # (features_kernel<LDS>: the plane fits the LDS at this size)
    waves_per_simd = threads / 64.0 / 4.0   # (two-wave workgroups: half a wave per SIMD on average)
    regs = (int(k["vgprs"] or 0) + int(k["agprs"] or 0) + 7) // 8 * 8
    lds = min(160.0 * 1024.0, float(k["static_lds_bytes"] or 0) + dyn_lds)
    share = min(1.0, max(waves_per_simd / 8.0, waves_per_simd * regs / 512.0, lds / (160.0 * 1024.0)))
    us = ticks * 0.01
    cu_us = us * share / obj_frames
    total += cu_us
    w.writerow([name, wgs, "%.0f" % us, "%.2f" % (us / wgs if wgs else 0.0), "%.3f" % share, "%.2f" % cu_us])
w.writerow(["TOTAL", "", "", "", "", "%.2f" % total])

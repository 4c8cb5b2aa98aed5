#!/usr/bin/env python3
"""CU residency budget of a bench run from a rocprofv3 kernel trace: for every engine kernel, CU x microseconds per object-frame =
sum over its dispatches of (duration x workgroups x the share of a CU one workgroup occupies) / object-frames tracked, where the
share is the largest of its wave slots (waves per SIMD / 8), its registers (waves per SIMD x (VGPRs + AGPRs, in granules of 8) /
512) and its LDS (bytes / 160 KB) -- the footprints from the compiler's own report (tools/kernel_resources.sh), grid and LDS per
dispatch from the trace.  An upper bound per kernel: it charges every workgroup of a dispatch for the whole dispatch, capped at
the device's 256 CUs.
    python tools/cu_budget.py <kernel_trace.csv> <kernel_resources.csv> <object_frames> [first_ctrl_upload] > budget.csv"""
import csv
import sys
from collections import defaultdict

trace, res_path, obj_frames = sys.argv[1], sys.argv[2], float(sys.argv[3])
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
res = {}
for r in csv.DictReader(open(res_path)):
    res[r["kernel"].split("<")[0].split("::")[-1] + ("<" + r["kernel"].split("<", 1)[1] if "<" in r["kernel"] else "")] = r
rows = [r for r in csv.DictReader(open(trace)) if "roft::" in r["Kernel_Name"] or "ctrl_upload" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ups = [i for i, r in enumerate(rows) if "ctrl_upload" in r["Kernel_Name"]]
if ups and first < len(ups):
    rows = rows[ups[first]:]
acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in rows:
    full = r["Kernel_Name"].split("(")[0].replace("void ", "")
    name = full.split("::")[-1]
    k = res.get(name) or res.get(name.split("<")[0])
    wg_threads = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1)
    n_wg = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * max(1, int(r.get("Grid_Size_Y", 1) or 1) // max(1, int(r.get("Workgroup_Size_Y", 1) or 1)))
    waves_per_simd = max(1.0, wg_threads / 64.0 / 4.0)
    regs = 0
    if k:
        regs = (int(k["vgprs"] or 0) + int(k["agprs"] or 0) + 7) // 8 * 8
    lds = float(r.get("LDS_Block_Size") or 0)
    share = max(waves_per_simd / 8.0, waves_per_simd * regs / 512.0, lds / (160.0 * 1024.0))
    share = min(1.0, share)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cus = min(256.0, n_wg * share)
    a = acc[name]
    a[0] += 1
    a[1] += dur
    a[2] += dur * cus
    a[3] = max(a[3], share)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches", "total_us", "cu_share_of_one_workgroup", "cu_us_total", "cu_us_per_object_frame"])
tot = 0.0
for name, (n, dur, cuus, share) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
    w.writerow([name, n, "%.1f" % dur, "%.3f" % share, "%.0f" % cuus, "%.2f" % (cuus / obj_frames)])
    tot += cuus
w.writerow(["TOTAL", "", "", "", "%.0f" % tot, "%.2f" % (tot / obj_frames)])

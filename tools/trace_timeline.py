#!/usr/bin/env python3
"""Prints the kernel timeline (per HIP stream) of the last frames of a rocprofv3 --kernel-trace csv and the busy
time of each stream: python tools/trace_timeline.py <kernel_trace.csv> [n_kernels] [--list]"""
import csv
import sys
from collections import defaultdict

f = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 400
rows = [r for r in csv.DictReader(open(f)) if "roft::" in r["Kernel_Name"] or "ctrl_upload" in r["Kernel_Name"]]
sel = rows[-n - 40:-40]
t0 = int(sel[0]["Start_Timestamp"])
busy = defaultdict(float)
per = defaultdict(lambda: [0, 0.0])
for r in sel:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    busy[r["Stream_Id"]] += e - s
    per[(r["Stream_Id"], name)][0] += 1
    per[(r["Stream_Id"], name)][1] += e - s
    if "--list" in sys.argv:
        print("s%s %-24s %9.1f %9.1f  dur %6.1f" % (r["Stream_Id"], name, s, e, e - s))
span = (int(sel[-1]["End_Timestamp"]) - t0) / 1e3
frames = sum(1 for r in sel if "skf_kernel" in r["Kernel_Name"])
print("span %.1f us, %d frames -> %.1f us/frame" % (span, frames, span / max(frames, 1)))
for k, v in sorted(busy.items()):
    print("stream %s busy %.1f us (%.0f%%), %.1f us/frame" % (k, v, 100 * v / span, v / max(frames, 1)))
for (st, name), (c, t) in sorted(per.items()):
    print("  s%s %-24s n=%4d avg %7.1f us  per-frame %6.1f us" % (st, name, c, t / c, t / max(frames, 1)))

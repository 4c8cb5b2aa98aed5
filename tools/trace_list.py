#!/usr/bin/env python3
"""Lists every engine kernel of a rocprofv3 --kernel-trace csv from the k-th FrameCtrl upload on (k = 2nd argument,
default 0): stream, hardware queue, kernel, start, end, duration (microseconds, relative to that upload):
    python tools/trace_list.py <kernel_trace.csv> [first_batch]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "roft::" in r["Kernel_Name"] or "ctrl_upload" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ups = [i for i, r in enumerate(rows) if "ctrl_upload" in r["Kernel_Name"]]
rows = rows[ups[k]:] if ups and k < len(ups) else rows
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("s%-2s q%-2s %-26s %9.1f %9.1f  %6.1f" % (r["Stream_Id"], r.get("Queue_Id", "?"), name, s, e, e - s))
if "--resources" in sys.argv:
    seen = {}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].split("::")[-1]
        key = (name, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("LDS_Block_Size"))
        if key not in seen:
            seen[key] = 1
            print("# %-26s wg %sx%s grid %sx%s lds %s vgpr %s agpr %s sgpr %s scratch %s" % (
                name, r.get("Workgroup_Size_X"), r.get("Workgroup_Size_Y"), r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("LDS_Block_Size"),
                r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size")))

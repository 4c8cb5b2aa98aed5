#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries committed under profiles/.

  tools/prof_summary.py stats <*_kernel_stats.csv> <out.csv>        keep only the roft:: kernels
  tools/prof_summary.py pmc   <*_counter_collection.csv> <out.csv>  per-kernel mean of each counter
  tools/prof_summary.py window <*_kernel_trace.csv> <name part> <first> <count>
                                                  mean duration of launches [first, first+count) of one kernel
"""
import csv
import sys
from collections import defaultdict


def stats(src, dst):
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if "roft::" in r["Name"]]
    total = sum(float(r["TotalDurationNs"]) for r in keep)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "share_of_roft_time"])
        for r in sorted(keep, key=lambda r: -float(r["TotalDurationNs"])):
            name = r["Name"].split("(")[0].replace("void ", "")
            w.writerow([name, r["Calls"], "%.1f" % (float(r["TotalDurationNs"]) / 1e3), "%.2f" % (float(r["AverageNs"]) / 1e3),
                        "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3),
                        "%.3f" % (float(r["TotalDurationNs"]) / total)])


def pmc(src, dst):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(src)):
        k = r.get("Kernel_Name", "")
        if "roft::" not in k:
            continue
        key = (k.split("(")[0].replace("void ", ""), r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"])
        acc[key][1] += 1
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_value_per_dispatch"])
        for (k, c), (s, n) in sorted(acc.items()):
            w.writerow([k, c, n, "%.3f" % (s / n)])


def window(src, part, first, count):
    rows = [r for r in csv.DictReader(open(src)) if part in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sel = rows[first:first + count]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
    print("%s: launches %d..%d of %d: mean %.2f us, min %.2f, max %.2f" % (part, first, first + len(sel) - 1, len(rows),
                                                                         sum(d) / len(d), min(d), max(d)))


if __name__ == "__main__":
    if sys.argv[1] == "window":
        window(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])

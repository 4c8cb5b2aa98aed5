#!/usr/bin/env python3
"""Profiler-free timeline of the engine's chains from HIP event marks: run bench.py (or any tool that calls
roft_engine_enable_timing(e, 2)) with ROFT_DUMP_MARKS=<file>; every mark is "stream name end_us duration_us" relative to the
first one (roft_engine_get_timing).  Prints the busy share of every stream over a window and, with --list, the marks.
    python tools/marks_timeline.py <file> [--from US] [--to US] [--list]"""
import sys
from collections import defaultdict

STREAM = {0: "mask", 1: "lane0", 2: "vel", 3: "lane1", 4: "up"}
path = sys.argv[1]
lo = float(sys.argv[sys.argv.index("--from") + 1]) if "--from" in sys.argv else 0.0
hi = float(sys.argv[sys.argv.index("--to") + 1]) if "--to" in sys.argv else 1e30
rows = []
for line in open(path):
    p = line.split()
    if len(p) != 4:
        continue
    rows.append((int(p[0]), p[1], float(p[2]), float(p[3])))
rows = [r for r in rows if lo <= r[2] <= hi]
if not rows:
    raise SystemExit("no marks in the window")
span = max(r[2] for r in rows) - min(r[2] - r[3] for r in rows)
busy = defaultdict(float)
per = defaultdict(lambda: [0, 0.0])
for s, name, end, dur in rows:
    if name == "-":
        continue
    busy[s] += dur
    per[(s, name)][0] += 1
    per[(s, name)][1] += dur
print("window %.0f us, %d marks" % (span, len(rows)))
for s in sorted(busy):
    print("stream %-6s busy %8.0f us = %3.0f %%" % (STREAM.get(s, s), busy[s], 100.0 * busy[s] / span))
for (s, name), (n, tot) in sorted(per.items()):
    print("  %-6s %-28s n=%4d avg %7.1f us" % (STREAM.get(s, s), name, n, tot / n))
if "--list" in sys.argv:
    for s, name, end, dur in sorted(rows, key=lambda r: r[2] - r[3]):
        print("%-6s %-28s %9.1f %9.1f  %7.1f" % (STREAM.get(s, s), name, end - dur, end, dur))

#!/usr/bin/env python3
"""Host microseconds inside roft_frames_submit / roft_step per batch of the driver-shaped window (bench_detail.json's batch trace).
usage: python tools/step_us.py <bench_detail.json>"""
import json, sys
d = json.load(open(sys.argv[1]))
for w in d["windows"]:
    print(round(w["value"]), [(b["frames"], b["submit_us"], b["step_us"], b["submitted_at_ms"], b["done_at_ms"]) for b in w["batches"]])

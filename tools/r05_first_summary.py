#!/usr/bin/env python3
"""gpurun_out/r05_fp_*.json (tools/r05_first_repeat.sh: bench.py as the first GPU process of a fresh lease, one gpurun call
each) -> profiles/r05_first_process_repeat.json: the distribution of `value`, of every window, of value_cold, and the host side."""
import glob
import json
import os
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "r05_fp_*.json")) + glob.glob(os.path.join(ROOT, "gpurun_out", "r05_first_*.json"))):
    try:
        d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception:
        continue
    rows.append(dict(file=os.path.basename(path), value=d["value"], runs=d["runs"], value_min=d["value_min"], value_max=d["value_max"],
                     value_cold=d.get("value_cold"), instrumented=(d.get("instrumented_window") or {}).get("value"),
                     host_enqueue_ms_per_step=d["host_enqueue_ms_per_step"], loadavg=d["host_state"]["loadavg_1min"],
                     roofline_frac=(d.get("roofline") or {}).get("frac"), k1_avg_us=(d.get("roofline") or {}).get("avg_launch_us"),
                     cpu_baseline=(d.get("cpu_baseline") or {}).get("value")))
vals = [r["value"] for r in rows]
wins = [w for r in rows for w in r["runs"]]
out = dict(command="python3 bench.py --gpus 1 --steps 20 --warmup 5 (first GPU process of a fresh lease, one gpurun invocation each)",
           invocations=len(rows),
           value=dict(median=st.median(vals), min=min(vals), max=max(vals), all=vals) if vals else None,
           windows=dict(count=len(wins), median=st.median(wins), min=min(wins), max=max(wins),
                        below_7e5=sum(w < 7e5 for w in wins)) if wins else None,
           value_cold=dict(median=st.median([r["value_cold"] for r in rows if r["value_cold"]]),
                           min=min(r["value_cold"] for r in rows if r["value_cold"])) if any(r["value_cold"] for r in rows) else None,
           rows=rows)
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_first_process_repeat.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("invocations", "value", "windows", "value_cold")}, indent=1))

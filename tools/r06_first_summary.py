#!/usr/bin/env python3
"""gpurun_out/r06_fp_<i>.out (tools/r06_first_repeat.sh) -> profiles/r06_first_process_repeat.json: the stdout of bench.py as the
first GPU process of a fresh lease, read the way the driver reads it (the last 8 KB, the last line, strict JSON)."""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "r06_fp_*.out")), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    raw = open(path, "rb").read()
    tail = raw[-8192:].decode()
    last = tail.strip().splitlines()[-1]
    whole = raw.decode().strip().splitlines()[-1]
    d = json.loads(last, parse_constant=lambda c: (_ for _ in ()).throw(ValueError("non-strict JSON: " + c)))
    need = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
    missing = [k for k in need if k not in d]
    rows.append(dict(file=os.path.basename(path), stdout_bytes=len(raw), last_line_bytes=len(whole), last_line_is_whole_in_8KB_tail=(last == whole),
                     missing_keys=missing, value=d["value"], runs=d["runs"], value_cold=d.get("value_cold"), ms_per_step=d["ms_per_step"],
                     roofline_frac=d["roofline"]["frac"], roofline_avg_launch_us=d["roofline"]["avg_launch_us"], roofline_traffic=d["roofline"]["traffic"],
                     cpu_baseline=d["cpu_baseline"]["value"], cpu_cores=d["cpu_baseline"]["cores"],
                     value_pcie_inclusive=d.get("value_pcie_inclusive"), value_pcie_inclusive_shared_scene=d.get("value_pcie_inclusive_shared_scene")))
vals = sorted(r["value"] for r in rows)
out = dict(what="python3 bench.py --gpus 1 --steps 20 --warmup 5 as the first GPU process of a fresh lease, one gpurun invocation each; "
                "the stdout read as the driver reads it (8 KB tail, last line, strict JSON)",
           invocations=len(rows), all_lines_parse=all(not r["missing_keys"] and r["last_line_is_whole_in_8KB_tail"] for r in rows),
           value_min=vals[0] if vals else None, value_median=vals[len(vals) // 2] if vals else None, value_max=vals[-1] if vals else None,
           windows_min=min(min(r["runs"]) for r in rows) if rows else None, windows_max=max(max(r["runs"]) for r in rows) if rows else None, runs=rows)
json.dump(out, open(os.path.join(ROOT, "profiles", "r06_first_process_repeat.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "runs"}))

#!/bin/bash
# Round 5: bench.py as the FIRST GPU process of a fresh lease (the driver's condition), then again with a rocm-smi poller
# running next to it (the driver samples rocm-smi every ~5 s while it times the bench).
# usage: tools/r05_first.sh <tag> [pytest]
tag=${1:-x}
mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_first_${tag}.json 2> gpurun_out/r05_first_${tag}.err
echo "first: rc=$?"
( while true; do rocm-smi --showuse --showpower --showclocks --json > /dev/null 2>&1; sleep 0.3; done ) &
poller=$!
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --pcie-frames 0 > gpurun_out/r05_smi_${tag}.json 2> gpurun_out/r05_smi_${tag}.err
echo "with smi poller: rc=$?"
kill $poller
python3 - <<PY
import json
for n in ("first","smi"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05_%s_${tag}.json" % n) if l.startswith("{")][-1])
        print(n, "value %.4g"%d["value"], "runs", ["%.3g"%r for r in d["runs"]], "cold %s"%d.get("value_cold"), "inst %.4g"%d["instrumented_window"]["value"], "host_enq %.4f"%d["host_enqueue_ms_per_step"])
        print("   batches", [(b["frames"], b["steady"], b["handoff"], b["early_lanes"], b["submit_us"], b["step_us"], b["done_at_ms"]) for b in d["batches"]])
        if d.get("roofline"): print("   roofline frac %.4f avg_us %.2f"%(d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))
    except Exception as e:
        print(n, "failed", e)
PY
if [ "$2" = "pytest" ]; then
  python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15
fi

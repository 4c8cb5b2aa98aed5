#!/bin/bash
# Round 6: the driver's condition, repeated -- `python3 bench.py --gpus 1 --steps 20 --warmup 5` as the FIRST GPU process of a
# fresh lease, N separate gpurun invocations (run from the dev container).  What is kept of each: the stdout exactly as the driver
# sees it (gpurun_out/r06_fp_<i>.out: its LAST line is what gets parsed) and the side file.  tools/r06_first_summary.py checks
# every last line the way a driver would (< 8 KB tail, json.loads, the contract's keys) and summarises the values.
N=${1:-5}
START=${2:-1}
for i in $(seq $START $((START + N - 1))); do
  for try in 1 2 3 4; do
    gpurun --timeout 600 -- "python3 bench.py --gpus 1 --steps 20 --warmup 5 --json-out gpurun_out/r06_fp_${i}_detail.json > gpurun_out/r06_fp_$i.out 2> gpurun_out/r06_fp_$i.err" > gpurun_out/r06_fp_$i.log 2>&1
    if grep -q "status=ok" gpurun_out/r06_fp_$i.log; then break; fi
    sleep 60
  done
done

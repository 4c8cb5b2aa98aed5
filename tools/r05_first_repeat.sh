#!/bin/bash
# Round 5: the driver's condition, repeated -- `python3 bench.py --gpus 1 --steps 20 --warmup 5` as the FIRST GPU process of a
# fresh lease, N separate gpurun invocations (run from the dev container); every line is kept in gpurun_out/r05_fp_<i>.json and
# tools/r05_first_summary.py turns them into profiles/r05_first_process_repeat.json.
N=${1:-8}
START=${2:-1}
for i in $(seq $START $((START + N - 1))); do
  for try in 1 2 3 4 5 6; do
    gpurun --timeout 600 -- "python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_fp_$i.json 2> gpurun_out/r05_fp_$i.err" > gpurun_out/r05_fp_$i.log 2>&1
    if grep -q "status=ok" gpurun_out/r05_fp_$i.log; then break; fi
    sleep 60
  done
done

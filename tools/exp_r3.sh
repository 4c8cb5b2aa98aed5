run() { # label steps warmup
  for i in 1 2 3; do timeout 300 python bench.py --steps $2 --warmup $3 --no-cpu-baseline --pcie-frames 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', $2, round(d['value']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3), round(d['host_enqueue_ms_per_step'],4))"; done
}
for st in "20 5" "240 12"; do
for q in 2 3 4 5 6; do
export GPU_MAX_HW_QUEUES=$q; run hwq$q $st; unset GPU_MAX_HW_QUEUES
done
done

run() { # splits
  for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --splits $1 --no-cpu-baseline --pcie-frames 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3), d['launches_per_frame'])"; done
}
for s in 2,6,6,6 2,6,6,3,3 2,6,6,4,2 2,6,6,5,1 2,3,3,3,3,3,3 2,6,3,3,3,3 1,1,6,6,6 2,6,6,2,2,2; do run $s; done

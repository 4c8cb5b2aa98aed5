timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pcie-frames 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['value_cold']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; done

timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -3
run() { timeout 200 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4))"; }
for i in 1 2 3; do run --steps 20 --warmup 5; done
run --steps 60 --warmup 12; run --steps 240 --warmup 16; run --steps 240 --warmup 16

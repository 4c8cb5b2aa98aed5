run() { timeout 200 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), end=' ')"; }
timeout 600 python -m pytest tests/test_configs_gpu.py -x -q --timeout 300 2>&1 | tail -1
sleep 15
echo "after pytest+sleep, no clock warm:"; for i in 1 2 3; do run --steps 20 --warmup 5 --clock-warm-ms 0; done; echo
sleep 15
echo "clock warm 400:"; for i in 1 2 3; do run --steps 20 --warmup 5; done; echo
sleep 15
echo "clock warm 100:"; for i in 1 2 3; do run --steps 20 --warmup 5 --clock-warm-ms 100; done; echo
sleep 15
echo "clock warm 1500:"; for i in 1 2 3; do run --steps 20 --warmup 5 --clock-warm-ms 1500; done; echo
echo "240 steps warm 0 / 400:"; run --steps 240 --warmup 16 --clock-warm-ms 0; run --steps 240 --warmup 16; echo

run() { timeout 300 python bench.py --no-cpu-baseline --pcie-frames 0 --no-extras --no-kernel-timing $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['config']['timed_batches'][:5], end=' | ')"; }
for ARGS in "--steps 20 --warmup 5" "--steps 240 --warmup 16" "--steps 240 --warmup 16 --batch 12" "--steps 60 --warmup 12" "--steps 60 --warmup 12 --batch 12" "--steps 20 --warmup 5 --batch 12"; do
  echo "== $ARGS"; for rep in 1 2 3; do run; done; echo
done
timeout 600 python -m pytest tests/test_batch_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -2

run() { timeout 300 python bench.py --no-cpu-baseline --pcie-frames 0 --no-extras $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['avg_launch_us'],1), end=' | ')"; }
for ARGS in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --rehearsal-ms 0 --clock-warm-ms 400" "--steps 20 --warmup 5 --rehearsal-ms 0" "--steps 20 --warmup 5 --rehearsal-ms 1500"; do
  echo "== $ARGS"; for rep in 1 2 3 4; do run; done; echo
done

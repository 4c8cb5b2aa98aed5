run() { timeout 200 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4))"; }
for v in 0 1 2; do export ROFT_EXP_MASK_PRIO=$v; echo prio $v; run --steps 20 --warmup 5; run --steps 20 --warmup 5;  run --steps 240 --warmup 16; run --steps 240 --warmup 16; done
unset ROFT_EXP_MASK_PRIO; export ROFT_NO_STREAM_PRIORITY=1; echo noprio;  run --steps 20 --warmup 5; run --steps 20 --warmup 5;  run --steps 240 --warmup 16; run --steps 240 --warmup 16

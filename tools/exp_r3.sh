run() { # lib steps warmup
  for i in 1 2 3; do ROFT_LIB_SO=build_ab/$1.so timeout 300 python bench.py --steps $2 --warmup $3 --no-cpu-baseline --pcie-frames 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', $2, round(d['value']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; done
}
for st in "20 5" "60 12" "240 12"; do
run nnhl $st; run featev2 $st
done

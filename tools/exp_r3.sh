# A/B of library variants inside one box: ROFT_LIB_SO=build_ab/<name>.so, three runs each at 20 / 60 / 240 steps
run() { # name steps warmup
  for i in 1 2 3; do ROFT_LIB_SO=build_ab/$1.so timeout 300 python bench.py --steps $2 --warmup $3 --no-cpu-baseline --pcie-frames 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', $2, round(d['value']), d.get('value_cold') and round(d['value_cold']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3), d['roofline'].get('measured_random_Gsectors_per_s'), d['roofline'].get('frac_of_measured_random_sector_rate'))"; done
}
for v in ${VARIANTS:-base featev}; do run $v 20 5; done
for v in ${VARIANTS:-base featev}; do run $v 60 12; done
for v in ${VARIANTS:-base featev}; do run $v 240 12; done

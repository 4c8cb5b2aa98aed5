# A/B of library builds inside ONE gpurun call (boxes differ by 10 - 30 %): VARIANTS="a b" name build_ab/<name>.so files made with
# tools/build_variant.sh (ROFT_SRC=<other tree> for a second source tree); three runs each at 20 / 60 / 240 steps.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'VARIANTS="base new" bash tools/exp_r3.sh > gpurun_out/ab.log 2>&1; cat gpurun_out/ab.log'
R=$(cd "$(dirname "$0")/.." && pwd)
run() { # lib steps warmup
  for i in 1 2 3; do ROFT_LIB_SO=$R/build_ab/$1.so timeout 300 python $R/bench.py --steps $2 --warmup $3 --no-cpu-baseline --pcie-frames 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', $2, round(d['value']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; done
}
for st in "20 5" "60 12" "240 12"; do for v in ${VARIANTS:-base}; do run $v $st; done; done

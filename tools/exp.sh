run() { timeout 200 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), end=' ')"; }
export ROFT_LIB_SO=$PWD/build_ab/s.so
echo "S=4 3 2"
for a in "--steps 240 --warmup 16" "--steps 20 --warmup 5"; do for rep in 1 2 3; do for S in 4 3 2; do ROFT_EXP_MASK_S=$S run $a; done; echo; done; done

# A/B of library builds inside ONE gpurun call (boxes differ by tens of percent): bash tools/ab.sh "<bench args>" a.so b.so ...
ARGS=$1; shift
run() { timeout 200 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), end=' ')"; }
for rep in 1 2 3; do for so in "$@"; do export ROFT_LIB_SO=$PWD/build_ab/$so; run; done; echo; done

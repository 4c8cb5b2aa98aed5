# A/B of (library, environment) pairs inside ONE gpurun call: bash tools/ab_mix.sh "<bench args>" "lib.so [VAR=val ...]" ...
ARGS=$1; shift
run() { set -- $1; so=$1; shift; env ROFT_LIB_SO=$PWD/build_ab/$so "$@" timeout 200 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), end=' ')"; }
for rep in 1 2 3; do for v in "$@"; do run "$v"; done; echo; done

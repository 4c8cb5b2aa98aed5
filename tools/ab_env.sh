# A/B of ENVIRONMENT settings on one build inside ONE gpurun call: bash tools/ab_env.sh "<bench args>" "A=1" "A=0 B=2" ...
ARGS=$1; shift
run() { env $1 timeout 200 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), end=' ')"; }
for rep in 1 2 3; do for v in "$@"; do run "$v"; done; echo; done

# A/B of (library, environment) pairs inside ONE gpurun call: bash tools/ab_mix.sh "<bench args>" "lib.so [VAR=val ...]" ...
# (one row per repetition, one column per pair; a run that fails prints ERR and the tail of its stderr)
ARGS=$1; shift
run() { set -- $1; so=$1; shift; env ROFT_LIB_SO=$PWD/build_ab/$so "$@" timeout 300 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing --json-out "" $ARGS 2>/tmp/ab_err.txt | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try: print(round(json.loads(t[-1])['value']), end=' ')
except Exception: print('ERR', end=' '); sys.stderr.write(open('/tmp/ab_err.txt').read()[-600:])
"; }
for rep in 1 2 3; do for v in "$@"; do run "$v"; done; echo; done

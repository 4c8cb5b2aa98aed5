O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_multirank_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 900 python tools/run_baseline_configs.py --out $O/baseline_configs.json > $O/baseline.log 2>&1; tail -3 $O/baseline.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3g/baseline_configs.json"))
for k,v in d.items():
    if isinstance(v, dict): print(k, {a:b for a,b in v.items() if not isinstance(b, dict)}, {a:b for a,b in v.items() if isinstance(b, dict)})
PY

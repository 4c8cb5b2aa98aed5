timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; done

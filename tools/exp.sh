run() { timeout 200 python bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), end=' ')"; }
echo "speculate 0 1 2(auto)"
for a in "--steps 20 --warmup 5" "--steps 60 --warmup 12" "--steps 240 --warmup 16"; do for rep in 1 2 3; do for sp in 0 1 2; do ROFT_SPECULATE=$sp run $a; done; echo; done; done
for sp in 0 1; do ROFT_SPECULATE=$sp timeout 300 python tools/live_latency.py --out /tmp/ll_$sp.json > /dev/null 2>&1; python -c "
import json; l=json.load(open('/tmp/ll_$sp.json'))
print('live speculate=$sp', {k:(round(v['median_us']), round(v['median_us_pose_frames'])) for k,v in l.items() if isinstance(v,dict)})"; done

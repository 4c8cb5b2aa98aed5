R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest_final.log 2>&1; tail -2 $O/gputest_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --json-out $O/bench_driver_shaped_${i}_detail.json > $O/bench_driver_shaped_$i.json 2> $O/bench.err; done
timeout 300 python bench.py --json-out $O/bench_64obj_detail.json > $O/bench_64obj.json 2>> $O/bench.err
timeout 300 python bench.py --steps 240 --warmup 16 --windows 3 --no-cpu-baseline --pcie-frames 0 --no-extras --json-out $O/bench_steady_240_detail.json > $O/bench_steady_240.json 2>> $O/bench.err
python - > $O/object_sweep.json <<PY
import json, subprocess, sys
out = []
for n in (8, 16, 32, 64, 128, 256):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "60", "--warmup", "12", "--objects", str(n), "--windows", "3" if n <= 64 else "1",
                        "--no-cpu-baseline", "--pcie-frames", "0", "--no-extras", "--json-out", "/tmp/roft_sweep_detail.json"], capture_output=True, text=True, timeout=600)
    d = json.load(open("/tmp/roft_sweep_detail.json"))
    out.append(dict(objects=n, value=d["value"], runs=d["runs"], ms_per_step=d["ms_per_step"], frames_per_sec_per_object=d["frames_per_sec_per_object"],
                    k1_avg_launch_us=d["roofline"]["avg_launch_us"], roofline_frac=d["roofline"]["frac"], launches_per_frame=d["launches_per_frame"],
                    kernels=d["kernels_post_run_breakdown"]))
json.dump(dict(what="python bench.py --steps 60 --warmup 12 --objects N (one MI355X): the per-GPU load of config #4 sharded over 8 / 4 / 2 / 1 GPUs is 8 / 16 / 32 / 64 objects", runs=out), sys.stdout, indent=1)
PY
python - > $O/object_sweep_20.json <<PY
import json, subprocess, sys
out = []
for n in (8, 16, 32, 64):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--objects", str(n), "--no-cpu-baseline", "--pcie-frames", "0", "--no-extras",
                        "--json-out", "/tmp/roft_sweep_detail.json"], capture_output=True, text=True, timeout=600)
    d = json.load(open("/tmp/roft_sweep_detail.json"))
    out.append(dict(objects=n, values=d["runs"], median=d["value"], frames_per_sec_per_object=d["value"] / n, ms_per_step=d["ms_per_step"]))
json.dump(dict(what="python bench.py --steps 20 --warmup 5 --objects N (one MI355X): value = median of the run's five timed windows (values)", runs=out), sys.stdout, indent=1)
PY
for f in bench_driver_shaped_1 bench_driver_shaped_2 bench_driver_shaped_3 bench_64obj bench_steady_240; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$f', round(d['value']), [round(v) for v in d['runs']], r.get('frac'), r.get('avg_launch_us'))"; done
python -c "
import json
for f in ('object_sweep_20','object_sweep'):
    d=json.load(open('$O/'+f+'.json')); print(f, [(r['objects'], round(r.get('median', r.get('value')))) for r in d['runs']])"

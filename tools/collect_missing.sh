#!/bin/bash
# the pieces of tools/collect_profiles.sh that failed in the first round-5 collection (memory check, launch indices)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05
mkdir -p $O; cd $R
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_shaped_$i.json 2>> $O/bench.err; done
timeout 300 python bench.py > $O/bench_64obj.json 2>> $O/bench.err
timeout 600 python bench.py --steps 240 --warmup 16 --windows 3 --no-cpu-baseline --pcie-frames 0 --no-extras > $O/bench_steady_240.json 2>> $O/bench.err
python - > $O/object_sweep.json <<PY
import json, subprocess, sys
out = []
for n in (8, 16, 32, 64, 128, 256):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "60", "--warmup", "12", "--objects", str(n), "--windows", "3" if n <= 64 else "1",
                        "--no-cpu-baseline", "--pcie-frames", "0", "--no-extras"], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        out.append(dict(objects=n, error=r.stderr[-300:]))
        continue
    d = json.loads(lines[-1])
    out.append(dict(objects=n, value=d["value"], runs=d["runs"], ms_per_step=d["ms_per_step"], frames_per_sec_per_object=d["frames_per_sec_per_object"],
                    k1_avg_launch_us=d["roofline"]["avg_launch_us"], roofline_frac=d["roofline"]["frac"], launches_per_frame=d["launches_per_frame"],
                    kernels=d["kernels_post_run_breakdown"]))
json.dump(dict(what="python bench.py --steps 60 --warmup 12 --objects N (one MI355X): the per-GPU load of config #4 sharded over 8 / 4 / 2 / 1 GPUs is 8 / 16 / 32 / 64 objects", runs=out), sys.stdout, indent=1)
PY
python - > $O/object_sweep_20.json <<PY
import json, subprocess, sys
out = []
for n in (8, 16, 32, 64):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--objects", str(n), "--no-cpu-baseline", "--pcie-frames", "0", "--no-extras"],
                       capture_output=True, text=True, timeout=600)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    out.append(dict(objects=n, values=d["runs"], median=d["value"], frames_per_sec_per_object=d["value"] / n, ms_per_step=d["ms_per_step"]))
json.dump(dict(what="python bench.py --steps 20 --warmup 5 --objects N (one MI355X): value = median of the run's five timed windows (values)", runs=out), sys.stdout, indent=1)
PY
rm -f $O/marks_20.txt $O/marks_240.txt
ROFT_BENCH_FULL_TIMING=1 ROFT_DUMP_MARKS=$O/marks_20.txt ROFT_BENCH_EXTRA_FRAMES=0 timeout 300 python bench.py --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --pcie-frames 0 --no-extras > /dev/null 2>> $O/bench.err
python tools/marks_timeline.py $O/marks_20.txt --list > $O/marks_timeline_20.txt
ROFT_BENCH_FULL_TIMING=1 ROFT_DUMP_MARKS=$O/marks_240.txt ROFT_BENCH_EXTRA_FRAMES=0 timeout 600 python bench.py --steps 240 --warmup 16 --windows 1 --no-cpu-baseline --pcie-frames 0 --no-extras > /dev/null 2>> $O/bench.err
python tools/marks_timeline.py $O/marks_240.txt --from 4000 --to 9000 > $O/marks_timeline_240.txt
python tools/marks_timeline.py $O/marks_240.txt --from 6000 --to 7400 --list | tail -n +16 >> $O/marks_timeline_240.txt
rm -f $O/marks_20.txt $O/marks_240.txt
ROFT_BENCH_DEVICE=0 ROFT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_one_gpu_gloo.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --pcie-frames 0 --no-extras --rehearsal-ms 0 > $O/bench_under_rocprof.json 2> /dev/null
python3 $R/tools/prof_summary.py stats $O/stats/*/*kernel_stats.csv $O/bench_kernel_stats.csv
python3 $R/tools/prof_summary.py window $O/stats/*/*kernel_trace.csv flow_measure_kernel 8 4 > $O/k1_timed_launches_under_rocprof.txt
python3 $R/tools/trace_list.py $O/stats/*/*kernel_trace.csv 2 --resources > $O/pipeline_timeline.txt
rm -rf $O/stats
cd $R; ls -la $O | awk '{print $5, $9}'

#!/bin/bash
# Collects the rocprofv3 / bench evidence of a round on the GPU box and condenses it into small files under
# gpurun_out/<tag>/ (copy the ones to keep into profiles/ with the tag as prefix).   usage: bash tools/collect_profiles.sh [tag]
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
# the driver-shaped run (what BENCH_rNN.json records) three times, the default run and a long steady-state run
# (bench.py prints the compact line the driver parses; the full record of a run is its --json-out side file: *_detail.json)
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --json-out $O/bench_driver_shaped_${i}_detail.json > $O/bench_driver_shaped_$i.json 2> $O/bench.err; done
timeout 300 python bench.py --json-out $O/bench_64obj_detail.json > $O/bench_64obj.json 2>> $O/bench.err
timeout 300 python bench.py --steps 240 --warmup 16 --windows 3 --no-cpu-baseline --pcie-frames 0 --no-extras --json-out $O/bench_steady_240_detail.json > $O/bench_steady_240.json 2>> $O/bench.err
timeout 900 python tools/run_baseline_configs.py --out $O/baseline_configs.json > /dev/null 2> $O/baseline.err
# objects per GPU: the per-GPU points of the 8 / 4 / 2 / 1-GPU strong-scaling curve of config #4 (64 / 32 / 16 / 8 objects) and beyond
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
# ... and the same points in the DRIVER's shape (--steps 20 --warmup 5): what a strong-scaling run of config #4 over 8 / 4 / 2 / 1 GPUs puts on one GPU
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
# the reference's executable over this engine: microseconds per frame at 1280x720, images read in place / staged
timeout 600 python tools/tracker_timing.py 120 --out $O/tracker_timing.json > /dev/null 2>> $O/bench.err
# profiler-free timelines (HIP event marks after every launch group): the driver-shaped window and the steady state
rm -f $O/marks_20.txt $O/marks_240.txt
ROFT_BENCH_FULL_TIMING=1 ROFT_DUMP_MARKS=$O/marks_20.txt ROFT_BENCH_EXTRA_FRAMES=0 timeout 300 python bench.py --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --pcie-frames 0 --no-extras > /dev/null 2>> $O/bench.err
python tools/marks_timeline.py $O/marks_20.txt --list > $O/marks_timeline_20.txt
ROFT_BENCH_FULL_TIMING=1 ROFT_DUMP_MARKS=$O/marks_240.txt ROFT_BENCH_EXTRA_FRAMES=0 timeout 300 python bench.py --steps 240 --warmup 16 --windows 1 --no-cpu-baseline --pcie-frames 0 --no-extras > /dev/null 2>> $O/bench.err
python tools/marks_timeline.py $O/marks_240.txt --from 4000 --to 9000 > $O/marks_timeline_240.txt
python tools/marks_timeline.py $O/marks_240.txt --from 6000 --to 7400 --list | tail -n +16 >> $O/marks_timeline_240.txt
rm -f $O/marks_20.txt $O/marks_240.txt
# the N > 1 code path executed: two ranks on this one GPU over gloo (the driver's multi-GPU runs use one GPU per rank and RCCL)
ROFT_BENCH_DEVICE=0 ROFT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --json-out $O/bench_2ranks_one_gpu_gloo_detail.json > $O/bench_2ranks_one_gpu_gloo.json 2>> $O/bench.err
ROFT_BENCH_DEVICE=0 ROFT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 20 --warmup 5 --shared-scene --json-out $O/bench_2ranks_one_gpu_gloo_shared_scene_detail.json > $O/bench_2ranks_one_gpu_gloo_shared_scene.json 2>> $O/bench.err
timeout 300 python tools/live_latency.py --out $O/live_latency.json > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
# kernel stats + timeline of the driver-shaped run, chains overlapping
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --windows 1 --k1-windows 1 --no-cpu-baseline --pcie-frames 0 --no-extras --rehearsal-ms 0 --json-out $O/bench_under_rocprof_detail.json > $O/bench_under_rocprof.json 2> /dev/null
python3 $R/tools/prof_summary.py stats $O/stats/*/*kernel_stats.csv $O/bench_kernel_stats.csv
# the roofline kernel over exactly the launches that bench.py's event pairs time in that run: --windows 1 = window 0 (2 warm-up
# batches (frames 0 | 1 - 4) + 4 timed), then ONE instrumented window (--k1-windows 1; 2 warm-up batches + 4 TIMED: launches 8 .. 11), then the
# breakdown frames:
# compare with roofline.avg_launch_us of bench_under_rocprof.json
python3 $R/tools/prof_summary.py window $O/stats/*/*kernel_trace.csv flow_measure_kernel 8 4 > $O/k1_timed_launches_under_rocprof.txt
python3 $R/tools/trace_list.py $O/stats/*/*kernel_trace.csv 2 --resources > $O/pipeline_timeline.txt
rm -rf $O/stats
# the same kernels with the chains serialised on one stream (each kernel's duration alone), on a longer run
export ROFT_ONE_STREAM=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $R/bench.py --steps 48 --warmup 7 --windows 1 --no-cpu-baseline --pcie-frames 0 --no-extras --rehearsal-ms 0 --no-kernel-timing > /dev/null 2>&1
unset ROFT_ONE_STREAM
python3 $R/tools/prof_summary.py stats $O/stats1/*/*kernel_stats.csv $O/bench_kernel_stats_one_stream.csv
rm -rf $O/stats1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python3 $R/bench.py --steps 48 --warmup 7 --windows 1 --no-cpu-baseline --pcie-frames 0 --no-extras --rehearsal-ms 0 --no-kernel-timing > /dev/null 2>&1
python3 $R/tools/prof_summary.py stats $O/stats2/*/*kernel_stats.csv $O/bench_kernel_stats_48.csv
# CU x us budget of that run (48 timed + 7 warm-up + 24 breakdown frames x 64 objects; footprints: profiles/${TAG}_kernel_resources.csv,
# made without a GPU by tools/kernel_resources.sh)
[ -f $R/profiles/${TAG}_kernel_resources.csv ] && python3 $R/tools/cu_budget.py $O/stats2/*/*kernel_trace.csv $R/profiles/${TAG}_kernel_resources.csv $((64 * 55)) > $O/cu_budget.csv
rm -rf $O/stats2
# HBM traffic, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass); every launch of the roofline kernel
# covers 8 frames x 64 objects in this run (24 timed frames after 8 warm-up frames, batches of 8)
# + the L2's view of the same launches: hits, misses and read requests to the fabric (K1's gathers miss: one request each)
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc -- python3 $R/bench.py --steps 24 --warmup 8 --windows 1 --no-align --no-cpu-baseline --pcie-frames 0 --no-extras --rehearsal-ms 0 --no-kernel-timing > /dev/null 2>&1
  python3 $R/tools/prof_summary.py pmc $O/pmc/*/*counter_collection.csv $O/pmc_$c.csv
  rm -rf $O/pmc
done
python3 - <<PY
import csv, json
rows = {r["kernel"].split("<")[0].split("::")[-1]: r for r in csv.DictReader(open("$O/pmc_FETCH_SIZE.csv"))}
k1 = rows["flow_measure_kernel"]
per_launch_kb = float(k1["mean_value_per_dispatch"])
obj_frames = 64 * 8
plane = 640 * 480 // 8
raw = per_launch_kb * 1024.0 / obj_frames
json.dump({"objects": 64, "shape": "A", "flow": "f32", "batch": 8, "dispatches": int(k1["dispatches"]),
           "fetch_size_kb_per_launch": per_launch_kb, "object_frames_per_launch": obj_frames,
           "fetch_bytes_per_object_frame_raw": raw,
           # MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies the 128-byte requests of a wide coalesced 16 B / lane streaming read
           # at 64 bytes -- the bit-plane read of this kernel is such a stream and is counted twice here; the 4- and 8-byte
           # sample gathers are left as counted (uncalibrated)
           "fetch_bytes_per_object_frame": raw + plane,
           "source": "rocprofv3 --pmc FETCH_SIZE, bench.py --steps 24 --warmup 8 --windows 1 --no-align (tools/collect_profiles.sh), KB x 1024 / 512 object-frames per launch + 38400 B plane correction"},
          open("$O/pmc_k1.json", "w"), indent=1)
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/of -- python3 $R/tools/bench_flow_producer.py > /dev/null 2>&1
python3 $R/tools/prof_summary.py stats $O/of/*/*kernel_stats.csv $O/flow_producer_kernel_stats.csv
rm -rf $O/of
cd $R
timeout 300 python tools/bench_flow_producer.py > $O/flow_producer.jsonl
du -sh $O; ls -la $O

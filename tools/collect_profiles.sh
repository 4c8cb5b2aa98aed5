#!/bin/bash
# Collects the rocprofv3 / bench evidence of a round on the GPU box and condenses it into small files under
# gpurun_out/<tag>/ (copy the ones to keep into profiles/).   usage: bash tools/collect_profiles.sh [tag]
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench_64obj.json 2> $O/bench.err
python tools/run_baseline_configs.py --out $O/baseline_configs.json > /dev/null 2> $O/baseline.err
python tools/live_latency.py --out $O/live_latency.json > /dev/null 2>&1
python tools/bench_flow_producer.py > $O/flow_producer.jsonl
python tools/bench_flow_producer.py --pairs 1 >> $O/flow_producer.jsonl
python tools/bench_flow_producer.py --pairs 16 --shape B --flow s16 >> $O/flow_producer.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> /dev/null
python3 $R/tools/prof_summary.py stats $O/stats/*/*kernel_stats.csv $O/bench_kernel_stats.csv
# the roofline kernel over exactly the timed launches of that run (12 warm-up frames, then 60): compare with roofline.avg_launch_us
# of bench_under_rocprof.json
python3 $R/tools/prof_summary.py window $O/stats/*/*kernel_trace.csv flow_measure_kernel 12 60 > $O/k1_timed_launches_under_rocprof.txt
python3 $R/tools/trace_timeline.py $O/stats/*/*kernel_trace.csv 600 > $O/pipeline_timeline.txt
python3 $R/tools/trace_timeline.py $O/stats/*/*kernel_trace.csv 120 --list | head -130 >> $O/pipeline_timeline.txt
rm -rf $O/stats
# the same kernels with the three chains serialised on one stream (the "alone" durations quoted in DESIGN.md section 5)
export ROFT_ONE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>&1
unset ROFT_ONE_STREAM
python3 $R/tools/prof_summary.py stats $O/stats1/*/*kernel_stats.csv $O/bench_kernel_stats_one_stream.csv
rm -rf $O/stats1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  python3 $R/tools/prof_summary.py pmc $O/pmc/*/*counter_collection.csv $O/pmc_$c.csv
  rm -rf $O/pmc
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/of -- python3 $R/tools/bench_flow_producer.py > /dev/null 2>&1
python3 $R/tools/prof_summary.py stats $O/of/*/*kernel_stats.csv $O/flow_producer_kernel_stats.csv
rm -rf $O/of
du -sh $O; ls -la $O

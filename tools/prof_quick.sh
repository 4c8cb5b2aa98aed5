R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/p1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 24 --warmup 6 --batch 6 --no-cpu-baseline --pcie-frames 0 > $O/bench.json 2>/dev/null
python3 $R/tools/prof_summary.py stats $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
export ROFT_ONE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $R/bench.py --steps 24 --warmup 6 --batch 6 --no-cpu-baseline --pcie-frames 0 > $O/bench1.json 2>/dev/null
python3 $R/tools/prof_summary.py stats $O/stats1/*/*kernel_stats.csv $O/kernel_stats_one_stream.csv
rm -rf $O/stats $O/stats1

# kernel timeline of one bench run: bash tools/prof_trace.sh <tag> <first batch listed> [bench args...]
R=$GRAFT_REPO_ROOT; TAG=$1; FIRST=$2; shift; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --pcie-frames 0 --no-kernel-timing "$@" > $O/bench.json 2>/dev/null
python3 $R/tools/prof_summary.py stats $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
python3 $R/tools/trace_list.py $O/stats/*/*kernel_trace.csv $FIRST --resources > $O/timeline.txt
rm -rf $O/stats

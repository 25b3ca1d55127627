#!/bin/bash
# rocprofv3 kernel trace of the bench (one update timeline + per-kernel stats).  Usage: tools/gpu_trace.sh <tag> [bench args]
TAG=${1:-t}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras "$@" > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 tools/trace_summary.py $OUT/kernel_trace.csv > $OUT/timeline.txt; tail -n 130 $OUT/timeline.txt

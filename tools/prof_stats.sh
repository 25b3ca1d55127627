#!/bin/bash
# usage: tools/prof_stats.sh <tag> <script.py> [args]: rocprofv3 kernel stats (avg duration per kernel) of a python script
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 "$@" > $OUT/run.log 2>&1
echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:14]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us avg  x{r['Calls']:>5}  {r['Name'][:110]}")
PY

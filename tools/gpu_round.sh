#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Usage: tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$OUT'/kernel_stats.csv; head -40 {}'

#!/bin/bash
# One GPU-box visit: parity tests, bench line, multi-rank rehearsal.  Usage: tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
rc=$?; echo "bench rc=$rc"; cat $OUT/bench.json; tail -3 $OUT/bench.err
[ $rc -ne 0 ] && exit $rc
# two ranks on the one GPU, gloo carrying the exchanges: rehearses the self-launch and the N>1 line
timeout -k 10 300 python bench.py --gpus 2 --devices 0,0 --backend gloo --steps 20 --warmup 5 --no-roofline > $OUT/bench_dp2.json 2> $OUT/bench_dp2.err
echo "bench dp2 rc=$?"; cat $OUT/bench_dp2.json; tail -3 $OUT/bench_dp2.err

#!/bin/bash
# tests + per-kernel microbench + bench line.  Usage: tools/gpu_quick.sh <tag>
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout -k 10 200 python tools/kernel_bench.py > $OUT/kb.log 2>&1; grep -v "^{" $OUT/kb.log | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-420 $OUT/bench.json

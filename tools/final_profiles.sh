#!/bin/bash
# End-of-round evidence in one box visit: bench line, timeline + stats, counters, per-config numbers, bf16 line, N>1
# rehearsal.  Usage: tools/final_profiles.sh <tag>   (outputs under gpurun_out/<tag>*)
TAG=${1:-fin}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench_line.json
tools/gpu_trace.sh ${TAG}_trace > $OUT/trace.out 2>&1; echo "trace rc=$?"; tail -1 $OUT/trace.out
tools/pmc_bench.sh ${TAG}_pmc > $OUT/pmc.out 2>&1; echo "pmc rc=$?"
timeout -k 10 300 python bench.py --dtype bf16 --task humanoid_run --batch 2048 --steps 60 --warmup 10 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "bf16 rc=$?"; cut -c1-200 $OUT/bench_bf16.json
timeout -k 10 300 python bench.py --gpus 2 --devices 0,0 --backend gloo --steps 20 --warmup 5 --no-roofline > $OUT/bench_dp2.json 2> $OUT/bench_dp2.err; echo "dp2 rc=$?"
X="--no-cpu-baseline --no-roofline --no-extras --steps 60 --warmup 10"
for spec in "cartpole_swingup 32 f32" "cheetah_run 256 f32" "quadruped_walk 512 f32" "humanoid_run 256 f32" "humanoid_run 32 f32" "humanoid_run 2048 f32" "humanoid_run 2048 bf16"; do
  set -- $spec
  timeout -k 10 200 python bench.py --task $1 --batch $2 --dtype $3 $X > $OUT/cfg_$1_$2_$3.json 2> $OUT/cfg_$1_$2_$3.err
  python - <<PY >> $OUT/configs.txt
import json
b=json.load(open("$OUT/cfg_$1_$2_$3.json"))
print("%-18s B=%-5s %-5s %8.4f ms/update %8.1f batch-256 equivalents/s" % ("$1", "$2", "$3", b["ms_per_step"], b["value"]))
PY
done
for mode in "" "--device-replay" "--device-replay copy"; do
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-extras $mode > $OUT/feed.json 2> $OUT/feed.err
  python - <<PY >> $OUT/configs.txt
import json
b=json.load(open("$OUT/feed.json"))
print("feed %-22s %8.1f updates/s %8.4f ms" % ("$mode" or "resident batch", b["value"], b["ms_per_step"]))
PY
done
cat $OUT/configs.txt

set -e
X="--no-cpu-baseline --no-roofline --steps 100 --warmup 10"
for spec in "cartpole_swingup 32" "quadruped_walk 512" "humanoid_run 256" "cheetah_run 1024" "humanoid_run 2048" "cheetah_run 64"; do
  set -- $spec
  python bench.py --task $1 --batch $2 $X > gpurun_out/cfg_$1_$2.json 2> gpurun_out/cfg_$1_$2.err
  python - <<PY
import json
b=json.load(open("gpurun_out/cfg_$1_$2.json"))
print("$1 B=$2: %.3f ms/update, %.1f batch-256 equivalents/s, frac %.3f" % (b["ms_per_step"], b["value"], b["frac_fp32_peak_whole_step"]))
PY
done
python bench.py --device-replay --no-cpu-baseline --no-roofline > gpurun_out/cfg_devreplay.json 2>gpurun_out/cfg_devreplay.err; cut -c60-140 gpurun_out/cfg_devreplay.json
python bench.py --host-batch --no-cpu-baseline --no-roofline > gpurun_out/cfg_hostbatch.json 2>gpurun_out/cfg_hostbatch.err; cut -c60-140 gpurun_out/cfg_hostbatch.json

mkdir -p gpurun_out/fin
tools/gpu_trace.sh fin_f32 > /dev/null 2>&1; cp gpurun_out/fin_f32/timeline.txt gpurun_out/fin/timeline_f32.txt; cp gpurun_out/fin_f32/kernel_stats.csv gpurun_out/fin/kernel_stats_f32.csv
tools/gpu_trace.sh fin_bf16 --dtype bf16 --task humanoid_run --batch 2048 > /dev/null 2>&1; cp gpurun_out/fin_bf16/timeline.txt gpurun_out/fin/timeline_bf16_b2048.txt; cp gpurun_out/fin_bf16/kernel_stats.csv gpurun_out/fin/kernel_stats_bf16_b2048.csv
DRQ_COMMIT=$(cat .commit_for_gpurun) tools/pmc_bench.sh pmc_bf16 --dtype bf16 > /dev/null 2>&1; cp gpurun_out/pmc_bf16/summary.txt gpurun_out/fin/pmc_bf16_b256.txt
timeout -k 10 300 python bench.py > gpurun_out/fin/bench_f32.json 2> /dev/null; cut -c1-260 gpurun_out/fin/bench_f32.json
timeout -k 10 300 python bench.py --task humanoid_run --batch 2048 --dtype bf16 > gpurun_out/fin/bench_bf16_b2048.json 2> /dev/null; cut -c1-260 gpurun_out/fin/bench_bf16_b2048.json
for t in "cartpole_swingup 32" "quadruped_walk 512" "humanoid_run 256" "humanoid_run 32" "cheetah_run 64"; do set -- $t; timeout -k 10 200 python bench.py --task $1 --batch $2 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 B=$2', round(d['ms_per_step'],4), 'ms/update', round(d['value'],1), 'batch-256 equiv/s', round(d['frac_fp32_peak_whole_step'],3))"; done
timeout -k 10 200 python bench.py --device-replay --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | cut -c1-120
timeout -k 10 200 python bench.py --host-batch --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | cut -c1-120

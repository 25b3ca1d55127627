set -o pipefail
mkdir -p gpurun_out/bf
timeout -k 10 600 python -m pytest tests/test_hip_bf16.py -q -s -k "update" > gpurun_out/bf/t.log 2>&1; rc=$?; grep -E "bf16:|passed|failed|Error|assert" gpurun_out/bf/t.log | tail -12; [ $rc -ne 0 ] && tail -30 gpurun_out/bf/t.log
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --dtype bf16 > gpurun_out/bf/b256.json 2> gpurun_out/bf/b256.err; echo "rc=$?"; cut -c1-330 gpurun_out/bf/b256.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --dtype bf16 --task humanoid_run --batch 2048 > gpurun_out/bf/b2048.json 2> gpurun_out/bf/b2048.err; echo "rc=$?"; cut -c1-330 gpurun_out/bf/b2048.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --task humanoid_run --batch 2048 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/bf/f2048.json 2> gpurun_out/bf/f2048.err; echo "rc=$?"; cut -c1-330 gpurun_out/bf/f2048.json

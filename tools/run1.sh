set -o pipefail
mkdir -p gpurun_out/c1
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "fused_aug or aug_vs" > gpurun_out/c1/ops.log 2>&1; rc=$?; tail -5 gpurun_out/c1/ops.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/conv1aug_ab.py 256 2>&1 | tail -9
timeout -k 10 300 python tools/conv1aug_ab.py 32 2>&1 | grep "variant  0"
timeout -k 10 600 python -m pytest tests/test_hip_step.py -q -m gpu -x -k "cheetah_b8 or cheetah_b256 or cartpole" > gpurun_out/c1/step.log 2>&1; rc=$?; tail -5 gpurun_out/c1/step.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/c1/bench.json

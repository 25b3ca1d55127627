mkdir -p gpurun_out/bf
timeout -k 10 300 python -m pytest tests/test_hip_bf16.py -q -k "gemm" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --dtype bf16 2> /dev/null | cut -c1-200
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --dtype bf16 --task humanoid_run --batch 2048 2> /dev/null | cut -c1-200

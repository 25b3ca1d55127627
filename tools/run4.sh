mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_hip_step.py -q -x -k "cheetah_b8 or cheetah_b256 or humanoid_b128 or small_h64 or bit_stable" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2> /dev/null | cut -c1-230

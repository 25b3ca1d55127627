OUT=gpurun_out/w3; mkdir -p $OUT
timeout -k 10 200 python tools/wino_bench.py 256 > $OUT/wb.log 2>&1; echo "wb rc=$?"; grep -v amdgpu.ids $OUT/wb.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_hip_step.py -m gpu -q -x --timeout 400 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log

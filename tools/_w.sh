OUT=gpurun_out/w20; mkdir -p $OUT
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-roofline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c60-200 $OUT/bench.json
timeout -k 10 700 python -m pytest tests/test_hip_ops.py tests/test_hip_step.py -m gpu -q -x --timeout 400 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log

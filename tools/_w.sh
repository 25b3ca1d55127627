OUT=gpurun_out/w15; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x --timeout 200 -k "winograd or training_batch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 200 python tools/wino_bench.py 256 > $OUT/wb.log 2>&1; echo "wb rc=$?"; grep -v amdgpu.ids $OUT/wb.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-roofline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c60-200 $OUT/bench.json
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-roofline > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench rc=$?"; cut -c60-200 $OUT/bench2.json
timeout -k 10 500 python -m pytest tests/test_hip_step.py -m gpu -q -x --timeout 400 -k "oracle" > $OUT/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -3 $OUT/pytest2.log

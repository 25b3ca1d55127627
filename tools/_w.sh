OUT=gpurun_out/w28; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x --timeout 200 -k "linear or gemm or trunk or skinny" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/gpu_trace.sh t28 > gpurun_out/t28.log 2>&1; grep -E "trunk_fwd|sum of kernel" gpurun_out/t28.log | head -6
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-roofline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c60-200 $OUT/bench.json
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-roofline --task humanoid_run > $OUT/benchh.json 2> $OUT/benchh.err; echo "benchh rc=$?"; cut -c60-200 $OUT/benchh.json

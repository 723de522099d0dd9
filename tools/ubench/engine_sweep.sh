mkdir -p gpurun_out/r31
(timeout 600 python -m pytest tests/test_gpu_w4.py tests/test_gpu_engine.py -q -x 2>&1 | tail -4) > gpurun_out/r31/test.log 2>&1
timeout 300 python tools/bench_engine.py --reps 20 --variants i8p,i8p+fused 2>&1 | tail -4 > gpurun_out/r31/bench.log
cat gpurun_out/r31/test.log gpurun_out/r31/bench.log

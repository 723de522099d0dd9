mkdir -p gpurun_out/r30
(timeout 300 python -m pytest tests/test_gpu_engine.py -q 2>&1 | tail -4) > gpurun_out/r30/test.log 2>&1
(timeout 400 python tools/bench_engine.py --reps 20 --no-ops --variants i8p,engine+fused,i8p+fused 2>&1 | tail -5) > gpurun_out/r30/bench.log 2>&1
(ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libeprobe.so timeout 300 python tools/ubench/probe_engine.py 1 > gpurun_out/r30/probe.log 2>&1)
cat gpurun_out/r30/test.log gpurun_out/r30/bench.log

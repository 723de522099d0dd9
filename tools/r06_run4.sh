set -x
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests/test_gpu_w4.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r06/slab_tests.txt
cat gpurun_out/r06/slab_tests.txt
timeout 600 python -m pytest tests/test_gpu_hostcpp.py tests/test_gpu_refcompile.py -x -q -m gpu -k "legacy or int4gptq" 2>&1 | tail -25 > gpurun_out/r06/a6_tests.txt
cat gpurun_out/r06/a6_tests.txt
timeout 1500 python tools/bench_slab.py --m 32 16 9 > gpurun_out/r06/slab_sweep_v2.txt 2>&1
cat gpurun_out/r06/slab_sweep_v2.txt | grep -v amdgpu.ids

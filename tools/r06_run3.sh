set -x
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_w4.py -x -q -m gpu -k "slab or fused_qkv or silu or mfma_gemm" 2>&1 | tail -25 > gpurun_out/r06/slab_tests.txt
cat gpurun_out/r06/slab_tests.txt
timeout 1200 python tools/bench_slab.py --m 32 16 9 > gpurun_out/r06/slab_sweep_v2.txt 2>&1
cat gpurun_out/r06/slab_sweep_v2.txt | grep -v amdgpu.ids

set -x
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_w4.py -x -q -m gpu -k "slab or fused_qkv or silu or deferred or phase_gemm or mfma_gemm or streaming or qwen2" 2>&1 | tail -15 > gpurun_out/r06/slab_tests.txt
cat gpurun_out/r06/slab_tests.txt
{
for m in 32 16 9; do
  echo "== default plan M=$m"; timeout 300 python tools/bench_gemv.py --mfma --m $m --layers 8 2>&1 | grep -v amdgpu.ids | head -8
done
echo "== phase kernel (ZL_W4_SLAB=-1) M=32"; ZL_W4_SLAB=-1 timeout 300 python tools/bench_gemv.py --mfma --m 32 --layers 8 2>&1 | grep -v amdgpu.ids | head -6
for geom in "4 1" "4 2" "4 4" "8 1" "8 2" "8 4"; do
  set -- $geom
  echo "== forced nw=$1 gpw=$2 M=32"; ZL_W4_SLAB_NW=$1 ZL_W4_SLAB_GPW=$2 timeout 300 python tools/bench_gemv.py --mfma --m 32 --layers 8 2>&1 | grep -v amdgpu.ids | head -6
done
} > gpurun_out/r06/slab_sweep.txt 2>&1
cat gpurun_out/r06/slab_sweep.txt | grep -v "^+"

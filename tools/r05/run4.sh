# round 5, GPU call 4: INT8 route with the reference's sum-of-squares order; ZL_I8P_EARLY variants of the batch-1 step
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
rm -f gpurun_out/r05_early.txt
for v in "" early1 early2 early4 ""; do
  so=""; [ -n "$v" ] && so="$PWD/zhilight_amd/build/variants/lib$v.so"
  echo "== variant ${v:-default}" >> gpurun_out/r05_early.txt
  ZHILIGHT_AMD_SO=$so BATCHES=1,2,4 REPS=60 timeout 300 python tools/ab_step.py base 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_early.txt
done
cat gpurun_out/r05_early.txt
( ZHILIGHT_AMD_SO=$PWD/zhilight_amd/build/variants/libearly2.so timeout 600 python -m pytest tests/test_gpu_w4.py -x -q -k "i8p or attn_merge or fused_qkv_rotary" 2>&1 | tail -4 ) > gpurun_out/r05_t_early2.txt
rm -f gpurun_out/parity_fullgeom.jsonl
( timeout 1200 python -m pytest tests/test_gpu_fullgeom.py tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_kvquant.py -x -q -s -k "int8 or w8a8" 2>&1 | tail -14 ) > gpurun_out/r05_t_int8.txt
cp gpurun_out/parity_fullgeom.jsonl gpurun_out/r05_parity_int8.jsonl
cat gpurun_out/r05_t_early2.txt; cut -c1-1800 gpurun_out/r05_t_int8.txt

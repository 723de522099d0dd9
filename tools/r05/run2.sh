# round 5, GPU call 2: LA tail v2 + DN defaults; per-kernel profiles at batch 8 / 32; weights A/B; long parity tests
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
( timeout 900 python -m pytest tests/test_gpu_attn_la.py -x -q 2>&1 | tail -6 ) > gpurun_out/r05_t_la2.txt
( timeout 900 python -m pytest tests/test_gpu_w4.py -x -q -k "deferred or fused_qkv_rotary" 2>&1 | tail -12 ) > gpurun_out/r05_t_dn2.txt
rm -f gpurun_out/r05_ab2.jsonl
OUT=gpurun_out/r05_ab2.jsonl BATCHES=1,2,4,8,16,32 timeout 900 python tools/ab_step.py base "ZL_ATTN_LA=0,ZL_DEFER_NORM=0" "ZL_ATTN_LA=0" "ZL_DEFER_NORM=0" "ZL_ATTN_LA=1" base 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab2.txt
ZL_BENCH_WEIGHTS=random OUT=gpurun_out/r05_ab2.jsonl BATCHES=1 timeout 300 python tools/ab_step.py "base" 2>&1 | grep -v amdgpu.ids | sed 's/base/base(random weights)/' >> gpurun_out/r05_ab2.txt
cat gpurun_out/r05_ab2.txt
for b in 8 32; do
  rm -rf gpurun_out/r05_prof_b$b; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_prof_b$b -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-ttft --no-extras --batch $b --steps 16 --warmup 2 > gpurun_out/r05_prof_b$b.log 2>&1
  cp $(find gpurun_out/r05_prof_b$b -name 'bench_kernel_stats.csv' | head -1) gpurun_out/r05_bench_b${b}_kernel_stats.csv
  rm -rf gpurun_out/r05_prof_b$b
  head -14 gpurun_out/r05_bench_b${b}_kernel_stats.csv
done
# long parity tests (CPU-oracle bound) side by side
( timeout 1500 python -m pytest tests/test_gpu_fullgeom.py -x -q -k "stack_of_eight or sixteen_full or thirty_two" 2>&1 | tail -15 ) > gpurun_out/r05_t_fullgeom.txt &
( for i in 1 2 3 4 5 6 7 8 9 10; do timeout 600 python -m pytest tests/test_gpu_refcompile.py -x -q 2>&1 | tail -1; done ) > gpurun_out/r05_refcompile_x10.txt 2>&1 &
( timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -5 ) > gpurun_out/r05_t_model.txt
wait
tail -4 gpurun_out/r05_t_la2.txt gpurun_out/r05_t_dn2.txt; cat gpurun_out/r05_t_fullgeom.txt | tail -8; cat gpurun_out/r05_refcompile_x10.txt; cat gpurun_out/r05_t_model.txt

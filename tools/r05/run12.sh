set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
( timeout 900 python -m pytest tests/test_gpu_attn_la.py tests/test_gpu_f4.py tests/test_gpu_comm.py tests/test_gpu_ops.py -x -q -k "last_arriver or dispatch_index or int8 or decode_attention or mla" 2>&1 | tail -4 ) > gpurun_out/r05_t_misc.txt; cat gpurun_out/r05_t_misc.txt
rm -rf gpurun_out/r05_prof_b1; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_prof_b1 -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-ttft --no-extras --steps 32 --warmup 2 > gpurun_out/r05_prof_b1.log 2>&1
cp $(find gpurun_out/r05_prof_b1 -name 'bench_kernel_stats.csv' | head -1) gpurun_out/r05_decode_kernel_stats.csv; rm -rf gpurun_out/r05_prof_b1
grep -E "k_w4a16_i8p|k_decode_attn|k_dense_gemv|k_embedding|k_greedy" gpurun_out/r05_decode_kernel_stats.csv | cut -c1-200
tail -c 600 gpurun_out/r05_prof_b1.log | head -c 400
BATCHES=1 REPS=60 timeout 300 python tools/ab_step.py base base 2>&1 | grep -v amdgpu.ids

# round 5, GPU call 5: prefill attention wave groups; INT8 depth test v2
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "prefill" 2>&1 | tail -6 ) > gpurun_out/r05_t_prefill.txt
cat gpurun_out/r05_t_prefill.txt
for g in 1 2 4 0; do echo "ZL_PREFILL_GROUPS=$g: $(ZL_PREFILL_GROUPS=$g timeout 200 python tools/bench_prefill.py --seq 1024 2>&1 | grep -v amdgpu | tail -1)"; done > gpurun_out/r05_prefill_groups.txt
for g in 1 4; do echo "ZL_PREFILL_GROUPS=$g seq 4096: $(ZL_PREFILL_GROUPS=$g timeout 200 python tools/bench_prefill.py --seq 4096 2>&1 | grep -v amdgpu | tail -1)"; done >> gpurun_out/r05_prefill_groups.txt
cat gpurun_out/r05_prefill_groups.txt
for g in 1 4; do
  rm -rf gpurun_out/r05_pf_prof; ZL_PREFILL_GROUPS=$g timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_pf_prof -o pf --output-format csv -- python tools/bench_prefill.py --seq 1024 > /dev/null 2>&1
  grep -E "k_prefill_attn|k_w4a16_gemm_wide" $(find gpurun_out/r05_pf_prof -name 'pf_kernel_stats.csv' | head -1) | cut -c1-260 > gpurun_out/r05_prefill_kernels_g$g.txt; cat gpurun_out/r05_prefill_kernels_g$g.txt
done
rm -rf gpurun_out/r05_pf_prof
rm -f gpurun_out/parity_fullgeom.jsonl
( timeout 900 python -m pytest tests/test_gpu_fullgeom.py -x -q -s -k "int8_depth_record" 2>&1 | tail -12 ) > gpurun_out/r05_t_int8depth2.txt
cp gpurun_out/parity_fullgeom.jsonl gpurun_out/r05_parity_int8_v2.jsonl
cut -c1-1200 gpurun_out/r05_t_int8depth2.txt

# round 5, GPU call 6: prefill attention work-item pairing; INT8 depth test v3
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "prefill" 2>&1 | tail -6 ) > gpurun_out/r05_t_prefill.txt
cat gpurun_out/r05_t_prefill.txt
for g in 0 2; do echo "ZL_PREFILL_GROUPS=$g: $(ZL_PREFILL_GROUPS=$g timeout 200 python tools/bench_prefill.py --seq 1024 2>&1 | grep -v amdgpu | tail -1)"; done > gpurun_out/r05_prefill_groups2.txt
for s in 512 2048 4096; do echo "seq $s: $(timeout 200 python tools/bench_prefill.py --seq $s 2>&1 | grep -v amdgpu | tail -1)"; done >> gpurun_out/r05_prefill_groups2.txt
cat gpurun_out/r05_prefill_groups2.txt
rm -rf gpurun_out/r05_pf_prof; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_pf_prof -o pf --output-format csv -- python tools/bench_prefill.py --seq 1024 > /dev/null 2>&1
cp $(find gpurun_out/r05_pf_prof -name 'pf_kernel_stats.csv' | head -1) gpurun_out/r05_prefill_kernel_stats.csv; rm -rf gpurun_out/r05_pf_prof
grep -E "k_prefill_attn|k_w4a16_gemm_wide|k_rmsnorm|rope" gpurun_out/r05_prefill_kernel_stats.csv | cut -c1-230
rm -f gpurun_out/parity_fullgeom.jsonl
( timeout 900 python -m pytest tests/test_gpu_fullgeom.py -x -q -s -k "int8_depth_record" 2>&1 | tail -12 ) > gpurun_out/r05_t_int8depth3.txt
cp gpurun_out/parity_fullgeom.jsonl gpurun_out/r05_parity_int8_v3.jsonl
cut -c1-1500 gpurun_out/r05_t_int8depth3.txt

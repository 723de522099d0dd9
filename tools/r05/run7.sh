# round 5, GPU call 7: prefill attention map A/B on one box; the default bench line
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
for g in -1 0 -1 0; do
  rm -rf gpurun_out/r05_pf_prof; ZL_PREFILL_GROUPS=$g timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_pf_prof -o pf --output-format csv -- python tools/bench_prefill.py --seq 1024 > gpurun_out/r05_pf.log 2>&1
  echo "ZL_PREFILL_GROUPS=$g: $(grep prefill gpurun_out/r05_pf.log | tail -1) | $(grep -E 'k_prefill_attn' $(find gpurun_out/r05_pf_prof -name 'pf_kernel_stats.csv' | head -1) | cut -d, -f1-4 | cut -c40-200)"
done > gpurun_out/r05_prefill_map_ab.txt 2>&1
rm -rf gpurun_out/r05_pf_prof
cat gpurun_out/r05_prefill_map_ab.txt
timeout 900 python bench.py > gpurun_out/r05_bench2.json 2> gpurun_out/r05_bench2.err; tail -c 4000 gpurun_out/r05_bench2.json; tail -3 gpurun_out/r05_bench2.err

# round 5, GPU call 11: the round's evidence (tools/prof_round.sh r05) + LA tests at the final policy
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_attn_la.py -x -q 2>&1 | tail -3 ) > gpurun_out/r05_t_la4.txt; cat gpurun_out/r05_t_la4.txt
PROF_PREFILL=1 bash tools/prof_round.sh r05 > gpurun_out/r05_prof_round.log 2>&1
tail -5 gpurun_out/r05_prof_round.log
for b in 8 32; do
  rm -rf gpurun_out/r05_prof_b$b; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_prof_b$b -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-ttft --no-extras --batch $b --steps 16 --warmup 2 > gpurun_out/r05_prof_b$b.log 2>&1
  cp $(find gpurun_out/r05_prof_b$b -name 'bench_kernel_stats.csv' | head -1) gpurun_out/r05_bench_b${b}_kernel_stats.csv
  rm -rf gpurun_out/r05_prof_b$b
done
rm -rf gpurun_out/r05_prof gpurun_out/r05_prof_b1 gpurun_out/r05_pmc_* gpurun_out/r05_tcc_*
ls gpurun_out | grep r05_ | head -40

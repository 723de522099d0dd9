mkdir -p gpurun_out/r06
export TMPDIR=/tmp
bash tools/prof_round.sh r06 > gpurun_out/r06/prof_round.log 2>&1
tail -c 3000 gpurun_out/r06_bench.json
for b in 8 32; do
rm -rf gpurun_out/r06/ss_prof_b$b; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r06/ss_prof_b$b -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-ttft --no-extras --batch $b --steps 16 --warmup 2 > /dev/null 2>&1
cp $(find gpurun_out/r06/ss_prof_b$b -name 'bench_kernel_stats.csv' | head -1) gpurun_out/r06_bench_b${b}_kernel_stats.csv; rm -rf gpurun_out/r06/ss_prof_b$b
done
ls -la gpurun_out/ | grep r06_

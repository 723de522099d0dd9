# row-statistics hand-off: op tests, model tests, batch 8 / 16 / 32 with and without
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
make -C oracle -s
timeout 1500 python -m pytest tests/test_gpu_w4.py -q -x -k "row_ss or row_statistics or slab" 2>&1 | tail -15 > gpurun_out/r06/ss_op_tests.txt
tail -15 gpurun_out/r06/ss_op_tests.txt
bench() { timeout 600 python bench.py --batch $1 --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'], d.get('logit_check'))"; }
{
for b in 5 8 16 32; do
  ZL_ROW_SS=0 bench $b "ss=off"
  bench $b "ss=on "
done
} > gpurun_out/r06/ss_bench.txt 2>&1
cat gpurun_out/r06/ss_bench.txt
for b in 8 32; do
rm -rf gpurun_out/r06/ss_prof_b$b; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r06/ss_prof_b$b -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-ttft --no-extras --batch $b --steps 16 --warmup 2 > /dev/null 2>&1
cp $(find gpurun_out/r06/ss_prof_b$b -name 'bench_kernel_stats.csv' | head -1) gpurun_out/r06/ss_bench_b${b}_kernel_stats.csv; rm -rf gpurun_out/r06/ss_prof_b$b
head -12 gpurun_out/r06/ss_bench_b${b}_kernel_stats.csv | cut -c1-160
done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullgeom.py -q 2>&1 | tail -25 > gpurun_out/r06/ss_model_tests.txt
tail -25 gpurun_out/r06/ss_model_tests.txt

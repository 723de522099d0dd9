mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r06/full_gpu_tests.log
tail -30 gpurun_out/r06/full_gpu_tests.log
for b in 1 8 32; do
  timeout 600 python bench.py --batch $b --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', $b, d['value'], d['ms_per_step'])"
done > gpurun_out/r06/bench_batches.txt 2>&1
cat gpurun_out/r06/bench_batches.txt

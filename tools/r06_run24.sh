export TMPDIR=/tmp
make -C oracle -s
bench() { timeout 600 python bench.py --batch $1 --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 5 8; do
  ZL_GEMV_ROWS=4 bench $b "gemv rows 4"
  bench $b "gemv rows 8"
done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -x 2>&1 | tail -4

export TMPDIR=/tmp
make -C oracle -s
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "packed_dense" 2>&1 | tail -3
python tools/ubench/bench_lm_head.py 2>&1 | grep rows
bench() { timeout 600 python bench.py --batch $1 --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 8 32; do
  ZL_LM_HEAD_PACKED=0 bench $b "row-major"
  bench $b "ZLD16M   "
done

export TMPDIR=/tmp
make -C oracle -s
bench() { timeout 600 python bench.py --batch $1 --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 3 4; do bench $b "slab from 3 rows (default)"; done
timeout 2000 python -m pytest tests/test_gpu_w4.py tests/test_gpu_model.py tests/test_gpu_fullgeom.py tests/test_gpu_hostcpp.py -q -x 2>&1 | tail -4

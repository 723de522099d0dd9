export TMPDIR=/tmp
make -C oracle -s
bench() { timeout 600 python bench.py --batch $1 --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 9 16 32; do
  bench $b "default    "
  ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libstatsfirst.so bench $b "stats first"
done
ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libstatsfirst.so timeout 600 python -m pytest tests/test_gpu_w4.py -q -k "row_statistics" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_w4.py -q -k "row_statistics" 2>&1 | tail -3

export TMPDIR=/tmp
make -C oracle -s
ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libringfirst.so timeout 900 python -m pytest tests/test_gpu_w4.py -q -x -k "slab or row_statistics or fused_qkv" 2>&1 | tail -3
bench() { timeout 600 python bench.py --batch $1 --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 8 16 32; do
  bench $b "x first   "
  ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libringfirst.so bench $b "ring first"
done

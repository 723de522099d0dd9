export TMPDIR=/tmp
make -C oracle -s
timeout 900 python -m pytest tests/test_gpu_kvquant.py -q -x 2>&1 | tail -2
bench() { timeout 600 python bench.py --batch $1 --steps 40 --warmup 5 --no-ttft --no-cpu-baseline --no-extras --kv-cache-dtype int8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 8 32; do
  ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libq8pf1.so bench $b "int8 KV, one chunk ahead "
  bench $b "int8 KV, two chunks ahead"
done

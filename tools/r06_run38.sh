export TMPDIR=/tmp
make -C oracle -s
ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libattnpf3.so timeout 900 python -m pytest tests/test_gpu_attn_la.py -q -x 2>&1 | tail -2
bench() { timeout 600 python bench.py --batch $1 --steps 40 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 4 8 16 32; do
  bench $b "PF 2"
  ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libattnpf3.so bench $b "PF 3"
done

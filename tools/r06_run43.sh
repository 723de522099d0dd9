export TMPDIR=/tmp
bench() { timeout 600 python bench.py --batch $1 --steps 40 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 8 16; do
  bench $b "default            "
  ZL_ATTN_LA_WAVES=8 bench $b "8 waves            "
  ZL_ATTN_LA_WAVES=8 ZL_ATTN_LA_SPLIT=576 bench $b "8 waves, 576 keys  "
  ZL_ATTN_LA_WAVES=2 bench $b "2 waves            "
done

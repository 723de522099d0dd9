export TMPDIR=/tmp
bench() { timeout 600 python bench.py --batch $1 --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 2 3 4; do
  bench $b "default          "
  ZL_ATTN_MERGE_MAX_B=4 bench $b "merge in attn_out"
  ZL_W4_SLAB_MIN_M=3 bench $b "slab from 3 rows "
  ZL_ATTN_MERGE_MAX_B=4 ZL_W4_SLAB_MIN_M=3 bench $b "both             "
done

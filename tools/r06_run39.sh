export TMPDIR=/tmp
bench() { timeout 600 python bench.py --batch $1 --steps 40 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
for b in 2 4 8 16; do
  bench $b "split default"
  for sp in 128 192 256 384 576 1088; do
    ZL_ATTN_LA_SPLIT=$sp bench $b "split $sp"
  done
done

mkdir -p gpurun_out/r06
export TMPDIR=/tmp
make -C oracle -s
bench() { timeout 600 python bench.py --batch $1 --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
{
for b in 7 8 9 16 32; do
  ZL_ROW_SS=0 bench $b "ss=off"
  bench $b "ss=on "
done
} > gpurun_out/r06/ss4_bench.txt 2>&1
cat gpurun_out/r06/ss4_bench.txt
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r06/full_suite2.txt
tail -15 gpurun_out/r06/full_suite2.txt

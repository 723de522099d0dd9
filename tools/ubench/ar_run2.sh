# N-process one-shot all-reduce on device 0 (tests/_ar_worker.py), repeated; usage: bash tools/ubench/ar_run2.sh [world] [repeats]
w=${1:-2}; n=${2:-3}
mkdir -p gpurun_out/r2
for i in $(seq 1 $n); do
  d=$(mktemp -d)
  for r in $(seq 0 $((w-1))); do python tests/_ar_worker.py $r $w $d 0 > gpurun_out/r2/w$r.log 2>&1 & done
  wait
  echo "run $i: $(grep -h 'RESULT\|FIRST' gpurun_out/r2/w*.log | tr '\n' ';')"
done

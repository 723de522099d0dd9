d=$(mktemp -d)
for r in 0 1 2 3; do python tests/_ar_worker.py $r 4 $d 0 > gpurun_out/r2/w$r.log 2>&1 & done
wait
grep -h "RESULT\|FIRST" gpurun_out/r2/w*.log

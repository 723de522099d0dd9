cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "int8 or quant" 2>&1 | tail -3
for b in 1 8 32; do timeout 300 python bench.py --no-cpu-baseline --no-ttft --quant int8 --batch $b 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"; done

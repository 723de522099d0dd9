cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "w8a8 or int8" 2>&1 | tail -3
for b in 1 8 32; do timeout 300 python bench.py --no-cpu-baseline --no-ttft --quant int8 --batch $b 2>&1 | tail -1 | cut -c1-190; done
ZL_W8_PHASE=0 timeout 300 python bench.py --no-cpu-baseline --no-ttft --quant int8 --batch 1 2>&1 | tail -1 | cut -c1-190

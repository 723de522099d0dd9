cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -3
for b in 1 8 32; do timeout 100 python tools/bench_attn.py --batch $b --unfused 2>&1 | grep batch=; done

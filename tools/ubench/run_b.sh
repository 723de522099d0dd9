cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_w4.py tests/test_gpu_model.py -x -q -m gpu -k "fused_qkv or model or decode or prefill or step" 2>&1 | tail -3
for b in 1 8 32; do timeout 300 python bench.py --no-cpu-baseline --no-ttft --batch $b 2>&1 | tail -1 | cut -c1-200; done
ZL_FUSE_QKV_ROPE=0 timeout 300 python bench.py --no-cpu-baseline --no-ttft --batch 1 2>&1 | tail -1 | cut -c1-200

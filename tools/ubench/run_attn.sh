cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -5
for b in 1 8 32; do
echo "== batch $b fused VALU"; timeout 100 python tools/bench_attn.py --batch $b 2>&1 | grep batch=
echo "unfused mfma:"; timeout 100 python tools/bench_attn.py --batch $b --unfused 2>&1 | grep batch=
echo "unfused mfma alias:"; timeout 100 python tools/bench_attn.py --batch $b --unfused --alias 2>&1 | grep batch=
echo "unfused valu:"; ZL_ATTN_MFMA=0 timeout 100 python tools/bench_attn.py --batch $b --unfused 2>&1 | grep batch=
done
echo "== batch 8 seq 8192"; timeout 100 python tools/bench_attn.py --batch 8 --seq 8192 2>&1 | grep batch=; timeout 100 python tools/bench_attn.py --batch 8 --seq 8192 --unfused 2>&1 | grep batch=

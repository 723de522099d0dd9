cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_w4.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
for m in 1 2 4; do timeout 100 python tools/bench_gemv.py --mfma --m $m 2>&1 | grep -E "plain|norm|layer"; done
echo "== old (ZL_W4_PHASE_SMALL=0)"
ZL_W4_PHASE_SMALL=0 timeout 100 python tools/bench_gemv.py --mfma --m 1 2>&1 | grep -E "plain|norm|layer"
timeout 300 python bench.py --no-cpu-baseline --no-ttft 2>&1 | tail -1 | cut -c1-200

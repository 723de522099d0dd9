cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_w4.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
for m in 8 16; do timeout 100 python tools/bench_gemv.py --mfma --m $m 2>&1 | grep -E "plain|norm|layer"; done
for b in 8 16; do timeout 300 python bench.py --no-cpu-baseline --no-ttft --batch $b 2>&1 | tail -1 | cut -c1-200; done

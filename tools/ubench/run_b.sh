cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_w4.py -x -q -m gpu -k "tiled" 2>&1 | tail -2
for m in 64 256 1024 4096; do timeout 200 python tools/bench_gemv.py --mfma --m $m --iters 10 --layers 4 2>&1 | grep -E "plain|layer"; done

cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_w4.py -x -q -m gpu -k "phase or mfma or tiled" 2>&1 | tail -15
for m in 8 16 32; do echo "== m=$m"; timeout 100 python tools/bench_gemv.py --mfma --m $m 2>&1 | grep -v amdgpu.ids; done

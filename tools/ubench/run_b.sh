cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_w4.py -x -q -m gpu 2>&1 | tail -3
for ks in 0 2 4; do echo "== ksplit $ks"; ZL_W4_PHASE_KSPLIT=$ks timeout 100 python tools/bench_gemv.py --mfma --m 32 2>&1 | grep -E "down|layer"; ZL_W4_PHASE_KSPLIT=$ks timeout 100 python tools/bench_gemv.py --mfma --m 24 2>&1 | grep -E "down"; done

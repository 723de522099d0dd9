cd $GRAFT_REPO_ROOT
echo "== phase at m=1,4 (ZL_W4_PHASE_MIN_M=1)"
for m in 1 4; do ZL_W4_PHASE_MIN_M=1 timeout 100 python tools/bench_gemv.py --mfma --m $m 2>&1 | grep -E "plain|layer"; done
echo "== m=32 down via phase (MAXK32 big)"
ZL_W4_PHASE_MAXK32=20000 timeout 100 python tools/bench_gemv.py --mfma --m 32 2>&1 | grep -E "plain|layer"
echo "== m=32 rounds sweep (qkv N=6144: 384 tiles)"
for r in 1 2 3; do echo "rounds=$r"; ZL_W4_PHASE_ROUNDS=$r timeout 100 python tools/bench_gemv.py --mfma --m 32 2>&1 | grep -E "qkv|^o "; done
echo "== m=8 rounds sweep"
for r in 1 2; do echo "rounds=$r"; ZL_W4_PHASE_ROUNDS=$r timeout 100 python tools/bench_gemv.py --mfma --m 8 2>&1 | grep -E "qkv +plain|^o +plain|down"; done
echo "== m=24 / m=20"
timeout 100 python tools/bench_gemv.py --mfma --m 24 2>&1 | grep -E "plain|layer"

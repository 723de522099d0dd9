python -m pytest tests/test_gpu_w4.py -x -q 2>&1 | tail -2
for xf in 0 1; do echo "== XFIRST=$xf"; ZL_W4_PHASE_XFIRST=$xf python tools/bench_gemv.py --mfma --m 1 --layers 8 2>&1 | grep -v amdgpu.ids | head -8; done
for xf in 0 1; do echo "== bench XFIRST=$xf"; ZL_W4_PHASE_XFIRST=$xf python bench.py --steps 50 --no-ttft --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])"; done

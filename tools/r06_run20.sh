mkdir -p gpurun_out/r06
export TMPDIR=/tmp
make -C oracle -s
python tools/debug_slab_norm.py 2>&1 | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/test_gpu_w4.py -q -x -k "row_ss or row_statistics or slab or fused_qkv" 2>&1 | tail -5
bench() { timeout 600 python bench.py --batch $1 --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 batch', $1, d['value'], d['ms_per_step'])"; }
{
for b in 6 8 12 16 24 32; do
  ZL_ROW_SS=0 bench $b "ss=off"
  bench $b "ss=on "
done
} > gpurun_out/r06/ss3_bench.txt 2>&1
cat gpurun_out/r06/ss3_bench.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullgeom.py -q 2>&1 | tail -8

mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_refcompile.py -x -q -m gpu -k "dual_stream or tensor_parallel_engine or legacy" 2>&1 | tail -15 > gpurun_out/r06/dual_stream_tests.txt
cat gpurun_out/r06/dual_stream_tests.txt
timeout 900 python -m pytest tests/test_gpu_zz_binding.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06/binding_tests.txt
cat gpurun_out/r06/binding_tests.txt
ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libklds.so timeout 1200 python -m pytest tests/test_gpu_attn_la.py tests/test_gpu_ops.py -x -q -m gpu -k "attn or attention" 2>&1 | tail -8 > gpurun_out/r06/klds_tests.txt
cat gpurun_out/r06/klds_tests.txt
for b in 1 8 32; do for v in "" klds; do
  if [ -z "$v" ]; then so=""; else so=zhilight_amd/build/variants/lib$v.so; fi
  echo "== batch $b variant ${v:-base}"
  ZHILIGHT_AMD_SO=$so timeout 600 python bench.py --batch $b --steps 30 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done > gpurun_out/r06/klds_bench.txt 2>&1
cat gpurun_out/r06/klds_bench.txt

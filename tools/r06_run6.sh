mkdir -p gpurun_out/r06
timeout 900 python tools/debug_slab.py > gpurun_out/r06/debug_slab.txt 2>&1
echo "debug ok lines: $(grep -c ' ok ' gpurun_out/r06/debug_slab.txt)"; grep -v "amdgpu.ids\| ok " gpurun_out/r06/debug_slab.txt | head -40
timeout 2400 python -m pytest tests/test_gpu_w4.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r06/slab_tests.txt
tail -40 gpurun_out/r06/slab_tests.txt
timeout 900 python tools/bench_slab.py --m 32 16 9 > gpurun_out/r06/slab_sweep_v2.txt 2>&1
grep "^==" gpurun_out/r06/slab_sweep_v2.txt
for b in 8 32; do timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --no-ttft --no-cpu-baseline --no-extras 2>&1 | tail -2 > gpurun_out/r06/bench_b$b.txt; cat gpurun_out/r06/bench_b$b.txt | cut -c1-600; done

mkdir -p gpurun_out/r06
timeout 300 python tools/ubench/tie_rate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/tie_rate.txt; cat gpurun_out/r06/tie_rate.txt
timeout 300 tools/ubench/cluster_gather_probe > gpurun_out/r06/cluster_gather_probe.txt 2>&1; cat gpurun_out/r06/cluster_gather_probe.txt
timeout 900 python -m pytest tests/test_gpu_refcompile.py -x -q -m gpu -k "dual_stream" 2>&1 | tail -12 > gpurun_out/r06/dual_stream_tests.txt; cat gpurun_out/r06/dual_stream_tests.txt
timeout 900 python -m pytest tests/test_gpu_zz_binding.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r06/binding_tests.txt; cat gpurun_out/r06/binding_tests.txt

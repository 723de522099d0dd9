mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_refcompile.py tests/test_gpu_hostcpp.py tests/test_gpu_zz_binding.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06/boundary_fusion_tests.txt
cat gpurun_out/r06/boundary_fusion_tests.txt
for f in 1 0; do
  echo "== ZL_BOUNDARY_FUSE=$f (CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1)"
  ZL_BOUNDARY_FUSE=$f CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 600 python tools/bench_boundary.py 2>&1 | tail -1 | cut -c1-900
done > gpurun_out/r06/boundary_path.txt 2>&1
cat gpurun_out/r06/boundary_path.txt

export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
timeout 260 python -m pytest tests/test_gpu_refcompile.py tests/test_gpu_hostcpp.py tests/test_gpu_zz_binding.py -m gpu -x -q 2>&1 | tail -4
for b in 32 8; do
  ZL_BOUNDARY_FUSE=1 CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 100 python tools/bench_boundary.py --batch $b 2>&1 | tail -1 | cut -c1-900 | tee -a gpurun_out/r06b/boundary_path_batches3.txt
done

# round 6, late: the event-record shim (deferred launches go out before an event is recorded) + the boundary's step at batch 1 / 8 / 32
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_refcompile.py tests/test_gpu_zz_binding.py -m gpu -x -q 2>&1 | tail -4
for b in 1 8 32; do
  ZL_BOUNDARY_FUSE=1 CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 600 python tools/bench_boundary.py --batch $b 2>&1 | tail -1 | cut -c1-900 | tee -a gpurun_out/r06b/boundary_path_batches.txt
done

export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
timeout 300 python -m pytest tests/test_gpu_refcompile.py -m gpu -x -q -k "llama_model_decode_steps or llama_model_prompt" 2>&1 | tail -3
for b in 8 32; do
  ZL_BOUNDARY_FUSE=1 CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 300 python tools/bench_boundary.py --batch $b 2>&1 | tail -1 | cut -c1-900 | tee -a gpurun_out/r06b/boundary_path_batches2.txt
done

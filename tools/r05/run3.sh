# round 5, GPU call 3: DN parity diagnosis, attention layout probes, boundary-path timing, INT8 depth record
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
for dn in 0 1; do
  ( ZL_DEFER_NORM=$dn timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "batch_sizes_cover" 2>&1 | grep -E "AssertionError|passed|failed|assert " | head -8 ) > gpurun_out/r05_dn_model_$dn.txt
done
rm -f gpurun_out/parity_fullgeom.jsonl
( ZL_DEFER_NORM=1 timeout 600 python -m pytest tests/test_gpu_fullgeom.py -x -q -k "stack_of_eight and 32-4" 2>&1 | tail -3 ) > gpurun_out/r05_dn_fullgeom_1.txt
cp gpurun_out/parity_fullgeom.jsonl gpurun_out/r05_dn_fullgeom_1.jsonl
for b in 8 32; do for fl in "" "--bhsd" "--alias"; do echo "batch $b unfused $fl: $(timeout 120 python tools/bench_attn.py --unfused --batch $b $fl 2>&1 | grep -v amdgpu | tail -1)"; done; done > gpurun_out/r05_attn_layout.txt 2>&1
cat gpurun_out/r05_attn_layout.txt
timeout 600 python tools/bench_boundary.py > gpurun_out/r05_boundary.json 2> gpurun_out/r05_boundary.err; tail -c 1500 gpurun_out/r05_boundary.json; tail -3 gpurun_out/r05_boundary.err
CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 600 python tools/bench_boundary.py > gpurun_out/r05_boundary_fused.json 2> gpurun_out/r05_boundary_fused.err; tail -c 1500 gpurun_out/r05_boundary_fused.json; tail -3 gpurun_out/r05_boundary_fused.err
( ZL_DEFER_NORM=0 timeout 900 python -m pytest tests/test_gpu_fullgeom.py -x -q -s -k "int8_depth_record" 2>&1 | tail -12 ) > gpurun_out/r05_t_int8depth.txt
cat gpurun_out/r05_dn_model_0.txt gpurun_out/r05_dn_model_1.txt gpurun_out/r05_dn_fullgeom_1.txt; cat gpurun_out/r05_t_int8depth.txt | cut -c1-1500

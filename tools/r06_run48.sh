export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
timeout 420 python tools/logit_check_draws.py 1234 1 2 3 4 5 6 2>/dev/null | tee gpurun_out/r06b/logit_check_draws.txt
timeout 600 python -m pytest tests/test_gpu_refcompile.py tests/test_gpu_zz_binding.py -m gpu -x -q 2>&1 | tail -3
for b in 1 8 32; do
  ZL_BOUNDARY_FUSE=1 CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 300 python tools/bench_boundary.py --batch $b 2>&1 | tail -1 | cut -c1-900 | tee -a gpurun_out/r06b/boundary_path_batches.txt
done
timeout 400 python bench.py > gpurun_out/r06b/bench_final.json 2> gpurun_out/r06b/bench_final.err; echo "bench rc $?"; tail -2 gpurun_out/r06b/bench_final.err
python -c "
import json
d=json.loads(open('gpurun_out/r06b/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['seed'], [ (o['batch'], o['value']) for o in d['other_batches']], d['boundary_path'].get('boundary_path_tokens_per_s'), {k:v for k,v in d['logit_check'].items() if k!='what'})
"

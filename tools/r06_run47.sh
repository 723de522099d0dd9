export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
for i in 1 2; do
  ZL_BENCH_PARITY_SOFT=1 timeout 300 python bench.py --no-ttft --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('run $i', d['value'], json.dumps({k:v for k,v in d['logit_check'].items() if k!='what'}))"
done
python - <<'P'
import torch
print("initial seeds", torch.initial_seed(), torch.cuda.initial_seed())
print(torch.randn(4, device="cuda"))
P

set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kvquant.py tests/test_gpu_model.py -x -q -m gpu -k "kv or quant or int8_kv" 2>&1 | tail -15
for b in 1 8 32; do
timeout 120 python tools/bench_attn.py --batch $b --seq 1024
timeout 120 python tools/bench_attn.py --batch $b --seq 1024 --q8
done
timeout 120 python tools/bench_attn.py --batch 8 --seq 8192
timeout 120 python tools/bench_attn.py --batch 8 --seq 8192 --q8
timeout 300 python bench.py --no-cpu-baseline --no-ttft --kv-cache-dtype int8 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-ttft --kv-cache-dtype int8 --batch 32 2>&1 | tail -1

cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kvquant.py tests/test_gpu_model.py -x -q -m gpu -k "kv or quant" 2>&1 | tail -3
for b in 1 8 32; do timeout 100 python tools/bench_attn.py --batch $b --q8 2>&1 | grep batch=; done
timeout 100 python tools/bench_attn.py --batch 8 --seq 8192 --q8 2>&1 | grep batch=
for b in 1 32; do timeout 300 python bench.py --no-cpu-baseline --no-ttft --kv-cache-dtype int8 --batch $b 2>&1 | tail -1 | cut -c1-200; done

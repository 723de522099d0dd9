cd $GRAFT_REPO_ROOT
for b in 8 16 32 64; do timeout 100 python tools/bench_attn.py --batch $b --unfused 2>&1 | grep batch=; done
timeout 100 python tools/bench_attn.py --batch 8 --seq 8192 --unfused 2>&1 | grep batch=
for b in 8 32; do timeout 300 python bench.py --no-cpu-baseline --no-ttft --batch $b 2>&1 | tail -1 | cut -c1-200; done
timeout 300 python bench.py --no-cpu-baseline --no-ttft --batch 32 --kv-cache-dtype int8 2>&1 | tail -1 | cut -c1-200

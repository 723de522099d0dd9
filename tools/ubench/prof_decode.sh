#!/bin/bash
# per-kernel averages of the batch-1 decode step (rocprofv3 --kernel-trace --stats; no counters): usage prof_decode.sh <tag> [ENV=VAL ...]
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
rm -rf gpurun_out/r3/prof_$tag
env "$@" rocprofv3 --kernel-trace --stats -d gpurun_out/r3/prof_$tag -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-ttft --no-extras --steps 32 --warmup 2 > gpurun_out/r3/prof_$tag.log 2>&1
f=$(find gpurun_out/r3/prof_$tag -name "bench_kernel_stats.csv" | head -1)
cp $f gpurun_out/r3/decode_kernel_stats_$tag.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r3/decode_kernel_stats_$tag.csv")))
tot = 0.0
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_w4a16", "k_decode_attn", "k_dense_gemv", "k_rmsnorm", "k_rope", "k_greedy", "k_embedding")):
        short = n.split("(anonymous namespace)::")[-1].split("(")[0][:60]
        print(f"{short:62s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
tail -1 gpurun_out/r3/prof_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tok/s', d['value'], 'ms/step', d['ms_per_step'])"
rm -rf gpurun_out/r3/prof_$tag

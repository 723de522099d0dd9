# per-kernel time of a decode step at a given batch: tools/ubench/prof_batch.sh <batch> [extra bench.py flags]
export TMPDIR=/tmp
b=$1; shift
rm -rf gpurun_out/prof_b$b
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b$b -o p --output-format csv -- python bench.py --no-cpu-baseline --no-ttft --steps 16 --warmup 2 --batch $b "$@" > gpurun_out/prof_b$b.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_b$b/**/p_kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
for r in rows[:14]:
    if 'at::native' in r['Name']: continue
    print(f"{r['Name'][:110]:110s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
tail -1 gpurun_out/prof_b$b.log | cut -c1-400

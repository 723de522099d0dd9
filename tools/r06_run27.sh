export TMPDIR=/tmp
for b in 2 4; do
rm -rf /tmp/p$b; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p$b -o bench --output-format csv -- python bench.py --no-cpu-baseline --no-ttft --no-extras --batch $b --steps 16 --warmup 2 2>&1 | tail -1 | cut -c1-200
python - <<PY
import csv,glob
f=glob.glob("/tmp/p$b/**/bench_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("batch $b")
for r in rows[:24]:
    if 'at::native' in r['Name'] or 'rocclr' in r['Name'] or 'pack' in r['Name'] or 'transpose' in r['Name'] or 'shuffle' in r['Name']: continue
    print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f}")
PY
done

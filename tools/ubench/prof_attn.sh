cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_attn
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_attn -o p --output-format csv -- python tools/bench_attn.py --batch ${1:-32} --unfused --iters 5 > gpurun_out/prof_attn.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_attn/**/p_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    if 'at::native' in r['Name']: continue
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} min={float(r['MinNs'])/1e3:8.2f}")
PY

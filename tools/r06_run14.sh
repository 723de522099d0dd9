mkdir -p gpurun_out/r06
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
for f in 1 0; do
rm -rf gpurun_out/prof_boundary$f
ZL_BOUNDARY_FUSE=$f CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_boundary$f -o p --output-format csv -- python tools/bench_boundary.py --layers 8 --iters 10 > gpurun_out/r06/prof_boundary$f.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_boundary$f/**/p_kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
print("== ZL_BOUNDARY_FUSE=$f")
for r in rows[:28]:
    if 'at::native' in r['Name']: continue
    print(f"{r['Name'][:120]:120s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
done > gpurun_out/r06/prof_boundary.txt 2>&1
cat gpurun_out/r06/prof_boundary.txt

export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06b
cd /tmp
for b in 8 32; do
ZL_BOUNDARY_FUSE=1 CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06b/prof_bb$b -o p --output-format csv -- python $R/tools/bench_boundary.py --batch $b --layers 8 --iters 10 > $R/gpurun_out/r06b/prof_bb$b.log 2>&1
python - <<P
import csv, glob
f = glob.glob("$R/gpurun_out/r06b/prof_bb$b/**/p_kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
print("== boundary, batch $b (8 layers x 10 iters + warm-up + graph; rows with 1/3 of the calls belong to the Python driver's comparison step)")
for r in rows[:34]:
    if "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
    print("%-110s calls=%6s avg_us=%8.2f" % (r["Name"].replace("(anonymous namespace)::", "")[:110], r["Calls"], float(r["AverageNs"]) / 1e3))
P
rm -rf $R/gpurun_out/r06b/prof_bb$b
done | tee $R/gpurun_out/r06b/boundary_batches_kernel_stats.txt

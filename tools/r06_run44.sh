# round 6, late: the boundary's decode attention on the matrix-core kernel's mask form + the split merge deferred into attn_out
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_w4.py -m gpu -x -q -k "attention or attn or merge" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_refcompile.py tests/test_gpu_hostcpp.py tests/test_gpu_zz_binding.py -m gpu -x -q 2>&1 | tail -6
for f in 1 0; do
  ZL_BOUNDARY_FUSE=$f CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 600 python tools/bench_boundary.py 2>&1 | tail -1 | cut -c1-900 | tee -a gpurun_out/r06b/boundary_path.txt
done
cd /tmp
ZL_BOUNDARY_FUSE=1 CPM_FUSE_QKV=1 CPM_FUSE_FF_IN=1 ROPE_CACHE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06b/prof_boundary1 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_boundary.py --layers 8 --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/r06b/prof_boundary1.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/r06b/prof_boundary1/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        print("%-120s calls=%6s avg_us=%8.2f" % (r["Name"][:120], r["Calls"], float(r["AverageNs"]) / 1e3))
P

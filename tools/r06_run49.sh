# final tree: per-kernel stats of the bench command (in-step rocprofv3 --kernel-trace --stats), headline leg first
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06b
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06b/prof_b1 -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --no-ttft --no-extras --steps 32 --warmup 2 > $R/gpurun_out/r06b/prof_b1.log 2>&1
cp $(find $R/gpurun_out/r06b/prof_b1 -name 'bench_kernel_stats.csv' | head -1) $R/gpurun_out/r06b/decode_kernel_stats.csv
rm -rf $R/gpurun_out/r06b/prof_b1
timeout 230 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06b/prof_all -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 16 --warmup 2 > $R/gpurun_out/r06b/prof_all.log 2>&1
cp $(find $R/gpurun_out/r06b/prof_all -name 'bench_kernel_stats.csv' | head -1) $R/gpurun_out/r06b/bench_kernel_stats.csv
rm -rf $R/gpurun_out/r06b/prof_all
ls -la $R/gpurun_out/r06b/*.csv; tail -c 300 $R/gpurun_out/r06b/prof_b1.log

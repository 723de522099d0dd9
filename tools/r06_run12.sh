mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for b in 8 32; do bash tools/ubench/prof_batch.sh $b --no-extras > gpurun_out/r06/prof_b$b.txt 2>&1; cat gpurun_out/r06/prof_b$b.txt | cut -c1-200; cp $(find gpurun_out/prof_b$b -name 'p_kernel_stats.csv' | head -1) gpurun_out/r06/bench_b${b}_kernel_stats.csv; done

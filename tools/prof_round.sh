set -x
export TMPDIR=/tmp
make -C oracle -s
timeout 900 python bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err; tail -c 2000 gpurun_out/bench_r01.json
rm -rf gpurun_out/prof_r01; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o bench --output-format csv -- python bench.py --no-cpu-baseline --steps 16 --warmup 2 > gpurun_out/prof_r01.log 2>&1; ls gpurun_out/prof_r01
timeout 600 tools/pmc.sh gpurun_out/pmc_r01_gateup k_w4a16_mfma -- python tools/prof_one.py 28672 4096 1 6 mfma > /dev/null; cat gpurun_out/pmc_r01_gateup/summary.txt

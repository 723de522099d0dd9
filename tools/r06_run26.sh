mkdir -p gpurun_out/r06
export TMPDIR=/tmp
make -C oracle -s
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r06/full_suite3.txt
tail -8 gpurun_out/r06/full_suite3.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/r06_run22.sh > gpurun_out/r06/prof_round2.log 2>&1
tail -c 1200 gpurun_out/r06_bench.json

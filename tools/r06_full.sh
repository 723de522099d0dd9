# full GPU suite + smoke on the current tree; results under gpurun_out/r06/
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
make -C oracle -s
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r06/full_suite.txt
tail -15 gpurun_out/r06/full_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r06/smoke.txt

# round 5, GPU call 8: the whole -m gpu suite at the current defaults
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
timeout 2400 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -40 > gpurun_out/r05_full_gpu_tests.log
tail -25 gpurun_out/r05_full_gpu_tests.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 python tools/r05/diag_tp.py 2>&1 | tail -12 ) > gpurun_out/r16_diag_tp.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_refcompile.py -q -m gpu -p no:cacheprovider -k "tensor_parallel or deepseek" 2>&1 | tail -30 ) > gpurun_out/r16_tp_tests.log 2>&1
for i in 1 2 3; do ( timeout 300 python -m pytest tests/test_gpu_refcompile.py -q -m gpu -p no:cacheprovider -k "tensor_parallel" 2>&1 | tail -2 ) >> gpurun_out/r16_tp_repeat.log 2>&1; done
head -3 gpurun_out/r16_diag_tp.log; tail -12 gpurun_out/r16_tp_tests.log; cat gpurun_out/r16_tp_repeat.log

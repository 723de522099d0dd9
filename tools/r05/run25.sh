#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 80 python -m pytest tests/test_gpu_zz_binding.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6 ) > gpurun_out/r25_binding_test.log 2>&1
( ZL_BINDING_EXTRA=1 ZL_BINDING_WATCHDOG=40 timeout 50 python tests/_binding_worker.py 2>&1 | grep -a "BINDING_RESULT\|Error\|error\|Exception\|assert" | tail -6 ) > gpurun_out/r25_binding_extra.log 2>&1
cat gpurun_out/r25_binding_test.log; tail -c 3000 gpurun_out/r25_binding_extra.log

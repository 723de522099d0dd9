#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 80 python -m pytest tests/test_gpu_zz_binding.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6 ) > gpurun_out/r25_binding_test.log 2>&1

cat gpurun_out/r25_binding_test.log

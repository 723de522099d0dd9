#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 40 python -m pytest tests/test_gpu_zz_sampling.py -q -m gpu -p no:cacheprovider 2>&1 | tail -25 ) > gpurun_out/r27_sampling.log 2>&1
cat gpurun_out/r27_sampling.log

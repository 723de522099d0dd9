#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 400 python -m pytest tests/test_gpu_binding.py -q -m gpu -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/r24_binding.log 2>&1
tail -40 gpurun_out/r24_binding.log

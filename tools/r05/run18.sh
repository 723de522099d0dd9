#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 800 python -m pytest tests/test_gpu_refcompile.py -q -m gpu -p no:cacheprovider -k "deepseek or tensor_parallel or mla_layer" 2>&1 | tail -45 ) > gpurun_out/r18_ds.log 2>&1
tail -45 gpurun_out/r18_ds.log

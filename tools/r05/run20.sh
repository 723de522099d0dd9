#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_refcompile.py -q -m gpu -p no:cacheprovider -k "engine_collectives" 2>&1 | tail -30 ) > gpurun_out/r20_coll.log 2>&1
tail -30 gpurun_out/r20_coll.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp DYN_BATCH_DEBUG=1
mkdir -p gpurun_out
( timeout 200 python tools/r05/try_binding.py 2>&1 | tail -60 ) > gpurun_out/r23_binding.log 2>&1
tail -60 gpurun_out/r23_binding.log

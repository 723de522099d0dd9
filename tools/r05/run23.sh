#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp DYN_BATCH_DEBUG=1
mkdir -p gpurun_out
( timeout 200 python tests/_binding_worker.py   # (at the time: tools/r05/try_binding.py, the same code before it became the test's child process) 2>&1 | tail -60 ) > gpurun_out/r23_binding.log 2>&1
tail -60 gpurun_out/r23_binding.log

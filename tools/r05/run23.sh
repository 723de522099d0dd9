#!/bin/bash
# call 23: the first generation through the reference's own zhilight.C (at the time this ran tools/r05/try_binding.py -- the same code
# before it became tests/_binding_worker.py, the child process of tests/test_gpu_zz_binding.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp DYN_BATCH_DEBUG=1
mkdir -p gpurun_out
( timeout 200 python tests/_binding_worker.py 2>&1 | tail -60 ) > gpurun_out/r23_binding.log 2>&1
tail -60 gpurun_out/r23_binding.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python tools/r05/diag_tp.py 2>&1 | tail -40 ) > gpurun_out/r15_diag_tp.log 2>&1
cat gpurun_out/r15_diag_tp.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp ZL_DUMP_DS2=1
mkdir -p gpurun_out
( timeout 800 python -m pytest tests/test_gpu_refcompile.py -q -m gpu -p no:cacheprovider -k "deepseek_v3_shaped_layer_sharded" 2>&1 | tail -8 ) > gpurun_out/r19_ds.log 2>&1
tail -8 gpurun_out/r19_ds.log; ls -la gpurun_out/*.npz

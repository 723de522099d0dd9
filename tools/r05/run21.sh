#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp ZL_DUMP_DS2=1 ZL_DEBUG_MOE=1
mkdir -p gpurun_out
( timeout 800 python -m pytest tests/test_gpu_refcompile.py -q -m gpu -p no:cacheprovider -k "deepseek_v3_shaped_layer_sharded" 2>&1 | grep -a "sum_experts\|passed\|failed" | head -20 ) > gpurun_out/r21_ds.log 2>&1
cat gpurun_out/r21_ds.log

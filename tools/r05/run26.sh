#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 70 python -m pytest tests/test_gpu_refcompile.py tests/test_gpu_hostcpp.py -q -m gpu -x -p no:cacheprovider -k "llama_model_decode_steps or engine_collectives or tensor_parallel_engine_two or hostcpp" 2>&1 | tail -4 ) > gpurun_out/r26_sanity.log 2>&1
cat gpurun_out/r26_sanity.log

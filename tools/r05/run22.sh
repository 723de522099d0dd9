#!/bin/bash
# round 5, call 22: confirmation of the host-library work at HEAD -- the whole reference-unit suite, the shim's own tests, config 5's
# kernels (they share the shim), the engine tests twice more, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_refcompile.py -q -m gpu -p no:cacheprovider 2>&1 | tail -25 ) > gpurun_out/r22_refcompile.log 2>&1
( timeout 500 python -m pytest tests/test_gpu_hostcpp.py tests/test_gpu_f4.py -q -m gpu -p no:cacheprovider 2>&1 | tail -8 ) > gpurun_out/r22_hostcpp_f4.log 2>&1
for i in 1 2; do ( timeout 400 python -m pytest tests/test_gpu_refcompile.py -q -m gpu -p no:cacheprovider -k "tensor_parallel or sharded or engine_collectives" 2>&1 | tail -2 ) >> gpurun_out/r22_engine_repeat.log 2>&1; done
( timeout 500 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 ) > gpurun_out/r22_bench.log 2>&1
tail -6 gpurun_out/r22_refcompile.log; tail -3 gpurun_out/r22_hostcpp_f4.log; cat gpurun_out/r22_engine_repeat.log; tail -c 600 gpurun_out/r22_bench.log

#!/bin/bash
# round 5, call 14: the host library (libzhilight_amd_host.so: engine, model_context.cpp compiled unmodified, host_*.cpp) on the GPU --
# the whole reference-unit suite on the restructured module, the new world-2 engine tests, the DeepSeek-V3-shaped layer, the hostcpp
# shim's own tests, then the bench line (its boundary leg drives RefLLaMA through the same module).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests/test_gpu_refcompile.py -q -m gpu -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/r14_refcompile.log 2>&1
( timeout 400 python -m pytest tests/test_gpu_hostcpp.py tests/test_gpu_f4.py -q -m gpu -x -p no:cacheprovider -k "hostcpp or reference_fp8block or flash_mla_binding_through" 2>&1 | tail -15 ) > gpurun_out/r14_hostcpp.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -5 ) > gpurun_out/r14_bench.log 2>&1
tail -25 gpurun_out/r14_refcompile.log; tail -5 gpurun_out/r14_hostcpp.log; tail -c 1500 gpurun_out/r14_bench.log

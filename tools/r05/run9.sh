# round 5, GPU call 9: 8-wave decode attention (long splits / no split), reference-unit tests on oracle/ references
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
( timeout 900 python -m pytest tests/test_gpu_attn_la.py -x -q 2>&1 | tail -6 ) > gpurun_out/r05_t_la3.txt; tail -3 gpurun_out/r05_t_la3.txt
rm -f gpurun_out/r05_ab3.jsonl
OUT=gpurun_out/r05_ab3.jsonl BATCHES=8 timeout 600 python tools/ab_step.py base "ZL_ATTN_LA_SPLIT=128" "ZL_ATTN_LA_SPLIT=288" "ZL_ATTN_LA_SPLIT=544" "ZL_ATTN_LA_SPLIT=1088" base 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab3.txt
OUT=gpurun_out/r05_ab3.jsonl BATCHES=16,32 timeout 600 python tools/ab_step.py base "ZL_ATTN_LA_SPLIT=128" "ZL_ATTN_LA_SPLIT=384" "ZL_ATTN_LA_SPLIT=544" "ZL_ATTN_LA_SPLIT=1088" base 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_ab3.txt
cat gpurun_out/r05_ab3.txt
( timeout 900 python -m pytest tests/test_gpu_refcompile.py -x -q -s -k "attention" 2>&1 | grep -E "err|passed|failed|Error" | head -30 ) > gpurun_out/r05_t_refattn.txt; cat gpurun_out/r05_t_refattn.txt

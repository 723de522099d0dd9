# round 5, GPU call 1: correctness of the last-arriver attention + deferred norm, A/B of the decode step, bench line on the 8(d) weights
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
( timeout 900 python -m pytest tests/test_gpu_attn_la.py -x -q 2>&1 | tail -25 ) > gpurun_out/r05_t_la.txt
( timeout 900 python -m pytest tests/test_gpu_w4.py -x -q -k "deferred or fused_qkv_rotary" 2>&1 | tail -25 ) > gpurun_out/r05_t_dn.txt
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "decode_attention" 2>&1 | tail -8 ) > gpurun_out/r05_t_ops.txt
tail -5 gpurun_out/r05_t_la.txt gpurun_out/r05_t_dn.txt gpurun_out/r05_t_ops.txt
rm -f gpurun_out/r05_ab1.jsonl
OUT=gpurun_out/r05_ab1.jsonl BATCHES=1 timeout 600 python tools/ab_step.py base "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=32" "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=32,ZL_ATTN_LA_HALF=1" \
   "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=64" "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=64,ZL_ATTN_LA_HALF=1" "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=128" base 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_ab1.txt
OUT=gpurun_out/r05_ab1.jsonl BATCHES=8,32 timeout 900 python tools/ab_step.py base "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=64" "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=128" "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=256" \
   "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=384" "ZL_DEFER_NORM=1" "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=128,ZL_DEFER_NORM=1" "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=256,ZL_DEFER_NORM=1" base 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_ab1.txt
cat gpurun_out/r05_ab1.txt
ZL_BENCH_PARITY_SOFT=1 timeout 600 python bench.py --no-extras --no-ttft > gpurun_out/r05_bench1.json 2> gpurun_out/r05_bench1.err; tail -c 1800 gpurun_out/r05_bench1.json; tail -3 gpurun_out/r05_bench1.err

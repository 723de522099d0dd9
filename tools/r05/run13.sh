set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle -s
rm -f gpurun_out/qwen_tp4.txt
( ZL_QWEN_TP4_LAYERS=1 ZL_QWEN_TP4_PROMPT=2048 timeout 1200 python -m pytest tests/test_gpu_comm.py -x -q -s -k "qwen2_72b" 2>&1 | tail -12 ) > gpurun_out/r05_t_qwen1.txt; cut -c1-1500 gpurun_out/r05_t_qwen1.txt
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "argmax or tensor_parallel" 2>&1 | tail -3 ) > gpurun_out/r05_t_argmax.txt; cat gpurun_out/r05_t_argmax.txt
BATCHES=8,32 REPS=40 timeout 300 python tools/ab_step.py base 2>&1 | grep -v amdgpu.ids

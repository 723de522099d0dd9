mkdir -p gpurun_out/r06
export TMPDIR=/tmp
make -C oracle -s
timeout 900 python -m pytest tests/test_gpu_w4.py -q -k "norm_from_row_statistics" 2>&1 | tail -15
./tools/ubench/x_delivery_probe > gpurun_out/r06/x_delivery_probe.txt 2>&1; cat gpurun_out/r06/x_delivery_probe.txt

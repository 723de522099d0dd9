export TMPDIR=/tmp
timeout 232 python -m pytest tests/test_gpu_refcompile.py tests/test_gpu_hostcpp.py tests/test_gpu_zz_binding.py -m gpu -q 2>&1 | tail -12 | cut -c1-600

"""Run ONE W4A16 GEMM shape a few times (for rocprofv3 --pmc / --kernel-trace).
usage: python tools/prof_one.py N K [M] [iters] [mfma]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_gemv import rand_w4, rand_w4m  # noqa: E402
from zhilight_amd import ops  # noqa: E402

n, k = int(sys.argv[1]), int(sys.argv[2])
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
mfma = len(sys.argv) > 5 and sys.argv[5] == "mfma"
dev = torch.device("cuda:0")
ws = [(rand_w4m if mfma else rand_w4)(n, k, 128, dev) for _ in range(6)]
x = torch.randn(m, k, dtype=torch.float16, device=dev)
out = torch.empty(m, n, dtype=torch.float16, device=dev)
gemm = ops.w4a16_gemm_mfma if mfma else ops.w4a16_gemm
for i in range(iters):
    gemm(x, ws[i % 6], out=out)
torch.cuda.synchronize()

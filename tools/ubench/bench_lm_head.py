"""us per launch of the decode batches' lm_head (zl_gemm_nt, 128256 x 4096 fp16) at 5..32 rows; weights rotate over 3 copies (HBM-cold).
usage: python tools/ubench/bench_lm_head.py   (ZHILIGHT_AMD_SO=... for a variant build)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zhilight_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
n, k = 128256, 4096
ws = [(torch.randn(n, k, dtype=torch.float16, device=dev) * 0.02) for _ in range(3)]
for m in (5, 8, 16, 32):
    x = torch.randn(m, k, dtype=torch.float16, device=dev)
    out = torch.empty(m, n, dtype=torch.float16, device=dev)
    for w in ws:
        ops.gemm_nt(x, w, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(30):
        ops.gemm_nt(x, ws[i % 3], out=out)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 30
    print(f"rows {m:2d}: {us:7.1f} us per launch  {n * k * 2 / us / 1e6:5.2f} TB/s")
wps = [ops.DenseMWeight(w) for w in ws]
for m in (5, 8, 16, 32):
    x = torch.randn(m, k, dtype=torch.float16, device=dev)
    out = torch.empty(m, n, dtype=torch.float16, device=dev)
    for w in wps:
        ops.gemm_nt_packed(x, w, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(30):
        ops.gemm_nt_packed(x, wps[i % 3], out=out)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 30
    print(f"rows {m:2d} (ZLD16M): {us:7.1f} us per launch  {n * k * 2 / us / 1e6:5.2f} TB/s")

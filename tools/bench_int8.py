"""int8 GEMM (Int8Linear hot kernel) on the Llama-3-8B linear shapes: python tools/bench_int8.py [--m 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=32)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
tot = 0.0
for name, n, k in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate", 14336, 4096), ("up", 14336, 4096), ("down", 4096, 14336)]:
    ws = [torch.randint(-127, 128, (n, k), dtype=torch.int8, device=dev) for _ in range(4)]
    x = torch.randint(-127, 128, (a.m, k), dtype=torch.int8, device=dev)
    out = torch.empty(a.m, n, dtype=torch.int32, device=dev)
    ops.int8_gemm_nt(x, ws[0], out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(a.iters):
            ops.int8_gemm_nt(x, ws[i % 4], out=out)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    tot += us
    print(f"{name:5s} M={a.m} N={n} K={k}: {us:8.2f} us  {n * k / us / 1e3:8.1f} GB/s ({n * k / us / 1e3 / 80:5.1f}% of 8 TB/s)")
print(f"layer total: {tot:.1f} us")

"""MLA decode attention over the latent cache (zl_mla_decode_attn_ex): time per launch pair (partial + combine), hipGraph, HIP events,
for the matrix-core kernel (algo 0, the default) and the VALU kernel (algo 1).
usage: python tools/bench_mla.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for b, h, L in ((1, 128, 1024), (1, 16, 1024), (8, 128, 1024), (32, 128, 1024), (1, 128, 8192)):
    q = torch.randn(b, h, 576, device=dev).to(torch.bfloat16)
    bufs = [torch.randn(L, 576, device=dev).to(torch.bfloat16) for _ in range(b)]
    addrs = torch.tensor([t.data_ptr() for t in bufs], dtype=torch.int64, device=dev)
    lens = torch.full((b,), L, dtype=torch.int32, device=dev)
    res = []
    for algo in (0, 1):
        ops.mla_decode_attention(q, lens, addrs, 0.1, L, algo=algo)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                ops.mla_decode_attention(q, lens, addrs, 0.1, L, algo=algo)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 20)
    flops = 2.0 * b * h * L * (576 + 512)
    print(f"batch {b:3d} heads {h:4d} keys {L:5d}: matrix cores {res[0]:8.1f} us ({flops / res[0] / 1e6:7.2f} TFLOP/s)   VALU {res[1]:8.1f} us"
          f"   cache {b * L * 1152 / 1e6:6.1f} MB")

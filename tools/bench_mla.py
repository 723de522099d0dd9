"""MLA decode attention over the latent cache (zl_mla_decode_attn_ex): time per launch pair (partial + combine), hipGraph, HIP events,
for the 16-heads-per-workgroup matrix-core kernel (algo 3), the wide kernel (algo 2: all 128 heads per workgroup, what algo 0 picks
from 8 tasks on) and the VALU kernel (algo 1).  The 20 launches of a graph rotate over enough cache sets to exceed the 256 MB
Infinity Cache (a decode step reads 61 layers' caches once each).
usage: python tools/bench_mla.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for b, h, L in ((1, 128, 1024), (1, 16, 1024), (8, 128, 1024), (32, 128, 1024), (32, 128, 4096), (1, 128, 8192)):
    q = torch.randn(b, h, 576, device=dev).to(torch.bfloat16)
    nset = max(1, min(20, int(300e6 // (b * L * 1152)) + 1))
    sets = []
    for _ in range(nset):
        bufs = [torch.randn(L, 576, device=dev).to(torch.bfloat16) for _ in range(b)]
        sets.append((bufs, torch.tensor([t.data_ptr() for t in bufs], dtype=torch.int64, device=dev)))
    lens = torch.full((b,), L, dtype=torch.int32, device=dev)
    res = {}
    for algo in (3, 2, 1):
        if algo == 2 and h != 128:
            res[algo] = float("nan")
            continue
        ops.mla_decode_attention(q, lens, sets[0][1], 0.1, L, algo=algo)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(20):
                ops.mla_decode_attention(q, lens, sets[i % nset][1], 0.1, L, algo=algo)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res[algo] = e0.elapsed_time(e1) * 1e3 / 20
    flops = 2.0 * b * h * L * (576 + 512)
    mb = b * L * 1152 / 1e6
    print(f"batch {b:3d} heads {h:4d} keys {L:5d}: 16 heads / workgroup {res[3]:8.1f} us   wide {res[2]:8.1f} us ({mb / res[2]:5.2f} TB/s of cache bytes, "
          f"{flops / res[2] / 1e6:7.2f} TFLOP/s)   VALU {res[1]:8.1f} us   cache {mb:6.1f} MB x {nset} sets")
    del sets

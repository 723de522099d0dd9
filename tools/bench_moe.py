"""Fused MoE GEMVs (zl_w4a16_moe_up / _down) at a decode step's shapes: per-launch time and algorithmic HBM rate.
usage: python tools/bench_moe.py [--m 1] [--hidden 2048] [--ff 768] [--experts 128] [--topk 8] [--shared 0] [--layers 24]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=1)
ap.add_argument("--hidden", type=int, default=2048)
ap.add_argument("--ff", type=int, default=768)
ap.add_argument("--experts", type=int, default=128)
ap.add_argument("--topk", type=int, default=8)
ap.add_argument("--shared", type=int, default=0)
ap.add_argument("--layers", type=int, default=24)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = 128


def stack(n, k, interleave):
    L = ops.W4Weight.layout(n, k, g)
    e = a.experts + a.shared
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (e, L.qw_bytes // 4), dtype=torch.int32, device=dev)
    sc = (torch.rand(e, L.scales_bytes // 2, device=dev) * 0.005 + 1e-4).half()
    zs = torch.randint(-2 ** 15, 2 ** 15 - 1, (e, L.zeros_bytes // 2), dtype=torch.int16, device=dev)
    return ops.W4MoEWeight(e, n, k, g, qw, sc, zs, interleave)


ups = [stack(2 * a.ff, a.hidden, True) for _ in range(a.layers)]
downs = [stack(a.hidden, a.ff, False) for _ in range(a.layers)]
t = a.topk + a.shared
ids = torch.stack([torch.randperm(a.experts, device=dev)[:a.topk] for _ in range(a.m)]).to(torch.int32)
wts = torch.rand(a.m, a.topk, device=dev)
x = torch.randn(a.m, a.hidden, device=dev).half()
mid = torch.empty(a.m, t, a.ff, dtype=torch.float16, device=dev)
out = torch.empty(a.m, a.hidden, dtype=torch.float16, device=dev)
for name, fn, byt in (("moe_up", lambda i: ops.moe_up(x, ups[i], ids, a.shared, out=mid), a.m * t * 2 * a.ff * a.hidden * (0.5 + 2.5 / g)),
                      ("moe_down", lambda i: ops.moe_down(mid, downs[i], ids, wts, a.shared, out=out), a.m * t * a.hidden * a.ff * (0.5 + 2.5 / g))):
    fn(0)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for i in range(a.layers):
                fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (a.iters * a.layers)
    print(f"{name:9s} M={a.m} experts/token={t} ({a.hidden} x {a.ff}): {us:8.2f} us/launch  {byt / us / 1e3:8.1f} GB/s ({byt / us / 1e3 / 80:.1f}% of 8 TB/s)  {byt / 1e6:.1f} MB")

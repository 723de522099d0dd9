"""Micro-benchmark of the W4A16 decode GEMM on the Llama-3-8B shapes: per-launch time
with HIP events, weights rotated through distinct buffers (> 256 MB Infinity Cache) so they come
from HBM.  Usage: python tools/bench_gemv.py [--m 1] [--iters 50] [--layers 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402


def rand_w4(n, k, g, dev, interleave=False):
    return ops.W4Weight.random(n, k, g, dev, row_interleave=interleave)


def rand_w4m(n, k, g, dev, interleave=False):
    return ops.W4MWeight.random(n, k, g, dev, row_interleave=interleave)


def alg_bytes(n, k, g, m):
    return k * n * (0.5 + 2.0 / g + 0.5 / g) + m * k * 2 + m * n * 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--mfma", action="store_true", help="bench the MFMA flavour instead of the bit-exact VALU kernel")
    a = ap.parse_args()
    gemm = ops.w4a16_gemm_mfma if a.mfma else ops.w4a16_gemm
    mk = rand_w4m if a.mfma else rand_w4
    dev = torch.device("cuda:0")
    shapes = [("qkv", 6144, 4096, 0), ("o", 4096, 4096, 0), ("gate_up", 28672, 4096, ops.EPI_SILU_MUL), ("down", 4096, 14336, 0)]
    ws = {name: [mk(n, k, 128, dev, interleave=bool(epi)) for _ in range(a.layers)] for name, n, k, epi in shapes}
    nw = torch.ones(14336, dtype=torch.float16, device=dev)
    tot_t, tot_b = 0.0, 0.0
    for name, n, k, epi in shapes:
        x = torch.randn(a.m, k, dtype=torch.float16, device=dev)
        out = torch.empty(a.m, n // 2 if epi else n, dtype=torch.float16, device=dev)
        for variant in ("plain", "norm"):
            if variant == "norm" and (k > 8192 or a.m > 16):
                continue  # the model never fuses a norm into the down projection / into multi-pass GEMMs
            kw = dict(norm_weight=nw[:k], norm_eps=1e-5) if variant == "norm" else {}
            for w in ws[name]:
                gemm(x, w, out=out, epilogue=epi, **kw)
            torch.cuda.synchronize()
            # capture the launches in a hipGraph: python/ctypes call overhead (~10 us) would otherwise
            # dominate the 2-10 us kernels
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for i in range(a.iters):
                    gemm(x, ws[name][i % a.layers], out=out, epilogue=epi, **kw)
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            b = alg_bytes(n, k, 128, a.m)
            tf = 2.0 * a.m * n * k / us / 1e6
            print(f"{name:8s} {variant:5s} M={a.m} N={n} K={k}: {us:8.2f} us/launch  {b / us / 1e3:8.1f} GB/s  ({b / us / 1e3 / 8000 * 100:5.1f}% of 8 TB/s)"
                  + (f"  {tf:7.1f} TFLOP/s ({tf / 2500 * 100:4.1f}% of 2.5 PF)" if a.m >= 32 else ""))
            if variant == "plain":
                tot_t += us
                tot_b += b
    print(f"layer total (plain): {tot_t:.2f} us, {tot_b / tot_t / 1e3:.1f} GB/s ({tot_b / tot_t / 1e3 / 80:.1f}% of 8 TB/s), x32 layers = {tot_t * 32:.0f} us")
    if a.m > 64:
        return
    # lm_head
    w = [torch.randn(128256, 4096, dtype=torch.float16, device=dev) * 0.02 for _ in range(2)]
    x = torch.randn(a.m, 4096, dtype=torch.float16, device=dev)
    out = torch.empty(a.m, 128256, dtype=torch.float16, device=dev)
    for i in range(2):
        ops.gemm_nt_small_m(x, w[i], out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10):
        ops.gemm_nt_small_m(x, w[i % 2], out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    b = 128256 * 4096 * 2
    print(f"lm_head M={a.m}: {us:.1f} us  {b / us / 1e3:.1f} GB/s ({b / us / 1e3 / 80:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()

"""AWQ checkpoints, two routes, per-launch time at decode batch sizes (HIP events, hipGraph, weights rotated through 8 copies):
  native   zl_awq_gemm on the (K, N/8) tensors as stored (AWQ_USE_EXLLAMA=0: no load-time work; csrc/awq_native.hip, VALU)
  repack   the same checkpoint re-tiled ONCE at load (shuffle_awq -> k-major -> ZLW4M, what AWQ_USE_EXLLAMA=1 does in the
           reference) and run on the matrix-core GEMV kernels (k_w4a16_i8p for 1-4 rows, k_w4a16_phase above)
usage: python tools/bench_awq.py [--iters 40]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from zhilight_amd import ops  # noqa: E402


def timed(fn, iters):
    fn(0)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(iters):
            fn(i)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    g = 128
    print("shape (N x K)        M   native us   repack us   native GB/s   repack GB/s")
    for name, n, k in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate|up", 28672, 4096), ("down", 4096, 14336)):
        copies = []
        for _ in range(8):
            qw, qz, sc, _, _ = synth.awq_hf(rng, k, n, g)
            t = [torch.from_numpy(v).to(dev) for v in (qw.view(np.int32), qz.view(np.int32), sc.view(np.float16))]
            km_q = ops.shuffle_awq(t[0].clone(), True)                       # (K/8, N) exllama order
            zu = t[1].clone()
            ops.awq_un_shuffle(zu)
            w = ops.W4MWeight.from_k_major(ops.transpose_2d(km_q), ops.transpose_2d(ops.q4_to_q8(zu)), ops.transpose_2d(t[2]), g)
            copies.append((t, w))
        bytes_ = k * n * (0.5 + 2.5 / g)
        for m in (1, 4, 8):
            x = torch.randn(m, k, dtype=torch.float16, device=dev)
            out = torch.empty(m, n, dtype=torch.float16, device=dev)
            t_nat = timed(lambda i: ops.awq_gemm(x, *copies[i % 8][0], g, out=out), a.iters)
            t_rep = timed(lambda i: ops.w4a16_gemm_mfma(x, copies[i % 8][1], out=out), a.iters)
            print(f"{name:8s} {n:5d}x{k:5d} {m:3d} {t_nat:10.2f} {t_rep:10.2f} {bytes_ / t_nat / 1e3:12.0f} {bytes_ / t_rep / 1e3:12.0f}")


if __name__ == "__main__":
    main()

"""fused qkv projection + rotary + KV scatter (zl_w4a16_qkv_rope_scatter) vs the plain / fused-norm projection of the same
shape, hipGraph chain over 8 distinct weights (Llama-3-8B: N = 6144, K = 4096, batch 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zhilight_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
h, hkv, d, k, L = 32, 8, 128, 4096, 1088
ws = [ops.W4MWeight.random((h + 2 * hkv) * d, k, 128, dev) for _ in range(8)]
x = torch.randn(1, k, dtype=torch.float16, device=dev)
nw = torch.ones(k, dtype=torch.float16, device=dev)
kb = [torch.zeros(L, hkv, d, dtype=torch.float16, device=dev)]
vb = [torch.zeros(L, hkv, d, dtype=torch.float16, device=dev)]
ka, va = ops.make_ptr_table(kb), ops.make_ptr_table(vb)
pos = torch.tensor([1024], dtype=torch.int32, device=dev)
cos, sin = ops.rope_cos_sin(pos, d, 5e5, True, None)
bl = torch.tensor([L], dtype=torch.int32, device=dev)
qo = torch.empty(1, h * d, dtype=torch.float16, device=dev)
out = torch.empty(1, (h + 2 * hkv) * d, dtype=torch.float16, device=dev)


def run(fn, name, iters=48):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i % 8)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) * 1e3 / (5 * iters):6.2f} us")


run(lambda i: ops.w4a16_gemm_mfma(x, ws[i], out=out), "plain")
run(lambda i: ops.w4a16_gemm_mfma(x, ws[i], out=out, norm_weight=nw, norm_eps=1e-5), "fused norm")
run(lambda i: ops.w4_qkv_rope_scatter(x, ws[i], cos, sin, pos, bl, ka, va, h, hkv, d, q_out=qo), "rotary + scatter")
run(lambda i: ops.w4_qkv_rope_scatter(x, ws[i], cos, sin, pos, bl, ka, va, h, hkv, d, norm_weight=nw, norm_eps=1e-5, q_out=qo),
    "fused norm + rotary + scatter")

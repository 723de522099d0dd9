"""Fused decode attention (rope + KV scatter + split-KV softmax + split merge) per-launch time, hipGraph-captured:
`layers` launches over distinct KV buffers per replay, like one decode step.
usage: python tools/bench_attn.py [--batch 1] [--seq 1024] [--layers 32]
(an in-launch split merge by the last-arriving workgroup was tried against the separate combine kernel:
13.1 vs ~13.9 us stand-alone, but +0.7 us per layer inside the decode step -- not kept)"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bhsd", action="store_true", help="KV buffers (Hkv, len_buf, D) instead of (len_buf, Hkv, D)")
    ap.add_argument("--alias", action="store_true", help="experiment: every task and layer reads the SAME KV buffer (cache-resident): the kernel's compute/latency floor")
    ap.add_argument("--unfused", action="store_true", help="rope+scatter launch, then zl_decode_attn (the matrix-core kernel when it applies)")
    ap.add_argument("--q8", action="store_true", help="INT8 KV cache: rope+quantise+scatter launch, then attention over codes")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    h, hkv, d = 32, 8, 128
    len_buf = (a.seq + 64 + 63) // 64 * 64
    shape = (a.layers, 2, hkv, len_buf, d) if a.bhsd else (a.layers, 2, len_buf, hkv, d)
    kv = [torch.randn(shape, dtype=torch.float16, device=dev) for _ in range(1 if a.alias else a.batch)]
    if a.alias:
        kv = [kv[0]] * a.batch
    k_addrs = torch.tensor([[t[l, 0].data_ptr() for t in kv] for l in range(a.layers)], dtype=torch.int64, device=dev)
    v_addrs = torch.tensor([[t[l, 1].data_ptr() for t in kv] for l in range(a.layers)], dtype=torch.int64, device=dev)
    if a.alias:
        k_addrs[:] = k_addrs[0, 0]
        v_addrs[:] = v_addrs[0, 0]
    i32 = dict(dtype=torch.int32, device=dev)
    pos = torch.full((a.batch,), a.seq, **i32)
    buf_lens = torch.full((a.batch,), len_buf, **i32)
    valid = torch.full((a.batch,), a.seq + 1, **i32)
    cos, sin = ops.rope_cos_sin(pos, d, 5e5, True, None)
    qkv = torch.randn(a.batch, (h + 2 * hkv) * d, dtype=torch.float16, device=dev)
    out = torch.empty(a.batch, h * d, dtype=torch.float16, device=dev)
    ws = ops.decode_attn_workspace(a.batch, 1, h, d, len_buf, dev)

    if a.q8:
        kv = [torch.randint(0, 256, shape, dtype=torch.uint8, device=dev) for _ in range(a.batch)]
        sc = [torch.rand(shape[:-1], dtype=torch.float32, device=dev) * 0.03 + 0.005 for _ in range(a.batch)]
        k_addrs = torch.tensor([[t[l, 0].data_ptr() for t in kv] for l in range(a.layers)], dtype=torch.int64, device=dev)
        v_addrs = torch.tensor([[t[l, 1].data_ptr() for t in kv] for l in range(a.layers)], dtype=torch.int64, device=dev)
        ks_addrs = torch.tensor([[t[l, 0].data_ptr() for t in sc] for l in range(a.layers)], dtype=torch.int64, device=dev)
        vs_addrs = torch.tensor([[t[l, 1].data_ptr() for t in sc] for l in range(a.layers)], dtype=torch.int64, device=dev)
        qb = torch.empty(a.batch, h * d, dtype=torch.float16, device=dev)

    qb2 = torch.empty(a.batch, h * d, dtype=torch.float16, device=dev)

    def run():
        if a.unfused:
            for l in range(a.layers):
                ops.rope_scatter_decode(cos, sin, qkv, pos, buf_lens, k_addrs[l], v_addrs[l], h, hkv, d, bshd=not a.bhsd, q_out=qb2)
                ops.multi_query_attention_rag_buffer(qb2.view(a.batch, 1, h, d), buf_lens, k_addrs[l], v_addrs[l], None,
                                                     1.0 / math.sqrt(d), len_buf, hkv, valid_lens=valid, bshd=not a.bhsd,
                                                     out=out.view(a.batch, 1, h, d), workspace=ws)
            return
        if a.q8:
            for l in range(a.layers):
                ops.rope_quant_scatter_decode(cos, sin, qkv, pos, buf_lens, k_addrs[l], v_addrs[l], ks_addrs[l], vs_addrs[l],
                                              h, hkv, d, bshd=not a.bhsd, q_out=qb)
                ops.multi_query_attention_rag_buffer_quant(qb.view(a.batch, 1, h, d), buf_lens, k_addrs[l], v_addrs[l],
                                                           ks_addrs[l], vs_addrs[l], None, 1.0 / math.sqrt(d), len_buf, hkv,
                                                           valid_lens=valid, bshd=not a.bhsd,
                                                           out=out.view(a.batch, 1, h, d), workspace=ws)
            return
        for l in range(a.layers):
            ops.decode_attention_fused(cos, sin, qkv, pos, buf_lens, valid, k_addrs[l], v_addrs[l], h, hkv, d,
                                       1.0 / math.sqrt(d), len_buf, bshd=not a.bhsd, out=out, workspace=ws)
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (a.iters * a.layers)
    kvb = a.batch * 2 * hkv * (a.seq + 1) * (d + 4 if a.q8 else d * 2)
    print(f"batch={a.batch} seq={a.seq}: {us:7.2f} us/launch   KV {kvb / 1e6:.2f} MB -> {kvb / us / 1e3:7.1f} GB/s"
          "")


if __name__ == "__main__":
    main()

"""Config 5 (row f4) micro-benchmarks as ONE JSON line, run by bench.py in a child process (a fault here must not cost the headline):
MLA decode attention (zl_mla_decode_attn, bf16, partial + combine) and the FP8 128x128-block linear (zl_fp8_block_gemm_group) on
DeepSeek-V3's expert shapes, hipGraph replays timed with HIP events.  usage: python tools/bench_config5.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(reps):
            fn(i)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    out = {"mla_decode_us": {}, "fp8_block_gemm_us": {}, "note": "hipGraph replay, HIP events; MLA: 128 heads x 1024 keys, bf16, partial + "
           "combine launches; FP8 block GEMM: rotating HBM-cold weights, TB/s = weight bytes / time"}
    for b in (1, 8, 32):
        q = torch.randn(b, 128, 576, device=dev).to(torch.bfloat16)
        bufs = [torch.randn(1024, 576, device=dev).to(torch.bfloat16) for _ in range(b)]
        addrs = torch.tensor([t.data_ptr() for t in bufs], dtype=torch.int64, device=dev)
        lens = torch.full((b,), 1024, dtype=torch.int32, device=dev)
        out["mla_decode_us"]["batch_%d" % b] = round(timed(lambda i=0: ops.mla_decode_attention(q, lens, addrs, 0.1, 1024), 20), 2)
    for name, n, k in (("experts_gate_up_4096x7168", 4096, 7168), ("experts_down_7168x2048", 7168, 2048)):
        nb = max(8, -(-640_000_000 // (n * k)))      # rotating weights: > 2 x the 256 MiB Infinity Cache, so every launch streams from HBM
        ws = [torch.randint(0, 120, (n, k), dtype=torch.uint8, device=dev) for _ in range(nb)]
        sw = torch.rand((n + 127) // 128, k // 128, dtype=torch.float32, device=dev) * 0.01 + 0.001
        for m in (1, 32):
            x = torch.randn(m, k, device=dev).to(torch.bfloat16)
            a8, sa = ops.fp8_per_token_cast(x)
            y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            us = timed(lambda i=0: ops.fp8_block_gemm(a8, sa, ws[i % nb], sw, out=y), nb)
            out["fp8_block_gemm_us"]["%s_m%d" % (name, m)] = {"us": round(us, 2), "tb_per_s": round(n * k / us / 1e6, 2)}
        wps = [ops.Fp8BlockMWeight(w) for w in ws]      # the same codes in the ZLF8M layout (1 KiB contiguous fragment loads)
        del ws
        for m in (1, 32):
            x = torch.randn(m, k, device=dev).to(torch.bfloat16)
            a8, sa = ops.fp8_per_token_cast(x)
            y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            us = timed(lambda i=0: ops.fp8_block_gemm(a8, sa, wps[i % nb], sw, out=y), nb)
            out["fp8_block_gemm_us"]["%s_m%d_packed" % (name, m)] = {"us": round(us, 2), "tb_per_s": round(n * k / us / 1e6, 2)}
        del wps
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

"""FP8 128x128-block linear of config 5 (zl_fp8_block_gemm_group, the deep_gemm_fp8_block_h20_group binding) at decode row counts, on
DeepSeek-V3's projection shapes: time per launch (hipGraph over 8 rotating HBM-cold weights, HIP events) and the weight bytes per
second it amounts to.  usage: python tools/bench_fp8_block.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [("experts gate|up", 4096, 7168), ("experts down", 7168, 2048), ("q_b", 24576, 1536), ("kv_a + q_a", 2112, 7168), ("o", 7168, 16384)]
for name, n, k in shapes:
    nb = max(2, min(8, int(600e6 // (n * k))))          # > 256 MB of distinct weights where that is affordable
    ws = [torch.randint(0, 120, (n, k), dtype=torch.uint8, device=dev) for _ in range(nb)]
    sw = torch.rand((n + 127) // 128, k // 128, dtype=torch.float32, device=dev) * 0.01 + 0.001
    line = f"{name:16s} N={n:6d} K={k:6d} ({n * k / 1e6:6.1f} MB):"
    for m in (1, 8, 32, 64):
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        a8, sa = ops.fp8_per_token_cast(x)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        for w in ws:
            ops.fp8_block_gemm(a8, sa, w, sw, out=out)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(16):
                ops.fp8_block_gemm(a8, sa, ws[i % nb], sw, out=out)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 16
        line += f"  M={m:2d} {us:7.1f} us ({n * k / us / 1e6:5.2f} TB/s)"
    print(line, flush=True)
    del ws

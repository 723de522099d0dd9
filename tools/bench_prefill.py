"""Prompt encode (TTFT) of the Llama-3-8B GPTQ-Int4 model: python tools/bench_prefill.py [--seq 1024] [--layers 32]
(run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seq", type=int, default=1024)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = ModelConfig.llama3_8b()
cfg.num_layers = a.layers
model = LLaMA(cfg, QuantConfig(5, 128), dev).init_random(seed=1)
len_buf = (a.seq + 64 + 63) // 64 * 64
ctx = model.new_context(1, len_buf, 0)
prompt = torch.randint(0, cfg.vocab_size, (a.seq,), device=dev, dtype=torch.int32)
model.prefill(ctx, 0, prompt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    model.prefill(ctx, 0, prompt)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
flops = 2.0 * a.seq * sum(l.weight.n * l.weight.k for lay in model.layers for l in (lay.qkv, lay.attn_out, lay.w_in_gated, lay.w_out))
print(f"prefill seq={a.seq} layers={a.layers}: {ms:.2f} ms  ({flops / ms / 1e9:.1f} TFLOP/s on the linears alone)")

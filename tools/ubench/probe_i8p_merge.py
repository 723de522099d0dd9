"""Timeline of the MERGING attn_out launch (k_w4a16_i8p<.., MERGE>) inside (decode attention splits -> merging projection) pairs
under hipGraph replay: the partials are freshly written by the attention kernel of the same pair, the projection weights rotate
through 8 HBM-cold buffers.  Probe build:  tools/ubench/variant.sh iprobe zhilight_amd/csrc/w4_i8p.hip -DZL_I8P_PROBE
usage: ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libiprobe.so python tools/ubench/probe_i8p_merge.py
(the early-ring variant it was written to compare against is profiles/r03_l2_hint_i8p.patch)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.bench_gemv import rand_w4m  # noqa: E402
from zhilight_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
h, hkv, d, L, valid = 32, 8, 128, 1152, 1025
names = ["entry", "loads issued", "(norm)", "merged+staged", "first item done", "stream done", "reduce barrier", "end"]
ws8 = [rand_w4m(4096, 4096, 128, dev) for _ in range(8)]
dk, dv = [torch.randn(L, hkv, d, device=dev).half()], [torch.randn(L, hkv, d, device=dev).half()]
q = torch.randn(1, 1, h, d, device=dev).half()
bl, vl = torch.tensor([L], dtype=torch.int32, device=dev), torch.tensor([valid], dtype=torch.int32, device=dev)
ka, va = ops.make_ptr_table(dk), ops.make_ptr_table(dv)
plan = ops.attn_merge_plan(1, h, hkv, d, L, ws8[0])
ws = ops.decode_attn_workspace(1, 1, h, d, L, dev)
hidden = torch.zeros(1, 4096, dtype=torch.float16, device=dev)


def pair(i):
    ops.decode_attention_splits(q, bl, ka, va, vl, 0.088, L, hkv, ws)
    ops.w4_attn_out_merge(ws, bl, vl, plan, 1, ws8[i % 8], residual=hidden, out=hidden, epilogue=ops.EPI_RESIDUAL)


for i in range(8):
    pair(i)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for i in range(16):
        pair(i)
gr.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    gr.replay()
e1.record()
torch.cuda.synchronize()
print(f"plan {plan}; (attention + merging projection) pair: {e0.elapsed_time(e1) / 20 / 16 * 1e3:.2f} us")
probe = torch.zeros(2048 * 8 * 8, dtype=torch.int64, device=dev)
_lib.lib().zl_debug_set_probe_i8p(C.c_void_p(probe.data_ptr()))
gr.replay()
torch.cuda.synchronize()
_lib.lib().zl_debug_set_probe_i8p(C.c_void_p(0))
t = probe.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] > 0].astype(np.float64)
t0 = t[:, 0].min()
t = np.where(t > 0, (t - t0) * 10.0, np.nan)
print(f"merging projection: waves={len(t)}  span {np.nanmax(t[:, 7]) / 1e3:.2f} us")
for i, nm in enumerate(names):
    c = t[:, i]
    if np.all(np.isnan(c)):
        continue
    print(f"  {nm:18s} min {np.nanmin(c) / 1e3:6.2f}  p10 {np.nanpercentile(c, 10) / 1e3:6.2f}  median {np.nanmedian(c) / 1e3:6.2f}"
          f"  p90 {np.nanpercentile(c, 90) / 1e3:6.2f}  max {np.nanmax(c) / 1e3:6.2f} us")

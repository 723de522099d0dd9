"""Timeline of k_w4a16_i8p from in-kernel wall-clock stamps (100 MHz), taken on the LAST launch of a replayed hipGraph chain
(HBM-cold weights rotated through 8 buffers, warm instruction cache: what a decode step sees).
Build the probe variant first:  tools/ubench/variant.sh iprobe zhilight_amd/csrc/w4_i8p.hip -DZL_I8P_PROBE
usage: ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libiprobe.so python tools/ubench/probe_i8p.py [M]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.bench_gemv import rand_w4m  # noqa: E402
from zhilight_amd import _lib, ops  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
names = ["entry", "x landed", "ring issued", "staged+barrier", "first item done", "stream done", "reduce barrier", "end"]
cases = [("qkv+norm", 6144, 4096, 0, True, False), ("qkv plain", 6144, 4096, 0, False, False),
         ("o+residual", 4096, 4096, ops.EPI_RESIDUAL, False, True),
         ("gate|up+norm+silu", 28672, 4096, ops.EPI_SILU_MUL, True, False), ("down+residual", 4096, 14336, ops.EPI_RESIDUAL, False, True)]
for label, n, k, epi, norm, resid in cases:
    ws = [rand_w4m(n, k, 128, dev, interleave=bool(epi & ops.EPI_SILU_MUL)) for _ in range(8)]
    x = torch.randn(m, k, dtype=torch.float16, device=dev)
    out = torch.zeros(m, n // 2 if epi & ops.EPI_SILU_MUL else n, dtype=torch.float16, device=dev)
    nw = torch.ones(k, dtype=torch.float16, device=dev)
    kw = {}
    if norm:
        kw.update(norm_weight=nw, norm_eps=1e-5)
    if resid:
        kw.update(residual=out)
    probe = torch.zeros(2048 * 8 * 8, dtype=torch.int64, device=dev)
    for w in ws:
        ops.w4a16_gemm_mfma(x, w, out=out, epilogue=epi, **kw)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(24):
            ops.w4a16_gemm_mfma(x, ws[i % 8], out=out, epilogue=epi, **kw)
    gr.replay()
    torch.cuda.synchronize()
    _lib.lib().zl_debug_set_probe_i8p(C.c_void_p(probe.data_ptr()))
    probe.zero_()
    gr.replay()
    torch.cuda.synchronize()
    t = probe.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0].astype(np.float64)
    t0 = t[:, 0].min()
    t = np.where(t > 0, (t - t0) * 10.0, np.nan)
    print(f"{label}: N={n} K={k} M={m} waves={len(t)}  span {np.nanmax(t[:, 7]) / 1e3:.2f} us")
    for i, nm in enumerate(names):
        c = t[:, i]
        if np.all(np.isnan(c)):
            continue
        print(f"  {nm:18s} min {np.nanmin(c) / 1e3:6.2f}  p10 {np.nanpercentile(c, 10) / 1e3:6.2f}  median {np.nanmedian(c) / 1e3:6.2f}"
              f"  p90 {np.nanpercentile(c, 90) / 1e3:6.2f}  max {np.nanmax(c) / 1e3:6.2f} us")
    _lib.lib().zl_debug_set_probe_i8p(C.c_void_p(0))
    del ws

"""Timeline of the slab kernel (k_w4a16_slab) from in-kernel wall-clock stamps (100 MHz), per Llama-3-8B shape.
Build the probe variant first:  tools/ubench/variant.sh sprobe zhilight_amd/csrc/w4_slab.hip -DZL_SLAB_PROBE -Iinclude -Izhilight_amd/csrc
usage: ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libsprobe.so python tools/ubench/probe_slab.py [M ...]
HBM-cold weights (a 512 MB spoiler between the warm-up and the stamped launch).  Per stamp: min / p10 / median / p90 / max over
all waves, relative to the first wave's entry."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zhilight_amd import _lib, ops  # noqa: E402

ms = [int(a) for a in sys.argv[1:]] or [32]
dev = torch.device("cuda:0")
names = ["entry", "prologue requested", "group 0 in registers", "group 1", "group 2", "group 3", "items done", "tile summed"]
cases = [("qkv", 6144, 4096, 0, False, False), ("o+residual", 4096, 4096, ops.EPI_RESIDUAL, True, False),
         ("gate|up+silu", 28672, 4096, ops.EPI_SILU_MUL, False, False), ("down+residual", 4096, 14336, ops.EPI_RESIDUAL, True, False),
         ("norm+gate|up+silu", 28672, 4096, ops.EPI_SILU_MUL, False, True)]
L = _lib.lib()
L.zl_debug_set_slab_probe.argtypes = [C.c_void_p]
for m in ms:
    for label, n, k, epi, resid, norm in cases:
        ws = [ops.W4MWeight.random(n, k, 128, dev, row_interleave=bool(epi & ops.EPI_SILU_MUL)) for _ in range(2)]
        x = torch.randn(m, k, dtype=torch.float16, device=dev)
        out = torch.zeros(m, n // 2 if epi & ops.EPI_SILU_MUL else n, dtype=torch.float16, device=dev)
        kw = dict(residual=out) if resid else {}
        if norm:
            kw.update(norm_weight=torch.ones(k, dtype=torch.float16, device=dev), row_ss=ops.row_ss(x))
        for w in ws:
            ops.w4a16_gemm_mfma(x, w, out=out, epilogue=epi, **kw)
        torch.cuda.synchronize()
        spoil = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        spoil.fill_(1)
        torch.cuda.synchronize()
        probe = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
        L.zl_debug_set_slab_probe(C.c_void_p(probe.data_ptr()))
        ops.w4a16_gemm_mfma(x, ws[0], out=out, epilogue=epi, **kw)
        torch.cuda.synchronize()
        L.zl_debug_set_slab_probe(C.c_void_p(0))
        t = probe.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 0] > 0].astype(np.float64)
        if not len(t):
            print(f"{label}: M={m}: no stamps (not the slab kernel?)")
            continue
        t0 = t[:, 0].min()
        t = np.where(t > 0, (t - t0) * 10.0, np.nan)
        print(f"{label}: N={n} K={k} M={m} waves={len(t)}  span {np.nanmax(t) / 1e3:.2f} us")
        for i, nm in enumerate(names):
            c = t[:, i]
            if np.all(np.isnan(c)):
                continue
            print(f"  {nm:22s} min {np.nanmin(c) / 1e3:6.2f}  p10 {np.nanpercentile(c, 10) / 1e3:6.2f}  median {np.nanmedian(c) / 1e3:6.2f}"
                  f"  p90 {np.nanpercentile(c, 90) / 1e3:6.2f}  max {np.nanmax(c) / 1e3:6.2f} us")
        del ws, spoil

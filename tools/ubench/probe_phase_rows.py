"""Timeline of the 5..32-row W4A16 decode launches from in-kernel wall-clock stamps (100 MHz), per Llama-3-8B shape.
Build the probe variant first:  tools/ubench/variant.sh pprobe zhilight_amd/csrc/w4_phase.hip -DZL_PHASE_PROBE
usage: ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libpprobe.so python tools/ubench/probe_phase_rows.py [M ...]
The plain (no fused norm) launches a batch-M step issues: qkv-shape, o + residual, gate|up + silu.mul, down + residual; HBM-cold
weights (a 512 MB spoiler between the warm-up and the stamped launch).  Per stamp: min / p10 / median / p90 / max over all waves,
relative to the first wave's entry."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.bench_gemv import rand_w4m  # noqa: E402
from zhilight_amd import _lib, ops  # noqa: E402

ms = [int(a) for a in sys.argv[1:]] or [32]
dev = torch.device("cuda:0")
names = ["entry", "ring issued", "x staged+barrier", "first item done", "stream done", "parked+barrier", "end", "aux"]
cases = [("qkv", 6144, 4096, 0, False), ("o+residual", 4096, 4096, ops.EPI_RESIDUAL, True),
         ("gate|up+silu", 28672, 4096, ops.EPI_SILU_MUL, False), ("down+residual", 4096, 14336, ops.EPI_RESIDUAL, True)]
L = _lib.lib()
L.zl_debug_set_probe_p.argtypes = [C.c_void_p, C.c_int]
for m in ms:
    for label, n, k, epi, resid in cases:
        ws = [rand_w4m(n, k, 128, dev, interleave=bool(epi & ops.EPI_SILU_MUL)) for _ in range(3)]
        x = torch.randn(m, k, dtype=torch.float16, device=dev)
        out = torch.zeros(m, n // 2 if epi & ops.EPI_SILU_MUL else n, dtype=torch.float16, device=dev)
        kw = dict(residual=out) if resid else {}
        probe = torch.zeros(4 * 16384 * 8, dtype=torch.int64, device=dev)
        for w in ws:
            ops.w4a16_gemm_mfma(x, w, out=out, epilogue=epi, **kw)
        torch.cuda.synchronize()
        spoil = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        spoil.fill_(1)
        torch.cuda.synchronize()
        L.zl_debug_set_probe_p(C.c_void_p(probe.data_ptr()), C.c_int(L.zl_debug_probe_seq()))
        ops.w4a16_gemm_mfma(x, ws[0], out=out, epilogue=epi, **kw)
        torch.cuda.synchronize()
        L.zl_debug_set_probe_p(C.c_void_p(0), C.c_int(0))
        t = probe.cpu().numpy().reshape(4, -1, 8)[0]
        t = t[t[:, 0] > 0].astype(np.float64)
        if not len(t):
            print(f"{label}: M={m}: no stamps (not the phase kernel?)")
            continue
        t0 = t[:, 0].min()
        t = np.where(t > 0, (t - t0) * 10.0, np.nan)
        print(f"{label}: N={n} K={k} M={m} waves={len(t)}  span {np.nanmax(t[:, 6]) / 1e3:.2f} us")
        for i, nm in enumerate(names):
            c = t[:, i]
            if np.all(np.isnan(c)):
                continue
            print(f"  {nm:18s} min {np.nanmin(c) / 1e3:6.2f}  p10 {np.nanpercentile(c, 10) / 1e3:6.2f}  median {np.nanmedian(c) / 1e3:6.2f}"
                  f"  p90 {np.nanpercentile(c, 90) / 1e3:6.2f}  max {np.nanmax(c) / 1e3:6.2f} us")
        del ws, spoil

"""Phase timeline of the MFMA W4A16 kernel from in-kernel wall-clock stamps.
Build the probe variant first:  tools/ubench/variant.sh probe zhilight_amd/csrc/w4_mfma.hip -DZL_W4M_PROBE
usage: ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libprobe.so python tools/ubench/probe_mfma.py N K [M]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.bench_gemv import rand_w4m  # noqa: E402
from zhilight_amd import _lib, ops  # noqa: E402

n, k = int(sys.argv[1]), int(sys.argv[2])
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
ws = [rand_w4m(n, k, 128, dev) for _ in range(4)]
x = torch.randn(m, k, dtype=torch.float16, device=dev)
out = torch.empty(m, n, dtype=torch.float16, device=dev)
probe = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
for w in ws:
    ops.w4a16_gemm_mfma(x, w, out=out)
torch.cuda.synchronize()
_lib.lib().zl_debug_set_probe_m(C.c_void_p(probe.data_ptr()))
ops.w4a16_gemm_mfma(x, ws[0], out=out)
torch.cuda.synchronize()
t = probe.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] > 0][:, :7].astype(np.float64)
t0 = t[:, 0].min()
t = np.where(t > 0, (t - t0) * 10.0, np.nan)  # 100 MHz -> ns
print(f"N={n} K={k} M={m} waves={len(t)}  kernel span {np.nanmax(t[:, 6]) / 1e3:.2f} us")
names = ["start", "ring issued", "x staged (barrier)", "first ring turn done", "stream done", "after barrier", "end"]
for i, nm in enumerate(names):
    c = t[:, i]
    print(f"  {nm:22s} min {np.nanmin(c) / 1e3:7.2f}  median {np.nanmedian(c) / 1e3:7.2f}  max {np.nanmax(c) / 1e3:7.2f} us")

"""Phase timeline of the W4A16 GEMV kernel from in-kernel wall-clock stamps (debug build, -DZL_W4_PROBE).
usage: python tools/ubench/probe_gemv.py N K [wgs_per_cu]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
so = "/tmp/libzl_probe.so"
srcs = [os.path.join(ROOT, "zhilight_amd/csrc", f) for f in ("w4_gemv.hip", "w4_layout.hip", "misc_ops.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
                       "-ffp-contract=off", "-DZL_W4_PROBE", "-o", so] + srcs)
from zhilight_amd import _lib  # noqa: E402
_lib.SO_PATH = so
_lib.SYMBOLS = [s for s in _lib.SYMBOLS if s in ("zl_version", "zl_status_string", "zl_device_cu_count", "zl_w4_layout", "zl_w4_pack",
                                                 "zl_w4_dequant", "zl_w4a16_gemm", "zl_decode_attn_workspace_bytes")]
_lib.SYMBOLS.remove("zl_decode_attn_workspace_bytes")
l = C.CDLL(so)
l.zl_status_string.restype = C.c_char_p
l.zl_decode_attn_workspace_bytes = lambda *a: 0
_lib._lib = l
from tools.bench_gemv import rand_w4  # noqa: E402
from zhilight_amd import ops  # noqa: E402

n, k = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    os.environ["ZL_W4_WGS_PER_CU"] = sys.argv[3]
dev = torch.device("cuda:0")
ws = [rand_w4(n, k, 128, dev) for _ in range(4)]
x = torch.randn(1, k, dtype=torch.float16, device=dev)
out = torch.empty(1, n, dtype=torch.float16, device=dev)
nw = torch.ones(k, dtype=torch.float16, device=dev)
probe = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
for w in ws:
    ops.w4a16_gemm(x, w, out=out, norm_weight=nw)
torch.cuda.synchronize()
l.zl_debug_set_probe(C.c_void_p(probe.data_ptr()))
ops.w4a16_gemm(x, ws[0], out=out, norm_weight=nw)
torch.cuda.synchronize()
t = probe.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] > 0][:, :5].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) * 10.0  # 100 MHz -> ns
print(f"N={n} K={k} waves={len(t)}  kernel span {t[:, 4].max() / 1e3:.2f} us")
names = ["start", "loads issued", "x staged (barrier)", "main loop done", "end"]
for i, nm in enumerate(names):
    print(f"  {nm:22s} min {t[:, i].min() / 1e3:7.2f}  median {np.median(t[:, i]) / 1e3:7.2f}  max {t[:, i].max() / 1e3:7.2f} us")
d = np.diff(t, axis=1)
for i, nm in enumerate(["issue", "stage+barrier", "main loop", "epilogue"]):
    print(f"  phase {nm:14s} median {np.median(d[:, i]) / 1e3:6.2f} us  p90 {np.percentile(d[:, i], 90) / 1e3:6.2f}")

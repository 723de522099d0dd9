"""Timeline of the in-launch-merge decode attention (k_decode_attn_mfma<.., LA>) from in-kernel wall-clock stamps (100 MHz).
Build the probe variant first:  tools/ubench/variant.sh aprobe zhilight_amd/csrc/attention.hip -DZL_ATTN_PROBE
usage: ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libaprobe.so python tools/ubench/probe_attn_la.py [batch ...]
Llama-3-8B geometry (32 / 8 heads, 128), 1025 visible keys in 1088-slot buffers, HBM-cold K / V (a 512 MB spoiler in between).
Per stamp: min / p10 / median / p90 / max over all waves, relative to the first wave's entry."""
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zhilight_amd import _lib, ops  # noqa: E402

bs = [int(a) for a in sys.argv[1:]] or [8, 32]
dev = torch.device("cuda:0")
names = ["entry", "first K/V requested", "first scores", "keys done", "waves merged", "record out + drained", "ticket known", "last arriver done"]
L = _lib.lib()
L.zl_debug_set_attn_probe.argtypes = [C.c_void_p]
h, hkv, d, slots, valid = 32, 8, 128, 1088, 1025
for b in bs:
    ks = [torch.randn(slots, hkv, d, dtype=torch.float16, device=dev) for _ in range(b)]
    vs = [torch.randn(slots, hkv, d, dtype=torch.float16, device=dev) for _ in range(b)]
    ka, va = ops.make_ptr_table(ks), ops.make_ptr_table(vs)
    q = torch.randn(b, 1, h, d, dtype=torch.float16, device=dev)
    lens = torch.full((b,), slots, dtype=torch.int32, device=dev)
    vl = torch.full((b,), valid, dtype=torch.int32, device=dev)
    ws = ops.decode_attn_la_workspace(b, h, hkv, slots, dev)
    split = int(os.environ.get("ZL_ATTN_LA_SPLIT", "0") or 0)
    run = lambda: ops.decode_attention_la(q, lens, ka, va, vl, 1.0 / math.sqrt(d), slots, hkv, ws, split_len=split)
    run()
    torch.cuda.synchronize()
    spoil = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    spoil.fill_(1)
    torch.cuda.synchronize()
    probe = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
    L.zl_debug_set_attn_probe(C.c_void_p(probe.data_ptr()))
    run()
    torch.cuda.synchronize()
    L.zl_debug_set_attn_probe(C.c_void_p(0))
    t = probe.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0].astype(np.float64)
    t0 = t[:, 0].min()
    t = np.where(t > 0, (t - t0) * 10.0, np.nan)
    eff = split or ops.decode_attn_la_split_len(b, hkv, slots)
    print(f"batch {b}: split {eff} keys, waves {len(t)}, span {np.nanmax(t) / 1e3:.2f} us")
    for i, nm in enumerate(names):
        c = t[:, i]
        if np.all(np.isnan(c)):
            continue
        print(f"  {nm:22s} n {int(np.sum(~np.isnan(c))):5d}  min {np.nanmin(c) / 1e3:6.2f}  p10 {np.nanpercentile(c, 10) / 1e3:6.2f}  median {np.nanmedian(c) / 1e3:6.2f}"
              f"  p90 {np.nanpercentile(c, 90) / 1e3:6.2f}  max {np.nanmax(c) / 1e3:6.2f} us")
    del ks, vs, spoil

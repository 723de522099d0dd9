"""Timeline of the loader / consumer engine kernels (w4_engine.hip) from in-kernel wall-clock stamps (100 MHz), taken on the
LAST launch of a replayed hipGraph chain (HBM-cold weights rotated through 8 buffers).
Build the probe variant first:  tools/ubench/variant.sh eprobe zhilight_amd/csrc/w4_engine.hip -DZL_ENG_PROBE
usage: ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libeprobe.so python tools/ubench/probe_engine.py [loaders]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zhilight_amd import _lib, ops  # noqa: E402

loaders = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["ZL_W4_SMALL_ALGO"] = "2"
dev = torch.device("cuda:0")
_lib.lib().zl_debug_engine_knobs(0)
loaders = 1
cases = [("qkv+norm", 6144, 4096, 0, True, False, 8), ("o+residual", 4096, 4096, ops.EPI_RESIDUAL, False, True, 4),
         ("gate|up+norm+silu", 28672, 4096, ops.EPI_SILU_MUL, True, False, 28), ("down+residual", 4096, 14336, ops.EPI_RESIDUAL, False, True, 14)]
for label, n, k, epi, norm, resid, nslots in cases:
    ws = [ops.W4MWeight.random(n, k, 128, dev, row_interleave=bool(epi & ops.EPI_SILU_MUL)) for _ in range(8)]
    x = torch.randn(1, k, dtype=torch.float16, device=dev)
    out = torch.zeros(1, n // 2 if epi & ops.EPI_SILU_MUL else n, dtype=torch.float16, device=dev)
    nw = torch.ones(k, dtype=torch.float16, device=dev)
    kw = {}
    if norm:
        kw.update(norm_weight=nw, norm_eps=1e-5)
    if resid:
        kw.update(residual=out)
    probe = torch.zeros(256 * 10 * 2 * 68, dtype=torch.int64, device=dev)
    for w in ws:
        ops.w4a16_gemm_mfma(x, w, out=out, epilogue=epi, **kw)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(24):
            ops.w4a16_gemm_mfma(x, ws[i % 8], out=out, epilogue=epi, **kw)
    gr.replay()
    torch.cuda.synchronize()
    _lib.lib().zl_debug_set_probe_engine(C.c_void_p(probe.data_ptr()))
    probe.zero_()
    gr.replay()
    torch.cuda.synchronize()
    _lib.lib().zl_debug_set_probe_engine(C.c_void_p(0))
    t = probe.cpu().numpy().reshape(256, 10, 2, 68).astype(np.float64)
    live = t[:, 0, 0, 0] > 0
    t = t[live]
    t0 = t[:, :, 0, 0][t[:, :, 0, 0] > 0].min()
    t = np.where(t > 0, (t - t0) * 0.01, np.nan)          # us
    med = lambda a: float(np.nanmedian(a))
    print(f"== {label}: N={n} K={k}, {int(live.sum())} workgroups, {loaders} loader(s), {nslots} slots per workgroup")
    print(f"  consumers: entry {med(t[:, :8, 0, 0]):.2f}  past barrier {med(t[:, :8, 0, 1]):.2f}  x landed {med(t[:, :8, 0, 2]):.2f}  staged {med(t[:, :8, 0, 3]):.2f}")
    print(f"  loader   : at barrier {med(t[:, 8, 0, 0]):.2f}  past barrier {med(t[:, 8, 0, 1]):.2f}")
    ent = np.nanmin(t[:, :8, 0, 0], axis=1)
    end = np.nanmax(t[:, :8, 1, 4:4 + nslots], axis=(1, 2))
    q = lambda a, p_: float(np.nanpercentile(a, p_))
    print(f"  workgroup entry: min {np.nanmin(ent):.2f} p50 {q(ent, 50):.2f} p90 {q(ent, 90):.2f} p99 {q(ent, 99):.2f} max {np.nanmax(ent):.2f}   "
          f"last fold per workgroup: min {np.nanmin(end):.2f} p50 {q(end, 50):.2f} p90 {q(end, 90):.2f} p99 {q(end, 99):.2f} max {np.nanmax(end):.2f}")
    idx = np.nonzero(live)[0]
    print("  last fold by workgroup index mod 8 (XCD): " + " ".join(f"{float(np.nanmedian(end[idx % 8 == x])):.1f}" for x in range(8)))
    print("  last fold by workgroup index // 32      : " + " ".join(f"{float(np.nanmedian(end[idx // 32 == x])):.1f}" for x in range(8)))
    print("  last fold, workgroups 0..31: " + " ".join(f"{end[i]:.0f}" for i in range(min(32, len(end)))))
    pub = t[:, 8, 1, 4:4 + nslots]
    print("  last publish per workgroup: min %.2f p50 %.2f p90 %.2f max %.2f" % (np.nanmin(np.nanmax(pub, axis=1)), q(np.nanmax(pub, axis=1), 50), q(np.nanmax(pub, axis=1), 90), np.nanmax(pub)))
    lat = t[:, 8, 1, 4:4 + nslots] - t[:, 8, 0, 4:4 + nslots] if loaders == 1 else None
    if lat is not None:
        print("  issue -> published latency by slot (median over workgroups): " + " ".join(f"{med(lat[:, i]):.2f}" for i in range(nslots)))
        print("  the same, by workgroup (slot 10 or the last): p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(
            [q(lat[:, min(10, nslots - 1)], p_) for p_ in (10, 50, 90)] + [float(np.nanmax(lat[:, min(10, nslots - 1)]))]))
    print("  slot   issued  published   fetched(w0..7 median)  folded   (us after the first wave's entry, medians over workgroups)")
    for s_ in range(nslots):
        li = 8 + (s_ % loaders)
        j = s_ // loaders
        print(f"  {s_:4d}  {med(t[:, li, 0, 4 + s_]):7.2f}  {med(t[:, li, 1, 4 + j]):9.2f}   {med(t[:, :8, 0, 4 + s_]):12.2f}       {med(t[:, :8, 1, 4 + s_]):7.2f}")
    print(f"  last fold: median {med(np.nanmax(t[:, :8, 1, 4:4 + nslots], axis=2)):.2f}  max {float(np.nanmax(t[:, :8, 1, 4:4 + nslots])):.2f}")
    del ws


# ---- the fused attn_out -> gate|up launch
import math
h, hkv, d, dm, ff, lens = 32, 8, 128, 4096, 14336, [1088]
w_o = [ops.W4MWeight.random(dm, h * d, 128, dev) for _ in range(8)]
w_ff = [ops.W4MWeight.random(2 * ff, dm, 128, dev, row_interleave=True) for _ in range(8)]
ln = torch.ones(dm, dtype=torch.float16, device=dev)
dk = [torch.randn(L_, hkv, d, device=dev).half() for L_ in lens]
dv = [torch.randn(L_, hkv, d, device=dev).half() for L_ in lens]
q = torch.randn(1, 1, h, d, device=dev).half()
bl = torch.tensor(lens, dtype=torch.int32, device=dev)
vl = torch.tensor([1025], dtype=torch.int32, device=dev)
plan = ops.attn_merge_plan(1, h, hkv, d, max(lens), w_o[0])
ws = ops.decode_attn_workspace(1, 1, h, d, max(lens), dev)
ka, va = ops.make_ptr_table(dk), ops.make_ptr_table(dv)
hidden = torch.randn(1, dm, device=dev).half()
act = torch.empty(1, ff, dtype=torch.float16, device=dev)


def chain():
    ops.engine_epoch_advance(dev)
    for i in range(8):
        ops.decode_attention_splits(q, bl, ka, va, vl, 1.0 / math.sqrt(d), max(lens), hkv, ws)
        assert ops.w4_attn_out_gate_up(ws, bl, vl, plan, 1, w_o[i], hidden, w_ff[i], ln, 1e-5, act, i)


chain()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    chain()
gr.replay()
torch.cuda.synchronize()
probe = torch.zeros(256 * 10 * 2 * 68, dtype=torch.int64, device=dev)
_lib.lib().zl_debug_set_probe_engine(C.c_void_p(probe.data_ptr()))
probe.zero_()
gr.replay()
torch.cuda.synchronize()
_lib.lib().zl_debug_set_probe_engine(C.c_void_p(0))
t = probe.cpu().numpy().reshape(256, 10, 2, 68).astype(np.float64)
t0 = t[:, :, 0, 0][t[:, :, 0, 0] > 0].min()
t = np.where(t > 0, (t - t0) * 0.01, np.nan)
med = lambda a: float(np.nanmedian(a))
mx = lambda a: float(np.nanmax(a))
print("== fused attn_out -> gate|up (32 slots per workgroup: 4 + 28), us after the first wave's entry, medians over workgroups (max)")
print(f"  consumers: at barrier {med(t[:, :8, 0, 0]):.2f}  merged {med(t[:, :8, 0, 2]):.2f}  staged {med(t[:, :8, 0, 3]):.2f}")
print(f"  phase 1 folds: " + " ".join(f"{med(t[:, :8, 1, 4 + i]):.2f}" for i in range(4)) + f"   epilogue done / publishing {med(t[:, 0, 1, 0]):.2f} ({mx(t[:, 0, 1, 0]):.2f})")
print(f"  phase 2: hidden row gathered {med(t[:, :8, 1, 2]):.2f} ({mx(t[:, :8, 1, 2]):.2f})  staged {med(t[:, :8, 1, 3]):.2f}")
print("  slot: issued / published / folded")
for s_ in range(32):
    print(f"  {s_:3d} {med(t[:, 8, 0, 4 + s_]):7.2f} {med(t[:, 8, 1, 4 + s_]):7.2f} {med(t[:, :8, 1, 4 + s_]):7.2f}")
print(f"  last fold: median {med(np.nanmax(t[:, :8, 1, 4:36], axis=(1, 2))):.2f} max {mx(t[:, :8, 1, 4:36]):.2f}")

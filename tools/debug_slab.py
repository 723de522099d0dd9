"""debug aid: w4_slab.hip against the phase kernel on the same operands, per forced geometry; prints which rows / columns differ"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
keys = ("ZL_W4_SLAB", "ZL_W4_SLAB_R", "ZL_W4_SLAB_NW", "ZL_W4_SLAB_GPW")
torch.manual_seed(0)
for (k, n) in [(2048, 264), (4096, 512), (1152, 184)]:
    w = ops.W4MWeight.random(n, k, 128, dev)
    for m in (9, 16, 17, 32):
        x = torch.randn(m, k, dtype=torch.float16, device=dev)
        for kk in keys:
            os.environ.pop(kk, None)
        os.environ["ZL_W4_SLAB"] = "-1"
        ref = ops.w4a16_gemm_mfma(x, w).float()
        os.environ.pop("ZL_W4_SLAB")
        cfgs = [None] + [(r, nw, gpw) for r in (1, 2, 4, 8) for nw in (4, 8) for gpw in (1, 2, 4)]
        for cfg in cfgs:
            for kk in keys:
                os.environ.pop(kk, None)
            if cfg:
                os.environ.update({"ZL_W4_SLAB_R": str(cfg[0]), "ZL_W4_SLAB_NW": str(cfg[1]), "ZL_W4_SLAB_GPW": str(cfg[2])})
            try:
                y = ops.w4a16_gemm_mfma(x, w).float()
            except Exception as e:  # noqa: BLE001
                print(f"k={k} n={n} m={m} cfg={cfg}: {str(e)[:60]}")
                continue
            torch.cuda.synchronize()
            d = (y - ref).abs()
            bad = d > 0.02 * ref.abs().mean()
            if bool(bad.any()):
                rows = sorted(set(torch.nonzero(bad)[:, 0].tolist()))
                cols = sorted(set(torch.nonzero(bad)[:, 1].tolist()))
                print(f"k={k} n={n} m={m} cfg={cfg}: BAD rows {rows[:20]} ({len(rows)})  cols {cols[:12]}.. ({len(cols)})  max {float(d.max()):.3f}")
            else:
                print(f"k={k} n={n} m={m} cfg={cfg}: ok  max diff {float(d.max()):.4f}")

import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from zhilight_amd import ops
import synth
dev = torch.device("cuda:0")
def rs_from_ss(ss, k, eps):
    m, parts = ss.shape
    p = ss.reshape(m, parts // 64, 16, 4)
    t = np.zeros((m, 16), np.float32)
    for u in range(parts // 64):
        t = t + ((p[:, u, :, 0] + p[:, u, :, 1]) + (p[:, u, :, 2] + p[:, u, :, 3]))
    while t.shape[-1] > 1:
        t = t[..., 0::2] + t[..., 1::2]
    val = t[:, 0] / np.float32(k) + np.float32(eps)
    return (np.float32(1.0) / np.sqrt(val, dtype=np.float32)).astype(np.float32)
import itertools
for (m, n, k, epi) in [(5, 32768, 4096, 0), (8, 32768, 4096, 0), (16, 32768, 4096, 0), (8, 28672, 4096, 0), (32, 32768, 4096, 0)]:
    rng = np.random.default_rng(1)
    w = ops.W4MWeight.random(n, k, 128, dev)
    x = synth.act(rng, m, k, 2.0)
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    xt = torch.from_numpy(x).to(dev)
    ss = ops.row_ss(xt)
    rs = rs_from_ss(ss.cpu().numpy(), k, 1e-5)
    xn = ((x.astype(np.float32) * rs[:, None]) * nw.astype(np.float32)[None, :]).astype(np.float16)
    os.environ["ZL_W4_SLAB"] = "2"
    got = ops.w4a16_gemm_mfma(xt, w, norm_weight=torch.from_numpy(nw).to(dev), norm_eps=1e-5, row_ss=ss).float().cpu().numpy()
    want = ops.w4a16_gemm_mfma(torch.from_numpy(xn).to(dev), w).float().cpu().numpy()
    bad = got != want
    print(m, n, k, "mismatch", bad.sum(), "of", bad.size)
    if bad.any():
        rows = np.unique(np.nonzero(bad)[0]); cols = np.nonzero(bad.any(axis=0))[0]
        print(" rows", rows[:20], " cols%112 hist", np.bincount((cols // 16) % (7 if n == 28672 else 8), minlength=8), "n cols", len(cols), "first cols", cols[:12])
        d = np.abs(got - want)[bad]
        print(" max diff", d.max(), "median", np.median(d), "rel", (d / (np.abs(want[bad]) + 1e-6)).max())
        # does it match a different normalisation (e.g. unnormalised groups)?

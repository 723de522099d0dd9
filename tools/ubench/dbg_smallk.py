import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import zl_oracle as oracle
import synth
from zhilight_amd import ops
dev = torch.device('cuda:0')
for (m, k, n) in [(384, 128, 256), (256, 256, 4096), (256, 512, 4096)]:
    rng = np.random.default_rng(70 + m)
    g = 128
    qw, qz, sc = synth.gptq_hf(rng, k, n, g)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, g)
    w = ops.W4MWeight.from_k_major(torch.from_numpy(km[0].view(np.int32)).to(dev), torch.from_numpy(km[1]).to(dev), torch.from_numpy(km[2]).view(torch.float16).to(dev), g)
    x = synth.act(rng, m, k)
    w16 = oracle.gptq_dequant_k_major(*km)
    ref = oracle.gemm_nt(oracle.h2u(x), w16, None, exact=True)
    y = ops.w4a16_gemm_tiled(torch.from_numpy(x).to(dev), w).float().cpu().numpy().astype(np.float64)
    d = np.abs(y - ref); rms = np.sqrt((ref ** 2).mean())
    bound = 2.0 ** -10 * np.abs(ref) + 2e-5 * rms
    bad = d > bound
    print(m, k, n, 'rms', rms, 'max d', d.max(), 'bad', bad.sum(), 'max ratio', (d / bound).max())
    if bad.any():
        r, c = np.nonzero(bad)
        print(' rows', np.unique(r)[:10], 'cols', np.unique(c)[:10], 'got', y[r[0], c[0]], 'ref', ref[r[0], c[0]])

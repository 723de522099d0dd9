import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import zl_oracle as oracle
import synth
from zhilight_amd import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(61)
m, k, nff, g = 300, 1024, 200, 128
qw1, qz1, sc1 = synth.gptq_hf(rng, k, nff, g)
qw2, qz2, sc2 = synth.gptq_hf(rng, k, nff, g)
km1, km2 = oracle.gptq_prepare_k_major(qw1, qz1, sc1, g), oracle.gptq_prepare_k_major(qw2, qz2, sc2, g)
cat = [np.concatenate([a, b], axis=0) for a, b in zip(km1, km2)]
t = lambda a, dt=None: (torch.from_numpy(a).to(dev) if dt is None else torch.from_numpy(a).view(dt).to(dev))
w = ops.W4MWeight.from_k_major(t(cat[0].view(np.int32)), t(cat[1]), t(cat[2], torch.float16), g, row_interleave=True)
x = synth.act(rng, m, k)
ge = oracle.gemm_nt(oracle.h2u(x), oracle.gptq_dequant_k_major(*km1), exact=True).astype(np.float16)
ue = oracle.gemm_nt(oracle.h2u(x), oracle.gptq_dequant_k_major(*km2), exact=True).astype(np.float16)
ref = oracle.u2h(oracle.silu_mul(oracle.h2u(ge), oracle.h2u(ue))).astype(np.float64)
for it in range(4):
    got = ops.w4a16_gemm_mfma(t(x), w, epilogue=ops.EPI_SILU_MUL).float().cpu().numpy().astype(np.float64)
    d = np.abs(got - ref)
    bad = d > 2.0 ** -9 * max(1.0, np.abs(ref).max())
    print(it, 'max d', d.max(), 'bad', bad.sum(), 'rows', np.unique(np.nonzero(bad)[0])[:12], 'cols', np.unique(np.nonzero(bad)[1])[:12])
    if bad.any():
        r, c = np.nonzero(bad); print('  got', got[r[0], c[0]], 'ref', ref[r[0], c[0]], 'nan', np.isnan(got).sum())

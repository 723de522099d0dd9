"""How often does a W4A16 kernel's fp16 output differ from the correctly rounded exact product?  (The full-geometry parity tests
measure how a 4-layer network amplifies such one-ulp differences; this measures their RATE per kernel.)
Same operands through w4_slab.hip (default for 5..32 rows) and w4_phase.hip (ZL_W4_SLAB=-1), Llama-3-8B shapes cut to 1024 output
columns, 32 rows; exact = the oracle's fp64 sum rounded once to fp16.
usage: python tools/ubench/tie_rate.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
import oracle.zl_oracle as oracle  # noqa: E402
from zhilight_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
for name, k, n in (("qkv / attn_out (K 4096)", 4096, 1024), ("down (K 14336)", 14336, 1024)):
    qw, qz, sc = synth.gptq_hf(rng, k, n, 128)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, 128)
    w = ops.W4MWeight.from_k_major(torch.from_numpy(km[0].view(np.int32)).to(dev), torch.from_numpy(km[1]).to(dev),
                                   torch.from_numpy(km[2].view(np.float16)).to(dev), 128)
    for m in (8, 32):
        x = synth.act(rng, m, k)
        exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km)
        want = exact.astype(np.float16)
        xt = torch.from_numpy(x).to(dev)
        res = {}
        for label, env in (("slab", None), ("phase", "-1")):
            os.environ.pop("ZL_W4_SLAB", None)
            if env:
                os.environ["ZL_W4_SLAB"] = env
            got = ops.w4a16_gemm_mfma(xt, w).cpu().numpy()
            diff = got.view(np.uint16) != want.view(np.uint16)
            ulp = np.abs(got.view(np.int16).astype(np.int32) - want.view(np.int16).astype(np.int32))
            res[label] = (float(diff.mean()), int(ulp.max()))
        os.environ.pop("ZL_W4_SLAB", None)
        print(f"{name} M={m}: outputs that are not the correctly rounded exact sum -- slab {res['slab'][0] * 100:.4f} % (max {res['slab'][1]} ulp), "
              f"phase {res['phase'][0] * 100:.4f} % (max {res['phase'][1]} ulp)")

"""precision probe of v_mfma_f32_16x16x32_fp8_fp8 through zl_fp8_block_gemm_group: +M*M - M*M + 126 small products in one 128-k block"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import zl_oracle as oracle
from zhilight_amd import ops
dev = torch.device("cuda:0")
enc = lambda v: oracle.f32_to_e4m3(np.array([v], np.float32))[0]
for big in (448.0, 64.0, 8.0, 1.0):
    for small in (0.125, 0.015625, 2.0 ** -9):
        for pos in (0, 40):     # the two big products in the same 32-k MFMA step or in different ones
            a = np.full((16, 128), enc(1.0), np.uint8)
            w = np.full((16, 128), enc(small), np.uint8)
            a[:, 0] = enc(big); w[:, 0] = enc(big)
            a[:, 1 + pos] = enc(-big); w[:, 1 + pos] = enc(big)
            sa = np.ones((1, 16), np.float32); sw = np.ones((1, 1), np.float32)
            got = ops.fp8_block_gemm(torch.from_numpy(a).to(dev), torch.from_numpy(sa).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(sw).to(dev), dtype=torch.float16)
            print(f"big {big:6.1f} small {small:.6f} pos {pos:2d}: got {float(got[0,0]):.6f} exact {126 * small:.6f}")

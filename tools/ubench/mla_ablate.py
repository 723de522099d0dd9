"""ablation of k_mla_decode_wide (zl_debug_mla bits: 1 no P.V, 2 no scores, 4 no staging waits / refills, 8 no records)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zhilight_amd import ops
from zhilight_amd._lib import lib
dev = torch.device("cuda:0")
for b, L in ((1, 8192), (32, 1024), (32, 4096)):
    h = 128
    q = torch.randn(b, h, 576, device=dev).to(torch.bfloat16)
    nset = max(1, min(20, int(300e6 // (b * L * 1152)) + 1))
    sets = []
    for _ in range(nset):
        bufs = [torch.randn(L, 576, device=dev).to(torch.bfloat16) for _ in range(b)]
        sets.append((bufs, torch.tensor([t.data_ptr() for t in bufs], dtype=torch.int64, device=dev)))
    lens = torch.full((b,), L, dtype=torch.int32, device=dev)
    out = []
    for bits in (0, 1, 2, 3, 4, 7, 8, 15):
        lib().zl_debug_mla(bits)
        ops.mla_decode_attention(q, lens, sets[0][1], 0.1, L, algo=2)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(20):
                ops.mla_decode_attention(q, lens, sets[i % nset][1], 0.1, L, algo=2)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        out.append("%d: %.1f" % (bits, e0.elapsed_time(e1) * 1e3 / 20))
    lib().zl_debug_mla(0)
    print(f"batch {b} keys {L}: us per launch pair by ablation bits  " + "  ".join(out), flush=True)
    del sets

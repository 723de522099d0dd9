"""Sweep of w4_slab.hip's geometry (tiles per workgroup R, waves per workgroup NW, 128-k groups per wave GPW -> K split) on the
Llama-3-8B projection shapes, per row count, next to the phase kernel (ZL_W4_SLAB=-1) and the planner's own pick.
usage: python tools/bench_slab.py [--m 32 16 9] [--iters 40] [--layers 6]
Per-launch time with HIP events over hipGraph-captured launches rotating through distinct weight buffers (HBM-cold)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhilight_amd import ops  # noqa: E402


def time_launches(fn, iters):
    fn(0)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(iters):
            fn(i)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[32, 16, 9])
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--shapes", nargs="+", default=["qkv", "o", "gate_up", "down"])
    ap.add_argument("--full", action="store_true", help="every forced geometry as well")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = {"qkv": (6144, 4096, 0), "o": (4096, 4096, 0), "gate_up": (28672, 4096, ops.EPI_SILU_MUL), "down": (4096, 14336, 0)}
    keys = ("ZL_W4_SLAB", "ZL_W4_SLAB_R", "ZL_W4_SLAB_NW", "ZL_W4_SLAB_GPW")
    for name in a.shapes:
        n, k, epi = shapes[name]
        ws = [ops.W4MWeight.random(n, k, 128, dev, row_interleave=bool(epi)) for _ in range(a.layers)]
        for m in a.m:
            x = torch.randn(m, k, dtype=torch.float16, device=dev)
            out = torch.empty(m, n // 2 if epi else n, dtype=torch.float16, device=dev)
            ops.w4_scratch_reserve(dev, 32, 28672)
            rows = []
            cfgs = [("phase", {"ZL_W4_SLAB": "-1"}), ("planned", {}), ("planned, many tiles too (slab = 2)", {"ZL_W4_SLAB": "2"})]
            for r in ((1, 2, 4, 7, 8) if a.full else ()):
                for nw in (4, 8):
                    for gpw in (1, 2, 4):
                        cfgs.append((f"r{r} nw{nw} gpw{gpw}", {"ZL_W4_SLAB_R": str(r), "ZL_W4_SLAB_NW": str(nw), "ZL_W4_SLAB_GPW": str(gpw)}))
            for label, env in cfgs:
                for kk in keys:
                    os.environ.pop(kk, None)
                os.environ.update(env)
                groups, tiles = k // 128, n // 16
                if env.get("ZL_W4_SLAB_R"):
                    r, nw, gpw = int(env["ZL_W4_SLAB_R"]), int(env["ZL_W4_SLAB_NW"]), int(env["ZL_W4_SLAB_GPW"])
                    ks = -(-groups // (nw * gpw))
                    grid = -(-tiles // r) * ks
                    if grid > 1024 or ks > 32 or (nw * gpw - groups >= 2 * gpw and ks == 1):
                        continue
                    label += f" ks{ks} grid{grid}"
                try:
                    us = time_launches(lambda i: ops.w4a16_gemm_mfma(x, ws[i % a.layers], out=out, epilogue=epi), a.iters)
                except Exception as e:  # noqa: BLE001
                    rows.append((1e9, f"{label}: {e}"))
                    continue
                rows.append((us, label))
            for kk in keys:
                os.environ.pop(kk, None)
            base = [u for u, l in rows if l == "phase"][0]
            print(f"== {name} N={n} K={k} M={m}: phase {base:.2f} us, planned {[u for u, l in rows if l == 'planned'][0]:.2f} us")
            for us, label in sorted(rows)[:8]:
                print(f"   {us:8.2f} us  {label}")
        del ws


if __name__ == "__main__":
    main()

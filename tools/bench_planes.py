"""Decode batches of 5..32 rows: the fp16-dequant phase kernel against the digit-plane route (zl_w4a16_planes + zl_w4a16_gemm_planes)
on the four projections of a Llama-3-8B layer, 32 launches on 32 layers' weights (HBM-cold), graph replay, HIP events.
    python tools/bench_planes.py [--rows 8,16,32] [--reps 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(graph, reps):
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def capture(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="8,16,32")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--layers", type=int, default=32)
    a = ap.parse_args()
    from zhilight_amd import ops
    from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig
    dev = torch.device("cuda:0")
    cfg = ModelConfig.llama3_8b()
    cfg.num_layers = a.layers
    model = LLaMA(cfg, QuantConfig(), device=dev)
    model.init_random(seed=0)
    L = model.layers
    nl = len(L)
    for m in [int(v) for v in a.rows.split(",")]:
        x4k = torch.randn(m, cfg.dim_model, device=dev).half()
        xff = torch.randn(m, cfg.dim_ff, device=dev).half()
        hid = torch.randn(m, cfg.dim_model, device=dev).half()
        out_qkv = torch.empty(m, 6144, dtype=torch.float16, device=dev)
        out_act = torch.empty(m, cfg.dim_ff, dtype=torch.float16, device=dev)
        xn = torch.empty_like(x4k)
        pl4k = ops.w4_planes(x4k)
        plff = ops.w4_planes(xff)
        fused_norm = m <= 8

        def ref(kind):
            def f():
                for l in L:
                    if kind == "qkv":
                        if fused_norm:
                            ops.w4_linear(x4k, l.qkv.weight, out=out_qkv, norm_weight=l.ln_attn, norm_eps=1e-5)
                        else:
                            ops.w4_linear(ops.rmsnorm(x4k, l.ln_attn, 1e-5, out=xn), l.qkv.weight, out=out_qkv)
                    elif kind == "o":
                        ops.w4_linear(x4k, l.attn_out.weight, out=hid, residual=hid, epilogue=ops.EPI_RESIDUAL)
                    elif kind == "gateup":
                        if fused_norm:
                            ops.w4_linear(x4k, l.w_in_gated.weight, out=out_act, norm_weight=l.ln_ff, norm_eps=1e-5, epilogue=ops.EPI_SILU_MUL)
                        else:
                            ops.w4_linear(ops.rmsnorm(x4k, l.ln_ff, 1e-5, out=xn), l.w_in_gated.weight, out=out_act, epilogue=ops.EPI_SILU_MUL)
                    else:
                        ops.w4_linear(xff, l.w_out.weight, out=hid, residual=hid, epilogue=ops.EPI_RESIDUAL)
            return f

        def planes(kind, what):
            def f():
                for l in L:
                    if kind == "qkv":
                        if what != "gemv":
                            ops.w4_planes(x4k, norm_weight=l.ln_attn, norm_eps=1e-5, out=pl4k)
                        if what != "conv":
                            ops.w4_linear_planes(pl4k, m, l.qkv.weight, out=out_qkv)
                    elif kind == "o":
                        if what != "gemv":
                            ops.w4_planes(x4k, out=pl4k)
                        if what != "conv":
                            ops.w4_linear_planes(pl4k, m, l.attn_out.weight, out=hid, residual=hid, epilogue=ops.EPI_RESIDUAL)
                    elif kind == "gateup":
                        if what != "gemv":
                            ops.w4_planes(x4k, norm_weight=l.ln_ff, norm_eps=1e-5, out=pl4k)
                        if what != "conv":
                            ops.w4_linear_planes(pl4k, m, l.w_in_gated.weight, out=out_act, epilogue=ops.EPI_SILU_MUL)
                    else:
                        if what != "gemv":
                            ops.w4_planes(xff, out=plff)
                        if what != "conv":
                            ops.w4_linear_planes(plff, m, l.w_out.weight, out=hid, residual=hid, epilogue=ops.EPI_RESIDUAL)
            return f
        row = {"rows": m}
        for kind in ("qkv", "o", "gateup", "down"):
            row[kind] = {"fp16_route_us": round(timed(capture(ref(kind)), a.reps) / nl, 2),
                         "planes_conv_us": round(timed(capture(planes(kind, "conv")), a.reps) / nl, 2),
                         "planes_gemv_us": round(timed(capture(planes(kind, "gemv")), a.reps) / nl, 2),
                         "planes_both_us": round(timed(capture(planes(kind, "both")), a.reps) / nl, 2)}
        row["layer_fp16_us"] = round(sum(row[k]["fp16_route_us"] for k in ("qkv", "o", "gateup", "down")), 2)
        row["layer_planes_us"] = round(sum(row[k]["planes_both_us"] for k in ("qkv", "o", "gateup", "down")), 2)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

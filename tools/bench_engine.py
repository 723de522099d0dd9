"""A/B of the batch-1 decode step's launch variants on the bench model (Llama-3-8B GPTQ-Int4, 32 layers, HBM-cold weights):
the integer-plane GEMV with its register ring (k_w4a16_i8p), the loader / consumer engine (w4_engine.hip), and the fused
attn_out -> gate|up launch.  Every variant: (a) the whole greedy step as one hipGraph, (b) the GEMV launches only
(LLaMA.encode(gemv_only=True)), (c) each of the four projections alone, 32 launches on the 32 layers' weights in model order.
HIP events around graph replays on the launch stream.  One JSON line per variant + a table.

    python tools/bench_engine.py [--layers 32] [--reps 20] [--variants i8p,engine,...] [--ring 0]"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VARIANTS = {
    "i8p": {},
    "engine": {"ZL_W4_SMALL_ALGO": "2"},
    "i8p+fused": {"ZL_FUSE_O_GATEUP": "1"},
    "engine+fused": {"ZL_W4_SMALL_ALGO": "2", "ZL_FUSE_O_GATEUP": "1"},
}
KEYS = ("ZL_W4_SMALL_ALGO", "ZL_FUSE_O_GATEUP", "ZL_W4_PHASE_ROUNDS")


def timed(graph, reps):
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def capture(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--variants", default="i8p,engine,i8p+fused,engine+fused")
    ap.add_argument("--ring", type=int, default=0, help="cap the engine's ring at this many rounds")
    ap.add_argument("--no-ops", action="store_true")
    ap.add_argument("--no-step", action="store_true", help="skip the whole-step and gemv-only graphs (per-projection numbers only)")
    a = ap.parse_args()
    from zhilight_amd import ops
    from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig
    dev = torch.device("cuda:0")
    from zhilight_amd._lib import lib
    lib().zl_debug_engine_knobs(a.ring)
    cfg = ModelConfig.llama3_8b()
    cfg.num_layers = a.layers
    model = LLaMA(cfg, QuantConfig(), device=dev)
    model.init_random(seed=0)
    len_buf = (a.seq + 4 * a.reps + 64 + 63) // 64 * 64
    rows = []
    ref_hidden = None
    for name in a.variants.split(","):
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(VARIANTS[name])
        torch.manual_seed(5)
        ctx = model.new_context(1, len_buf, a.seq, fill_random=True)
        ctx.tokens.fill_(11)
        # one eager step from a fixed state: every variant must leave the same hidden row
        model.encode(ctx)
        h = model.last_hidden.clone()
        if ref_hidden is None:
            ref_hidden = h
        same = bool(torch.equal(h, ref_hidden))
        if a.no_step:
            g_step = g_gemv = None
            t_step = t_gemv = float("nan")
        else:
            g_step = capture(lambda: model.step_greedy(ctx))
            t_step = timed(g_step, a.reps)
            bufs = model._buffers(1)
            bufs["hidden"].normal_()
            g_gemv = capture(lambda: model.encode(ctx, gemv_only=True))
            t_gemv = timed(g_gemv, a.reps)
        row = {"variant": name, "ring_cap": a.ring, "ms_per_step": round(t_step * 1e3, 4), "tokens_per_s": round(1.0 / t_step, 1),
               "us_per_layer_step": round((t_step * 1e6 - 166.0) / a.layers, 2),
               "gemv_only_us_per_layer": round(t_gemv * 1e6 / a.layers, 2), "hidden_equals_i8p": same,
               "engine_err": int(ops.engine_state(dev)["err"].item())}
        if not a.no_ops and "fused" not in name:
            x4k = torch.randn(1, cfg.dim_model, device=dev).half()
            xff = torch.randn(1, cfg.dim_ff, device=dev).half()
            hid = torch.randn(1, cfg.dim_model, device=dev).half()
            out_qkv = torch.empty(1, 6144, dtype=torch.float16, device=dev)
            out_act = torch.empty(1, cfg.dim_ff, dtype=torch.float16, device=dev)
            L = model.layers

            def op_qkv():
                for l in L:
                    ops.w4_linear(x4k, l.qkv.weight, out=out_qkv, norm_weight=l.ln_attn, norm_eps=1e-5)

            def op_o():
                for l in L:
                    ops.w4_linear(x4k, l.attn_out.weight, out=hid, residual=hid, epilogue=ops.EPI_RESIDUAL)

            def op_gu():
                for l in L:
                    ops.w4_linear(x4k, l.w_in_gated.weight, out=out_act, norm_weight=l.ln_ff, norm_eps=1e-5, epilogue=ops.EPI_SILU_MUL)

            def op_down():
                for l in L:
                    ops.w4_linear(xff, l.w_out.weight, out=hid, residual=hid, epilogue=ops.EPI_RESIDUAL)
            for nm, fn in (("qkv_norm", op_qkv), ("o_res", op_o), ("gateup_norm_silu", op_gu), ("down_res", op_down)):
                row["us_" + nm] = round(timed(capture(fn), a.reps) * 1e6 / a.layers, 2)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del g_step, g_gemv, ctx
        torch.cuda.empty_cache()
    for k in KEYS:
        os.environ.pop(k, None)
    cols = sorted({k for r in rows for k in r if k != "variant"})
    print("\n%-14s " % "variant" + " ".join("%22s" % c for c in cols))
    for r in rows:
        print("%-14s " % r["variant"] + " ".join("%22s" % r.get(c, "") for c in cols))


if __name__ == "__main__":
    main()

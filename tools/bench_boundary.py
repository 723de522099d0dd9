"""The decode step as the BOUNDARY offers it (VERDICT r04 missing 3 / item 5): the reference's own model::LLaMA -- src/model/llama.cpp,
block.cpp, attention.cpp, feedforward.cpp, linear.cpp compiled unmodified (zhilight_amd/_ref/zl_reflinear*.so, built where the
reference tree exists; this script never reads it) -- running LLaMA::encode (:75-151) + get_logits (:159-165) on hostcpp/nn_amd.cpp
and the C ABI: one launch and one pooled ctx.tensor per reference op, no fused front ends, 32 full Llama-3-8B layers, batch 1,
1024 keys of history, the SAME synthetic GPTQ checkpoint bench.py times (LLaMA.init_synthetic, seed 1234).  Prints ONE JSON line:
tokens/s eager and under hipGraph replay, next to the Python driver's (zhilight_amd/llama.py: fused launches under hipGraph) on
the same weights in the same process, and the distance between the two paths' logits.
usage: python tools/bench_boundary.py [--layers L] [--iters N] [--batch B]      (env CPM_FUSE_QKV / CPM_FUSE_FF_IN / ROPE_CACHE: the reference's
own switches, read by its code)"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1, help="decode tasks in the step (every task: --seq keys of history)")
    args = ap.parse_args()
    from zhilight_amd import _lib, build
    from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig
    _lib.lib()
    path = build.refcompile_target()
    if not os.path.exists(path):
        print(json.dumps({"error": "zl_reflinear was not built (no reference tree at build time)"}))
        return
    sys.path.insert(0, os.path.dirname(path))
    import zl_reflinear as ref
    dev = torch.device("cuda:0")
    cfg = ModelConfig.llama3_8b()
    cfg.num_layers = args.layers
    rsd = {}

    def sink(sd):
        for k, v in sd.items():
            rsd["m." + k[len("llama."):]] = np.ascontiguousarray(v.detach().cpu().numpy())
    model = LLaMA(cfg, QuantConfig(5, 128), dev).init_synthetic(seed=1234, sink=sink)
    rs = cfg.rope_scaling or {}
    if (rs.get("rope_type", rs.get("type")) or "default") not in ("default",):
        # the harness builds the reference's ModelConfig with plain rotary frequencies; the Python driver must rotate the same way
        cfg.rope_scaling = None
    seq = args.seq
    len_buf = (seq + 1 + 63) // 64 * 64
    rm = ref.RefLLaMA(cfg.num_layers, cfg.dim_model, cfg.num_heads, cfg.num_kv_heads, cfg.dim_head, cfg.dim_ff, cfg.vocab_size, eps=cfg.eps,
                      rope_theta=cfg.rope_theta, quant_type=5, group_size=128)
    ref.weight_cache_clear()
    rm.load(rsd, "m")
    del rsd
    torch.manual_seed(7)
    nb = args.batch
    ctx = model.new_context(nb, len_buf, seq, fill_random=True)
    for b in range(nb):
        k = torch.stack([ctx.kv[b][l, 0, :seq] for l in range(cfg.num_layers)]).cpu().numpy()
        v = torch.stack([ctx.kv[b][l, 1, :seq] for l in range(cfg.num_layers)]).cpu().numpy()
        rm.set_history(b, len_buf, np.ascontiguousarray(k), np.ascontiguousarray(v))
    tokens = (17 + 3 * np.arange(nb)).astype(np.int32)
    ctx.tokens.copy_(torch.from_numpy(tokens))
    pos = np.full(nb, seq, np.int32)
    mask = np.tile((np.arange(len_buf) <= seq).astype(np.int8), nb)
    # parity of the two paths on the same weights and history (before anything is timed)
    got_ref = rm.decode_step(tokens, pos, mask).astype(np.float64)
    got_py = model.encode(ctx).float().cpu().numpy().astype(np.float64)
    scale = np.abs(got_py).max()
    dist = float(np.abs(got_ref - got_py).max() / scale)
    t = rm.time_decode_steps(tokens, pos, mask, warmup=3, iters=args.iters, graph=True)
    # the Python driver, same process, same weights: hipGraph replay of step_greedy
    model.step_greedy(ctx)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        model.step_greedy(ctx)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    py_ms = e0.elapsed_time(e1) / args.iters
    out = {"what": "reference model::LLaMA::encode + get_logits on the boundary (hostcpp/nn_amd.cpp over the C ABI), %d layers, batch %d, %d keys; "
                   "one launch + one pooled ctx.tensor per reference op" % (cfg.num_layers, nb, seq),
           "switches": {k: os.environ.get(k) for k in ("CPM_FUSE_QKV", "CPM_FUSE_FF_IN", "ROPE_CACHE")},
           "boundary_path_tokens_per_s": round(nb * 1e3 / t["eager_ms"], 1), "boundary_path_ms_per_step": round(t["eager_ms"], 4),
           "boundary_path_graph_tokens_per_s": round(nb * 1e3 / t["graph_ms"], 1) if "graph_ms" in t else None,
           "boundary_path_graph_ms_per_step": round(t["graph_ms"], 4) if "graph_ms" in t else None,
           "graph_error": t.get("graph_error"),
           "python_driver_tokens_per_s": round(nb * 1e3 / py_ms, 1), "python_driver_ms_per_step": round(py_ms, 4),
           "ratio_boundary_over_driver": round(py_ms / (t.get("graph_ms") or t["eager_ms"]), 3),
           "logits_boundary_vs_driver_max_over_max": round(dist, 6)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- decode throughput of the MI355X-native ZhiLight hot path.

Workload (BASELINE.json configs[1]): Llama-3-8B GPTQ-Int4 (group 128), TP=1, batch-1 greedy decode at
KV length ~1024, synthetic random-init weights of the real shapes and a random KV history
(no checkpoints / datasets in this environment).  A "step" is ONE decode step of the whole model:
embedding -> 32 x (fused RMSNorm+qkv W4A16 GEMV, RoPE + KV scatter, GQA decode attention on the matrix cores,
o_proj GEMV + residual, fused RMSNorm + gate|up GEMV + silu*mul, down GEMV + residual) ->
fused final norm + fp16 lm_head GEMV -> argmax -> position / KV bookkeeping, captured in one hipGraph.

    python bench.py --gpus N --steps K --warmup W [--batch B]

N > 1 (launched with torch.distributed.run, one rank per GPU): the decode path shards as independent
TP=1 replicas (one model + its own requests per GPU, no data-path collective), i.e. weak scaling;
`value` is the whole-job aggregate tokens/s.  Rank 0 prints ONE JSON line with, besides the contract
fields, `roofline` (the W4A16 GEMV kernel: algorithmic bytes / HIP-event time, vs 8 TB/s HBM) and
`cpu_baseline` (the CPU oracle timed on a bounded sample on this box's host cores).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def alg_bytes_w4(n, k, g, m):
    """canonical GPTQ bytes (int4 + fp16 scale + 4-bit zero per group) + activations."""
    return k * n * (0.5 + 2.0 / g + 0.5 / g) + m * k * 2 + m * n * 2


def cpu_baseline(cfg, batch, seq):
    """The reference has no CPU path for the model; this times OUR CPU restatement (oracle/, kind
    "port", R flavour, OpenMP over all host cores) on a bounded sample of one decode step:
    the four W4A16 linears of ONE of the 32 layers + 1/16 of the lm_head rows + one layer's decode
    attention, extrapolated to a full step."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import zl_oracle as zo
    rng = np.random.default_rng(0)
    g = 128
    hd, kvd = cfg.num_heads * cfg.dim_head, cfg.num_kv_heads * cfg.dim_head
    shapes = [(hd + 2 * kvd, cfg.dim_model), (cfg.dim_model, hd), (2 * cfg.dim_ff, cfg.dim_model), (cfg.dim_model, cfg.dim_ff)]
    t_lin = 0.0
    for n, k in shapes:
        qw = rng.integers(0, 2 ** 32, size=(n, k // 8), dtype=np.uint64).astype(np.uint32)
        qz = rng.integers(0, 16, size=(n, k // g), dtype=np.uint8)
        sc = (np.abs(rng.standard_normal((n, k // g))) * 0.0025 + 1e-4).astype(np.float16).view(np.uint16)
        x = rng.standard_normal((batch, k)).astype(np.float16).view(np.uint16)
        t0 = time.perf_counter()
        zo.gptq_gemm_k_major(x, qw, qz, sc)
        t_lin += time.perf_counter() - t0
    rows = cfg.vocab_size // 16
    w = (rng.standard_normal((rows, cfg.dim_model)) * 0.02).astype(np.float16).view(np.uint16)
    x = rng.standard_normal((batch, cfg.dim_model)).astype(np.float16).view(np.uint16)
    t0 = time.perf_counter()
    zo.gemm_nt(x, w)
    t_head = (time.perf_counter() - t0) * 16
    kb = [rng.standard_normal((seq, cfg.num_kv_heads, cfg.dim_head)).astype(np.float16).view(np.uint16) for _ in range(batch)]
    q = rng.standard_normal((batch, 1, cfg.num_heads, cfg.dim_head)).astype(np.float16).view(np.uint16)
    mask = np.ones(batch * seq, np.int8)
    t0 = time.perf_counter()
    zo.mqa_rag_buffer(q, np.full(batch, seq, np.int32), kb, kb, mask, cfg.num_kv_heads, 0.088)
    t_attn = time.perf_counter() - t0
    t_step = cfg.num_layers * (t_lin + t_attn) + t_head
    return {
        "value": batch / t_step, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
        "extrapolated": "one of %d layers + 1/16 of the lm_head rows timed, scaled to a full step" % cfg.num_layers,
        "sample": (f"oracle/ R-flavour CPU port (the reference has no CPU path): 1 of {cfg.num_layers} layers' W4A16 linears "
                   f"({t_lin:.2f} s) + decode attention ({t_attn:.2f} s) + 1/16 of lm_head rows, extrapolated to a full step"),
        "note": "a bit-faithful EMULATION of the reference kernels' fp16 arithmetic (soft-float rounding points, their reduction order), "
                "not an optimised CPU GEMV: a reported baseline, never a ratio to quote",
    }


def roofline_int8(model, cfg, batch, dev):
    """dominant kernel of the int8 route = k_w8a8_phase (streaming W8A8 GEMM with the scale-back fused: qkv, attn_out,
    w_in|w_gated, w_out = 4 launches per layer, 1 B per weight); beyond 32 rows the tiled GEMM runs instead"""
    from zhilight_amd import ops
    roof = None
    # dominant kernel of the int8 route = k_w8a8_phase (streaming W8A8 GEMM with the scale-back fused: qkv, attn_out,
    # w_in|w_gated, w_out = 4 launches per layer, 1 B per weight); beyond 32 rows the tiled GEMM runs instead
    lays = model.layers
    stream = batch <= 32
    xq = torch.randint(-127, 128, (batch, cfg.dim_ff), dtype=torch.int8, device=dev)
    sxv = torch.full((batch,), 0.02, dtype=torch.float32, device=dev)
    hid = torch.zeros(batch, cfg.dim_model, dtype=torch.float16, device=dev)
    if stream:
        launches = []
        for lay in lays:
            launches += [(lay.qkv.stream_weight(), ops.W8_BACK, None), (lay.attn_out.stream_weight(), ops.W8_BACK_ADD, hid),
                         (lay._gated_stream_weight(), ops.W8_ACT_SILU, None), (lay.w_out.stream_weight(), ops.W8_BACK_ADD, hid)]
        xq_by_k = {k: xq[:, :k].contiguous() for k in {w.k for w, _, _ in launches}}
        outs = {(w.n, e): torch.empty(batch, w.n // 2 if e == ops.W8_ACT_SILU else w.n, dtype=torch.float16, device=dev)
                for w, e, _ in launches}

        def gemms():
            for w, e, add in launches:
                ops.w8a8_gemm_phase(xq_by_k[w.k], sxv, w, e, addend=add, out=outs[(w.n, e)])
        per_launch = sum(w.n * w.k + batch * (w.k + 2 * w.n) for w, _, _ in launches) / len(launches)
        kdesc = "k_w8a8_phase (W8A8 streaming GEMM + fused scale-back, 4 launches/layer)"
    else:
        launches = [(lin, lin.dim_in) for lay in lays for lin in lay.linears()]
        xq_by_k = {k: xq[:, :k].contiguous() for k in {l.dim_in for l, _ in launches}}

        def gemms():
            for lin, k in launches:
                lin.gemm(xq_by_k[k])
        per_launch = sum(l.dim_in * l.dim_out + batch * (l.dim_in + 4 * l.dim_out) for l, _ in launches) / len(launches)
        kdesc = "k_int8_gemm_tiled (int8 x int8 -> int32, 5 launches/layer)"
    gemms()
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        gemms()
    g2.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g2.replay()
    e1.record()
    torch.cuda.synchronize()
    t_launch = e0.elapsed_time(e1) * 1e-3 / (10 * len(launches))
    achieved = per_launch / t_launch / 1e9
    roof = {"bound": "hbm", "kernel": kdesc, "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "bytes_per_launch": int(per_launch), "us_per_launch": round(t_launch * 1e6, 3),
            "note": "avg over the int8 GEMM launches of one step incl. inter-kernel gaps (graph replay, HIP events)"}
    return roof


def roofline_w4(model, cfg, batch, dev, ctx, layers_override=False):
    """The W4A16 GEMV launches of one step -- 4 x 32, IN THE VARIANTS THE STEP ITSELF LAUNCHES (fused RMSNorm + rotary + KV
    scatter on qkv, split merge + residual on attn_out, norm + silu*mul on gate|up, residual on down: LLaMA.encode(...,
    gemv_only=True)), model order, on the real HBM-cold weights (3.6 GB >> 256 MB Infinity Cache), captured without the
    attention / embedding / lm_head kernels; HIP events on the launch stream."""
    from zhilight_amd import ops
    bufs = model._buffers(batch)
    lins = [lin for layer in model.layers for lin in (layer.qkv, layer.attn_out, layer.w_in_gated, layer.w_out)]
    bufs["hidden"].normal_()
    bufs["attn"].normal_()
    in_step = False
    if batch <= 8:                                     # beyond 8 rows the two RMSNorms are launches of their own: plain projections
        try:
            model.encode(ctx, gemv_only=True)
            in_step = True
        except ops.ZLError:
            in_step = False                            # a route without the fused decode launches

    def gemvs():
        if in_step:
            model.encode(ctx, gemv_only=True)
            return
        for layer in model.layers:
            ops.w4_linear(bufs["hidden"], layer.qkv.weight, out=bufs["qkv"])
            ops.w4_linear(bufs["attn"], layer.attn_out.weight, out=bufs["hidden"], residual=bufs["hidden"], epilogue=ops.EPI_RESIDUAL)
            ops.w4_linear(bufs["hidden"], layer.w_in_gated.weight, out=bufs["act"], epilogue=ops.EPI_SILU_MUL)
            ops.w4_linear(bufs["act"], layer.w_out.weight, out=bufs["hidden"], residual=bufs["hidden"], epilogue=ops.EPI_RESIDUAL)
    gemvs()
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        gemvs()
    g2.replay()
    torch.cuda.synchronize()
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g2.replay()
    e1.record()
    torch.cuda.synchronize()
    nl = len(lins) + (len(model.layers) if in_step and batch > 8 else 0)   # beyond 8 rows the qkv norm is its own (counted) launch
    t_gemv_only = e0.elapsed_time(e1) * 1e-3 / (reps * len(lins))
    # IN-STEP time of the same launches (what the 0.70 target is about): the whole greedy step minus the step with exactly
    # these launches left out (LLaMA.encode(skip_gemv=True): embedding, rope table, attention, lm_head, greedy bookkeeping), both
    # as hipGraph replays on a fresh context, HIP events.  Falls back to the GEMV-only graph where that mode does not apply.
    t_launch, how = t_gemv_only, "gemv-only graph"
    if in_step and batch <= 4 and not model.tp:
        try:
            len_buf = int(ctx.max_len_buf)
            pos0 = int(ctx.positions[0].item()) if ctx.positions.numel() else 0
            pos0 = min(pos0, len_buf - 40)

            def graph_of(skip):
                c2 = model.new_context(batch, len_buf, pos0, fill_random=True)
                c2.tokens.copy_(ctx.tokens)
                model.step_greedy(c2, skip_gemv=skip)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    model.step_greedy(c2, skip_gemv=skip)
                g.replay()
                torch.cuda.synchronize()
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    g.replay()
                b_.record()
                torch.cuda.synchronize()
                t = a.elapsed_time(b_) * 1e-3 / reps
                del g, c2
                return t
            t_full, t_rest = graph_of(False), graph_of(True)
            if t_full > t_rest > 0:
                t_launch, how = (t_full - t_rest) / len(lins), "step graph minus the step without these launches"
            torch.cuda.empty_cache()
        except Exception as e:                                  # noqa: BLE001 (the roofline leg must never cost the headline)
            sys.stderr.write("bench: in-step GEMV timing unavailable (%s)\n" % str(e).splitlines()[0])
    tot_bytes = sum(alg_bytes_w4(lin.weight.n, lin.weight.k, lin.weight.group_size, batch) for lin in lins)
    per_launch = tot_bytes / len(lins)
    achieved = per_launch / t_launch / 1e9
    mfma = isinstance(model.layers[0].qkv.weight, ops.W4MWeight)
    small = os.environ.get("ZL_W4_SMALL_ALGO", "0") != "1"
    # 1..2 rows: k_w4a16_i8p (integer planes) for all four; 3..4: the same except the long-K down projection (k_w4a16_slab);
    # 5..32: k_w4a16_slab (2-D K-split tiles, round 6) for all four
    kname = ("k_w4a16_i8p" if batch <= 2 and small else "k_w4a16_i8p+k_w4a16_slab" if batch <= 4 and small else "k_w4a16_slab") if mfma else "k_w4a16_gemm"
    # HBM traffic per launch: measured off-line with rocprofv3 --pmc (a counter pass cannot run inside
    # this process); the committed summary is per kernel flavour and for these four shapes only
    traffic = None
    try:
        import glob
        tfile = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_gemv_traffic.json")))[-1]
        with open(tfile) as fh:
            tj = json.load(fh)
        if tj.get("kernel", "").startswith(kname.split("+")[0]) and not layers_override and batch == 1:
            traffic = int(tj["avg_bytes_per_launch"])
    except (OSError, ValueError, KeyError, IndexError):
        traffic = None
    roof = {"bound": "hbm", "kernel": kname + " (W4A16 GEMV, 4 launches/layer, the step's own fused variants)" if in_step
            else kname + " (W4A16 GEMV, 4 launches/layer, plain variants)", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "bytes_per_launch": int(per_launch), "us_per_launch": round(t_launch * 1e6, 3), "launches_timed": nl,
            "timing": how, "us_per_launch_gemv_only_graph": round(t_gemv_only * 1e6, 3),
            "traffic_source": "offline rocprofv3 --pmc pass (profiles/r*_gemv_traffic.json), not measured in this run" if traffic is not None else None,
            "note": "avg over the %d GEMV launches of one step incl. the kernel boundaries between them (graph replay, HIP events); "
                    "in-step rocprofv3 averages of the same kernels: profiles/r06_bench_kernel_stats.csv (batch 8 / 32: r06_bench_b8 / b32_kernel_stats.csv)" % len(lins)}
    return roof


def step_bytes_of(model, cfg, batch, seq, int8):
    """whole-step algorithmic bytes: quantised linears + fp16 lm_head + KV read"""
    if int8:
        lin_bytes = sum(l.dim_in * l.dim_out + 2 * l.dim_out for lay in model.layers for l in lay.linears())
    else:
        lin_bytes = sum(alg_bytes_w4(l.weight.n, l.weight.k, 128, batch) for lay in model.layers
                        for l in (lay.qkv, lay.attn_out, lay.w_in_gated, lay.w_out))
    return lin_bytes + cfg.vocab_size * cfg.dim_model * 2 + batch * cfg.num_layers * 2 * cfg.num_kv_heads * seq * cfg.dim_head * 2


def extra_decode_run(model, cfg, batch, seq, steps, warmup, dev, int8):
    """north_star asks for batch 1 / 8 / 32 (and BASELINE configs[2] is the INT8 route at batch 32): the same timed
    region as the headline (hipGraph replay of the whole step, synchronise on both sides), one GPU, reported as extra
    keys of the one JSON line."""
    len_buf = (seq + warmup + steps + 4 + 63) // 64 * 64
    ctx = model.new_context(batch, len_buf, seq, fill_random=True)
    ctx.tokens.copy_(torch.randint(0, cfg.vocab_size, (batch,), device=dev, dtype=torch.int32))
    model.step_greedy(ctx)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        model.step_greedy(ctx)
    for _ in range(warmup):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    roof = roofline_int8(model, cfg, batch, dev) if int8 else roofline_w4(model, cfg, batch, dev, ctx)
    del graph, ctx
    torch.cuda.empty_cache()
    return {"workload": "Llama-3-8B %s TP=1 batch=%d decode seq=%d" % ("INT8 (AutoInt8 linears)" if int8 else "GPTQ-Int4 g128", batch, seq),
            "batch": batch, "value": round(batch * steps / elapsed, 2), "unit": "tokens/s", "ms_per_step": round(elapsed / steps * 1e3, 4),
            "steps": steps, "warmup": warmup,
            "step_hbm_roofline_frac": round(step_bytes_of(model, cfg, batch, seq, int8) / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": roof}


def config5_leg(timeout_s=180):
    """row f4 numbers (MLA decode attention, FP8 block GEMM) from tools/bench_config5.py in a CHILD process: whatever happens there
    -- an exception, a fault, a hang -- costs this field only, never the headline line"""
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "bench_config5.py")
    try:
        r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "exit %d: %s" % (r.returncode, (r.stderr or r.stdout).strip().splitlines()[-1][:200] if (r.stderr or r.stdout).strip() else "")}
        return json.loads(lines[-1])
    except Exception as e:                                  # noqa: BLE001 (timeouts, a missing interpreter, bad JSON: report, do not raise)
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def boundary_leg(timeout_s=240):
    """the decode step as the drop-in BOUNDARY offers it -- the reference's own model::LLaMA (compiled unmodified,
    zhilight_amd/_ref) running LLaMA::encode + get_logits on hostcpp/nn_amd.cpp over the C ABI, one launch per reference op, same
    synthetic checkpoint, 32 layers, batch 1 -- timed by tools/bench_boundary.py in a CHILD process under the switches a deployment
    sets (CPM_FUSE_QKV=1, CPM_FUSE_FF_IN=1, ROPE_CACHE=1: the reference reads them itself), eager and under hipGraph replay, next to
    the Python driver's fused step on the same weights in the same process.  Whatever happens there costs this field only."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "bench_boundary.py")
    env = dict(os.environ, CPM_FUSE_QKV="1", CPM_FUSE_FF_IN="1", ROPE_CACHE="1")
    try:
        r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=timeout_s, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "exit %d: %s" % (r.returncode, (r.stderr or r.stdout).strip().splitlines()[-1][:200] if (r.stderr or r.stdout).strip() else "")}
        return json.loads(lines[-1])
    except Exception as e:                                  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def check_against_oracle(model, batch, dev):
    """Before anything is timed: the weights bench.py generates (directly in the packed ZLW4M layout, never seen by a
    test) are unpacked again (zl_w4m_unpack) and the four W4A16 linears of layer 0 and of the last layer, run through the
    HIP path exactly as the step runs them, must reproduce the CPU oracle's exact product of the unpacked operands.
    Returns the worst max|err| / max|ref| (raises above 1e-3, north_star's bar)."""
    import numpy as np
    from zhilight_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import zl_oracle as zo
    worst = 0.0
    gen = torch.Generator(device=dev).manual_seed(99)
    for layer in (model.layers[0], model.layers[-1]):
        for lin in (layer.qkv, layer.attn_out, layer.w_in_gated, layer.w_out):
            w = lin.weight
            if not isinstance(w, ops.W4MWeight):
                return None                             # ZL_W4_ALGO=exact: that kernel is bit-tested against R elsewhere
            x = torch.randn(batch, w.k, device=dev, generator=gen).to(torch.float16)
            y = ops.w4_linear(x, w).float().cpu().numpy().astype(np.float64)
            if w.row_interleave:
                y = np.concatenate([y[:, 0::2], y[:, 1::2]], axis=1)
            qw, qz, sc = (t.cpu().numpy() for t in w.to_k_major())
            ref = zo.gptq_gemm_k_major_exact(x.cpu().numpy().view(np.uint16), qw.view(np.uint32), qz, sc.view(np.uint16))
            err = float(np.abs(y - ref).max() / np.abs(ref).max())
            worst = max(worst, err)
            if not err <= 1e-3:
                raise SystemExit("bench: HIP W4A16 linear %s deviates from the CPU oracle: %.3g" % (lin.name, err))
    return worst


def check_logits_against_oracle(model, ctx, dev):
    """a LOGIT of the timed model, not isolated linears.  Before anything is timed the step bench.py is about to
    time -- all 32 layers, the synthetic 1 024-key history, the full lm_head -- runs once eagerly, and the CPU oracle evaluates the
    SAME network: every packed weight unpacked again (zl_w4m_unpack: fused q|k|v rows split, gate / up de-interleaved), the KV
    buffers and norm weights copied from the device, exact (fp64) linears rounded once to fp16 ("E", the flavour the 1e-3 bar of
    tests/test_gpu_fullgeom.py is held against).  Reports max|logit - ref| / max|ref|; never raises (a failure here must not cost
    the bench line: it is reported as {"error": ...})."""
    import time
    import numpy as np
    try:
        t0 = time.time()
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import zl_oracle as zo
        from test_gpu_model import OracleModel
        from zhilight_amd import ops
        cfg = model.cfg
        b = ctx.tokens.numel()
        hd, kvd = cfg.num_heads * cfg.dim_head, cfg.num_kv_heads * cfg.dim_head
        u16 = lambda t: t.detach().to(torch.float16).cpu().numpy().view(np.uint16)
        om = OracleModel.__new__(OracleModel)
        om.o, om.cfg, om.g, om.kv_quant, om.exact_attention = zo, cfg, 128, False, False
        rs = cfg.rope_scaling
        om.rope_kind = (rs.get("rope_type", rs.get("type")) or "plain") if rs else "plain"
        if om.rope_kind == "default":
            om.rope_kind = "plain"
        tokens = ctx.tokens.cpu().numpy().astype(np.int32)
        pos = [int(p) for p in ctx.positions.cpu().numpy()]
        # the embedding rows of the step's tokens as a table of their own (token i -> row i), norms and lm_head as they are
        om.sd = {"model.embed_tokens.weight": u16(model.token_embedding[torch.from_numpy(tokens).long().to(dev)]).view(np.float16),
                 "model.norm.weight": u16(model.output_layernorm).view(np.float16), "lm_head.weight": u16(model.lm_head).view(np.float16)}
        om.km = {}

        def unpack(w):
            qw, qz, sc = (t.cpu().numpy() for t in w.to_k_major())          # checkpoint row order (a row-interleaved weight is undone)
            return qw.view(np.uint32), qz, sc.view(np.uint16)

        def rows(km, sl):
            return tuple(np.ascontiguousarray(a[sl]) for a in km)
        ff = cfg.dim_ff
        for i, layer in enumerate(model.layers):
            for lin in (layer.qkv, layer.attn_out, layer.w_in_gated, layer.w_out):
                if not isinstance(lin.weight, ops.W4MWeight) or lin.perm is not None or lin.bias is not None:
                    return None
            pfx = "model.layers.%d." % i
            om.sd[pfx + "input_layernorm.weight"] = u16(layer.ln_attn).view(np.float16)
            om.sd[pfx + "post_attention_layernorm.weight"] = u16(layer.ln_ff).view(np.float16)
            qkv, gu = unpack(layer.qkv.weight), unpack(layer.w_in_gated.weight)
            om.km[pfx + "self_attn.q_proj"] = rows(qkv, slice(0, hd))
            om.km[pfx + "self_attn.k_proj"] = rows(qkv, slice(hd, hd + kvd))
            om.km[pfx + "self_attn.v_proj"] = rows(qkv, slice(hd + kvd, hd + 2 * kvd))
            om.km[pfx + "self_attn.o_proj"] = unpack(layer.attn_out.weight)
            om.km[pfx + "mlp.gate_proj"] = rows(gu, slice(0, ff))            # w_in (the activated half), then w_gated
            om.km[pfx + "mlp.up_proj"] = rows(gu, slice(ff, 2 * ff))
            om.km[pfx + "mlp.down_proj"] = unpack(layer.w_out.weight)
        om.len_buf = int(ctx.kv[0].shape[2])
        om.kb = [[u16(ctx.kv[t][i, 0]).copy() for t in range(b)] for i in range(cfg.num_layers)]
        om.vb = [[u16(ctx.kv[t][i, 1]).copy() for t in range(b)] for i in range(cfg.num_layers)]
        got = model.encode(ctx).float().cpu().numpy().astype(np.float64)
        ids = np.arange(b, dtype=np.int32)
        ref, _ = om.step(ids, pos, flavour="E", commit=False)
        # the network's own conditioning: E with ONE fp16 rounding in 2 000 of layer 0's q projection moved by an ulp ("T": what any
        # other equally exact kernel does), and the reference's arithmetic ("R": fp16 partial sums of its warp-reduce kernel)
        ref_t, _ = om.step(ids, pos, flavour="T", commit=False)
        ref_r, _ = om.step(ids, pos, flavour="R", commit=False)
        scale = np.abs(ref).max()
        dist = lambda a, c: float(np.abs(a - c).max() / scale)
        err, t_e, r_e, err_r = dist(got, ref), dist(ref_t, ref), dist(ref_r, ref), dist(got, ref_r)
        best = ref.argmax(axis=1)
        same = bool(all(ref[r, best[r]] - ref[r, int(got[r].argmax())] <= 2e-3 * scale for r in range(b)))
        return {"what": "logits of the step that is timed (32 layers, %d keys of history, %d-row lm_head) vs the CPU oracle's evaluation of the "
                        "same network from the unpacked bench weights.  E: exact linears rounded once to fp16; T: E with one rounding in 2000 "
                        "of layer 0's q projection moved by an ulp (the network's conditioning); R: the reference's fp16 partial sums.  "
                        "Distances are max|a - b| / max|E|" % (pos[0], ref.shape[1]),
                "vs_E": round(err, 6), "T_vs_E": round(t_e, 6), "R_vs_E": round(r_e, 6), "vs_R": round(err_r, 6),
                # north_star's bar, unconditioned: within 1e-3 of the exact-linear network AND no further from the reference's arithmetic
                # than its own fp16 noise + 1e-3 (bench.py exits non-zero when this is false on the 8(d) weights)
                "within_north_star_bar": bool(err <= 1e-3 and err_r <= 1e-3 + r_e),
                # the two bars of tests/test_gpu_fullgeom.py: the conditioned one against E, north_star's against the reference path
                "within_conditioned_bar_vs_E": bool(err <= max(1e-3, 1.25 * t_e)), "within_bar_vs_R": bool(err_r <= 1e-3 + r_e),
                "greedy_token_agrees": same, "seconds": round(time.time() - t0, 1)}
    except Exception as e:      # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=0, help="debug only: fewer layers (marks the result invalid)")
    ap.add_argument("--no-ttft", action="store_true", help="skip the prompt-encode (TTFT) leg")
    ap.add_argument("--tp", action="store_true",
                    help="tensor-parallel decode over the N ranks (one model sharded over the GPUs, RCCL all-reduce after attn_out / "
                         "w_out, vocab-parallel lm_head) instead of N independent replicas; strong scaling")
    ap.add_argument("--quant", choices=["gptq", "int8"], default="gptq",
                    help="gptq = BASELINE configs[1] (the headline); int8 = configs[2] (AutoInt8 linears, use --batch 32)")
    ap.add_argument("--kv-cache-dtype", choices=["fp16", "int8"], default="fp16",
                    help="int8: the reference's KV_CACHE_DTYPE=int8 cache (u8 codes + fp32 scales); not the headline config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1234,
                    help="seed of the synthetic KV history and step tokens (torch's default generator is seeded from the OS per process "
                         "on this build: unseeded, every run timed -- and logit-checked -- another draw; profiles/r06_logit_check_draws.txt)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra legs of the default run (batch 8 / 32 of the headline model, INT8 route at batch 32)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from zhilight_amd import ops  # noqa: F401  (raises if the HIP library is missing)
    from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig

    cfg = ModelConfig.llama3_8b()
    if args.layers:
        cfg.num_layers = args.layers
    batch, seq = args.batch, args.seq
    int8 = args.quant == "int8"
    tp = None
    if args.tp and world > 1:
        from zhilight_amd.parallel import DirectTPGroup, TPGroup, max_over_ranks
        # the direct transports (own RCCL communicator + one-shot peer-read all-reduce with the residual add fused: one launch
        # per exchange, capturable); torch.distributed is then only the bootstrap channel
        try:
            tp = TPGroup() if os.environ.get("ZL_TP_BACKEND") == "torch" else DirectTPGroup()
        except Exception as e:                 # noqa: BLE001
            sys.stderr.write("bench: direct TP transports unavailable (%s); using torch.distributed collectives\n" % str(e).splitlines()[0])
            tp = TPGroup()
        torch.manual_seed(1234)            # every rank must draw the same tokens / KV contents
    model = LLaMA(cfg, QuantConfig(2, 0) if int8 else QuantConfig(5, 128), dev, tp=tp)
    # weights: SURVEY 8(d)'s synthetic GPTQ checkpoint (the recipe of tests/synth.py / tests/test_gpu_fullgeom.py, drawn on the device
    # and taken through the real load path) -- the draw the hard 1e-3 logit bar is held on.  The INT8 route and the TP leg keep
    # the direct packed draw (init_random); ZL_BENCH_WEIGHTS=random forces it for an A/B.
    synth_weights = not int8 and tp is None and os.environ.get("ZL_BENCH_WEIGHTS", "synthetic") != "random"
    if synth_weights:
        model.init_synthetic(seed=1234 + rank)
    else:
        model.init_random(seed=1234 + rank)
    len_buf = (seq + args.warmup + args.steps + 4 + 63) // 64 * 64
    torch.manual_seed(args.seed)           # the history and the tokens: one reproducible draw (every rank the same one)
    ctx = model.new_context(batch, len_buf, seq, fill_random=True, kv_cache_dtype=None if args.kv_cache_dtype == "fp16" else "int8")
    ctx.tokens.copy_(torch.randint(0, cfg.vocab_size, (batch,), device=dev, dtype=torch.int32))
    oracle_check = check_against_oracle(model, batch, dev) if (rank == 0 and not int8 and tp is None) else None
    logit_check = None
    if (rank == 0 and world == 1 and not int8 and tp is None and batch == 1 and not args.no_cpu_baseline and not args.layers
            and args.kv_cache_dtype == "fp16" and os.environ.get("ZL_BENCH_LOGIT_CHECK", "1") != "0"):
        logit_check = check_logits_against_oracle(model, ctx, dev)

    # ---- TTFT leg (rank 0, reported next to the decode metric): encode a `seq`-token prompt of one task
    # (M-tiled W4A16 GEMMs, causal attention) and pick the first token.  HIP events around eager launches.
    ttft_ms = None
    if rank == 0 and not args.no_ttft and tp is None:
        pctx = model.new_context(1, len_buf, 0)
        prompt = torch.randint(0, cfg.vocab_size, (seq,), device=dev, dtype=torch.int32)
        model.prefill(pctx, 0, prompt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            model.prefill(pctx, 0, prompt)
        e1.record()
        torch.cuda.synchronize()
        ttft_ms = e0.elapsed_time(e1) / reps
        del pctx
        model._bufs = {k: v for k, v in model._bufs.items() if not (isinstance(k, tuple) and k[0] == "prefill")}
        torch.cuda.empty_cache()

    # TP: every rank encodes the same prompt (the collectives need all of them); once single-stream, once as the
    # reference's DUAL_STREAM=1 route (all-reduce of one half on the second stream behind the other half's compute)
    ttft_dual_ms = None
    if tp is not None and not args.no_ttft:
        gen = torch.Generator(device=dev).manual_seed(7)
        prompt = torch.randint(0, cfg.vocab_size, (seq,), device=dev, dtype=torch.int32, generator=gen)

        def ttft_leg(dual):
            saved = {k: os.environ.get(k) for k in ("DUAL_STREAM", "DUAL_STREAM_THRESHOLD")}
            os.environ["DUAL_STREAM"] = "1" if dual else "0"
            os.environ["DUAL_STREAM_THRESHOLD"] = str(min(1024, seq - 1))
            try:
                pctx = model.new_context(1, len_buf, 0)
                model.prefill(pctx, 0, prompt)
                torch.cuda.synchronize()
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    model.prefill(pctx, 0, prompt)
                e1.record()
                torch.cuda.synchronize()
                return max_over_ranks(e0.elapsed_time(e1) / 3, dev)
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        ttft_ms = ttft_leg(False)
        ttft_dual_ms = ttft_leg(True)
        torch.cuda.empty_cache()

    def step():
        model.step_greedy(ctx)   # encode + greedy pick + advance of the batch state, all on the device

    step()  # eager once: allocates the step's buffers, caches device attributes
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    captured = False
    try:
        with torch.cuda.graph(graph):
            step()
        replay = graph.replay
        captured = True
    except RuntimeError as e:              # a collective that cannot be captured: run the step eagerly
        if tp is None:
            raise
        sys.stderr.write("bench: graph capture of the TP step failed (%s); timing eager launches\n" % str(e).splitlines()[0])
        torch.cuda.synchronize()
        replay = step
    for _ in range(args.warmup):
        replay()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        replay()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tp_expired = None
    if tp is not None and hasattr(tp, "oneshot"):
        tp_expired = tp.oneshot.status()       # bounded waits that expired (their outputs are NaN-poisoned by the kernel)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline leg (rank 0): the dominant kernel (W4A16 GEMV: k_w4a16_phase / k_w4a16_mfma by default, k_w4a16_gemm
    # under ZL_W4_ALGO=exact; int8 route: k_w8a8_phase)
    roof = None
    if rank == 0 and int8:
        roof = roofline_int8(model, cfg, batch, dev)
    if rank == 0 and not int8 and tp is None:
        roof = roofline_w4(model, cfg, batch, dev, ctx, bool(args.layers))

    # ---- extra legs of the default single-GPU run: batch 8 / 32 on the headline model, BASELINE configs[2] (INT8, batch 32)
    extras = None
    if (rank == 0 and world == 1 and not args.no_extras and not int8 and tp is None and batch == 1 and not args.layers
            and args.kv_cache_dtype == "fp16"):
        esteps, ewarm = max(8, args.steps // 2), max(2, args.warmup // 2)
        extras = [extra_decode_run(model, cfg, b, seq, esteps, ewarm, dev, False) for b in (8, 32)]
        del ctx, graph
        model8 = LLaMA(cfg, QuantConfig(2, 0), dev).init_random(seed=4321)
        extras.append(extra_decode_run(model8, cfg, 32, seq, esteps, ewarm, dev, True))
        del model8
        torch.cuda.empty_cache()

    if rank == 0:
        value = (1 if tp else world) * batch * args.steps / elapsed
        step_bytes = step_bytes_of(model, cfg, batch, seq, int8)
        out = {
            "metric": "decode tokens/s (Llama-3-8B %s, TP=1 per GPU, batch %d, seq %d)" % ("INT8" if int8 else "GPTQ-Int4", batch, seq),
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if tp else "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic", "seed": args.seed,
            "config": {"workload": ("Llama-3-8B INT8 (AutoInt8 linears) TP=1 batch=%d decode seq=%d (BASELINE configs[2])" if int8 else
                                    "Llama-3-8B GPTQ-Int4 g128 TP=1 batch=%d decode seq=%d (BASELINE configs[1])") % (batch, seq),
                       "layers": cfg.num_layers,
                       "parallelism": ("tp%d (one model over %d GPUs, RCCL all-reduce)" % (world, world)) if tp else "dp%d (independent TP=1 replicas)" % world,
                       "global_batch": (1 if tp else world) * batch, "seq_len": seq, "valid": not args.layers,
                       "kv_cache_dtype": args.kv_cache_dtype,
                       "w4_algo": None if int8 else ("mfma" if isinstance(model.layers[0].qkv.weight, ops.W4MWeight) else "exact"),
                       "weights": "SURVEY 8(d) synthetic GPTQ checkpoint (uniform nibbles, zeros 1..15, half-normal scales), device-drawn, "
                                  "loaded through the checkpoint path" if synth_weights else "packed-layout random draw (init_random)"},
            "per_gpu_tokens_per_s": round(value / world, 2),
            "note_tp": ("TP transports: %s, RCCL communicator over %d ranks (0 = none), one-shot exchange expired waits: %s, step %s; the "
                        "builder's boxes have one GPU: the exchange step is covered there by a two-process model test on one device "
                        "(tests/test_gpu_comm.py)" % (type(tp).__name__, getattr(tp, "rccl_ranks", world if tp else 0), tp_expired,
                                                      "captured in one hipGraph" if captured else "launched eagerly")) if tp else None,
            "ttft_ms": None if ttft_ms is None else round(ttft_ms, 3),
            "ttft_dual_stream_ms": None if ttft_dual_ms is None else round(ttft_dual_ms, 3),
            "ttft_note": "prompt of seq tokens, one task, first greedy token; eager launches, HIP events, mean of 3",
            "step_hbm_roofline_frac": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": roof,
            "other_batches": extras,
            "config5": config5_leg() if (world == 1 and not args.no_extras and not int8 and tp is None and batch == 1 and not args.layers) else None,
            "boundary_path": boundary_leg() if (world == 1 and not args.no_extras and not int8 and tp is None and batch == 1 and not args.layers) else None,
            "oracle_check": None if oracle_check is None else {
                "what": "layer 0 and last layer, four W4A16 linears each, HIP output vs the CPU oracle's exact product of the "
                        "unpacked (zl_w4m_unpack) bench weights, before the timed region", "max_err_over_max_ref": round(oracle_check, 6)},
            "logit_check": logit_check,
        }
        if world == 1 and not args.no_cpu_baseline and not int8:
            out["cpu_baseline"] = cpu_baseline(cfg, batch, seq)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
        if (synth_weights and isinstance(logit_check, dict) and logit_check.get("within_north_star_bar") is False
                and os.environ.get("ZL_BENCH_PARITY_SOFT", "0") != "1"):
            sys.stderr.write("bench: the timed model's logits miss north_star's 1e-3 bar on the SURVEY 8(d) weights: %s\n" % json.dumps(logit_check))
            sys.exit(3)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

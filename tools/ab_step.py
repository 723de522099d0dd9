"""A/B of the decode step under environment settings, in ONE process (one model load, one box, back to back): for every batch in
BATCHES and every setting the step is re-captured (hipGraph), timed over REPS replays x 3 (best and median), and the hidden rows
before the output norm are compared with the FIRST setting's of that batch.  Weights: bench.py's (SURVEY 8(d) synthetic GPTQ
checkpoint), 1024 cached tokens in 1152-slot buffers -- bench.py's geometry.

usage: BATCHES=1,8,32 python tools/ab_step.py "base" "ZL_ATTN_LA=1" "ZL_ATTN_LA=1,ZL_ATTN_LA_SPLIT=64" ...
       ("base" or "" = no override; every setting starts from the environment the tool was launched with)
Output: one line per (batch, setting) on stdout and, with OUT=path, the same as JSON lines."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig  # noqa: E402

REPS = int(os.environ.get("REPS", "40"))
BATCHES = [int(v) for v in os.environ.get("BATCHES", "1").split(",")]
dev = torch.device("cuda:0")
cfg = ModelConfig.llama3_8b()
cfg.num_layers = int(os.environ.get("LAYERS", "32"))
model = LLaMA(cfg, QuantConfig(5, 128), dev)
if os.environ.get("ZL_BENCH_WEIGHTS", "synthetic") == "random":
    model.init_random(seed=1234)
else:
    model.init_synthetic(seed=1234)
settings = sys.argv[1:] or ["base"]
base_env = dict(os.environ)
out = open(os.environ["OUT"], "a") if os.environ.get("OUT") else None
for batch in BATCHES:
    ref = None
    for st in settings:
        os.environ.clear()
        os.environ.update(base_env)
        if st not in ("", "base"):
            for kv in st.split(","):
                k, v = kv.split("=")
                os.environ[k] = v
        torch.manual_seed(7)                            # the same cached keys / values and tokens for every setting
        ctx = model.new_context(batch, 1152, 1024, fill_random=True)
        ctx.tokens.copy_(torch.randint(0, cfg.vocab_size, (batch,), device=dev, dtype=torch.int32))
        try:
            model.step_greedy(ctx)
            torch.cuda.synchronize()
            hid = model.last_hidden.float().clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                model.step_greedy(ctx)
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(3):
                e0.record()
                for _ in range(REPS):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / REPS)
        except Exception as e:                          # noqa: BLE001 (a setting that does not run must not cost the others)
            print(f"batch {batch:3d} {st:60s} FAILED: {type(e).__name__}: {str(e)[:200]}", flush=True)
            continue
        if ref is None:
            ref, same, dmax = hid, "-", 0.0
        else:
            dmax = float((hid - ref).abs().max() / ref.abs().max())
            same = "bit-identical" if dmax == 0.0 else "max|d|/max|ref| %.3g, differing %.1f %%" % (dmax, 100.0 * float((hid != ref).float().mean()))
        best, med = min(ts), statistics.median(ts)
        print(f"batch {batch:3d} {st:60s} best {best * 1e3:8.1f} us/step {batch * 1e3 / best:8.1f} tok/s | median {batch * 1e3 / med:8.1f} tok/s | hidden vs first: {same}", flush=True)
        if out:
            out.write(json.dumps({"batch": batch, "setting": st, "us_per_step_best": round(best * 1e3, 2), "tok_s_best": round(batch * 1e3 / best, 1),
                                  "tok_s_median": round(batch * 1e3 / med, 1), "hidden_vs_first_max_rel": dmax}) + "\n")
            out.flush()
        del g, ctx
        torch.cuda.empty_cache()

"""Batch-1 decode step (Llama-3-8B GPTQ shapes, 1024 cached tokens, hipGraph replay) under several environment settings in ONE
process: the step is re-captured per setting, timed over REPS replays, and the hidden state before the output norm is compared
with the first setting's (settings that only move bytes earlier must not change a bit).
usage: python tools/ubench/step_variants.py "ZL_W4_L2_HINT=0" "ZL_W4_L2_HINT=1" "ZL_W4_L2_HINT=3" ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig  # noqa: E402

REPS = int(os.environ.get("REPS", "300"))
dev = torch.device("cuda:0")
cfg = ModelConfig.llama3_8b()
cfg.num_layers = int(os.environ.get("LAYERS", "32"))
model = LLaMA(cfg, QuantConfig(5, 128), dev).init_random(seed=1)
settings = sys.argv[1:] or ["ZL_W4_L2_HINT=0", "ZL_W4_L2_HINT=1", "ZL_W4_L2_HINT=2", "ZL_W4_L2_HINT=3"]
ref = None
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for st in settings:
        for kv in st.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
        torch.manual_seed(7)                            # the same cached keys / values for every setting
        ctx = model.new_context(1, 1152, 1024, fill_random=True)
        ctx.tokens.fill_(17)
        model.step_greedy(ctx)
        torch.cuda.synchronize()
        hid = model.last_hidden.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            model.step_greedy(ctx)
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(REPS):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / REPS)
        same = "-" if ref is None else ("bit-identical" if torch.equal(hid, ref) else "DIFFERENT (max |d| %.3g)" % (hid.float() - ref.float()).abs().max().item())
        if ref is None:
            ref = hid
        print(f"{st:40s} {best * 1e3:8.1f} us/step  {1e3 / best:7.1f} tok/s   hidden vs first: {same}", flush=True)
        del g, ctx

"""bench.py's logit check (check_logits_against_oracle: the timed 32-layer model's logits against the CPU oracle's E / T / R evaluations
of the same network) over several draws of the synthetic KV history and step token -- the weights stay bench.py's (init_synthetic,
seed 1234).  Why: torch's default generator is seeded from the OS per process on this build, so until bench.py seeded it every run
checked another draw, and one run of round 6 landed on a draw whose own conditioning (T_vs_E) was 7 x the usual one.
usage: python tools/logit_check_draws.py [seed ...]      one JSON line per seed"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig
    seeds = [int(a) for a in sys.argv[1:]] or [1234, 1, 2, 3, 4, 5, 6, 7]
    dev = torch.device("cuda:0")
    cfg = ModelConfig.llama3_8b()
    model = LLaMA(cfg, QuantConfig(5, 128), dev)
    model.init_synthetic(seed=1234)
    seq, len_buf = 1024, (1024 + 8 + 64 + 4 + 63) // 64 * 64
    for s in seeds:
        torch.manual_seed(s)
        ctx = model.new_context(1, len_buf, seq, fill_random=True)
        ctx.tokens.copy_(torch.randint(0, cfg.vocab_size, (1,), device=dev, dtype=torch.int32))
        lc = bench.check_logits_against_oracle(model, ctx, dev)
        lc = {k: v for k, v in lc.items() if k != "what"}
        print(json.dumps({"seed": s, "token": int(ctx.tokens[0]), **lc}), flush=True)
        del ctx
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

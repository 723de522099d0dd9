"""One rank's share of a tensor-parallel model as a decode-step timing (BASELINE configs[3]: Qwen2-72B GPTQ-Int4
TP = 4).  The rank's shard -- 16 of 64 heads, 2 of 8 kv heads, 7424 of 29696 feed-forward columns, 38016 of 152064 vocabulary rows,
80 layers: 9.1 GB of packed int4 weights -- runs the real TP decode step (column-parallel qkv / gate|up, row-parallel attn_out /
w_out, vocab-parallel lm_head) on ONE GPU; the exchange is EMULATED: every all-reduce is the one-shot kernel of the shipped
transport (zl_ar_all_reduce, residual add fused) launched with a world of one -- publish, flag, reduce, add: the launch and its
local work without the peers' bytes -- and the logits gather is a local concatenation.  What the number is: the per-rank compute +
launch time of a TP = 4 step, i.e. an upper bound on tokens/s of the group before link time (a 16 KB hidden row per exchange).

    python tools/bench_tp_shard.py [--layers 80] [--steps 20] [--batch 1]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=80)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--tp", type=int, default=4)
    a = ap.parse_args()
    from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig
    from zhilight_amd.parallel import OneShotAllReduce, TPGroup
    dev = torch.device("cuda:0")

    class EmulatedTP(TPGroup):
        def __init__(self, size):
            self.group, self.rank, self.size = None, 0, size
            addr, _ = OneShotAllReduce.alloc(1 << 20)
            self.oneshot = OneShotAllReduce(0, 1, [addr], 1 << 20, dev)

        def all_reduce_sum(self, t):
            return self.oneshot.all_reduce(t)

        def all_reduce_add(self, part, hidden):
            return self.oneshot.all_reduce(part, residual=hidden, out=hidden)

        def all_gather_columns(self, t):
            return torch.cat([t] * self.size, dim=-1)

    full = ModelConfig(num_layers=a.layers, dim_model=8192, num_heads=64, dim_head=128, dim_ff=29696, vocab_size=152064, num_kv_heads=8,
                       eps=1e-6, rope_theta=1e6)
    model = LLaMA(full, QuantConfig(5, 128), dev, tp=EmulatedTP(a.tp)).init_random(seed=1)
    c = model.cfg
    len_buf = (a.seq + a.steps + 8 + 63) // 64 * 64
    ctx = model.new_context(a.batch, len_buf, a.seq, fill_random=True)
    ctx.tokens.copy_(torch.randint(0, full.vocab_size, (a.batch,), device=dev, dtype=torch.int32))
    model.step_greedy(ctx)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        model.step_greedy(ctx)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    wbytes = model.weight_bytes() + model.lm_head.numel() * 2
    kv = a.batch * c.num_layers * 2 * c.num_kv_heads * a.seq * c.dim_head * 2
    rec = {"workload": "Qwen2-72B GPTQ-Int4 g128, one rank of TP=%d (exchange emulated on one GPU), batch %d, KV %d" % (a.tp, a.batch, a.seq),
           "layers": c.num_layers, "local_heads": c.num_heads, "local_kv_heads": c.num_kv_heads, "local_dim_ff": c.dim_ff,
           "ms_per_step": round(dt * 1e3, 4), "tokens_per_s_upper_bound": round(a.batch / dt, 1), "us_per_layer": round(dt * 1e6 / c.num_layers, 2),
           "weight_bytes_local": int(wbytes), "hbm_TBps": round((wbytes + kv) / dt / 1e12, 3), "hbm_frac_of_8TBps": round((wbytes + kv) / dt / 8e12, 4),
           "exchanges_per_step": 2 * c.num_layers, "exchange_message_bytes": a.batch * full.dim_model * 2}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()

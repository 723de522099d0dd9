"""Worker of tests/test_gpu_comm.py::test_model_on_the_direct_transport_*: one rank of a TP model on its REAL exchange step --
LLaMA(tp=DirectTPGroup(...)): one-shot peer-read all-reduce over hipIpc-mapped buffers with the residual add fused, logits
all-gather -- decoding under hipGraph capture, checked against the unsharded model every rank builds from the same seed.
usage: python _tp_worker.py <rank> <world> <exchange dir> <device index of this rank> <rccl 0|1>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

torch.set_num_threads(1)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rank, world, xdir, devi, rccl = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5] == "1"
torch.cuda.set_device(devi)
dev = torch.device("cuda", devi)
dist.init_process_group("gloo", init_method="file://" + os.path.join(xdir, "rdv"), rank=rank, world_size=world)
from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig  # noqa: E402
from zhilight_amd.parallel import DirectTPGroup  # noqa: E402
from test_gpu_model import _hf_state  # noqa: E402

rng = np.random.default_rng(41)
cfg = ModelConfig(num_layers=2, dim_model=1024, num_heads=8, dim_head=128, dim_ff=2048, vocab_size=512, num_kv_heads=2,
                  eps=1e-5, rope_theta=5e5)
sd = {k: torch.from_numpy(v) for k, v in _hf_state(rng, cfg, 128).items()}
ref_model = LLaMA(cfg, QuantConfig(5, 128), dev).load_state_dict(sd)
tp = DirectTPGroup(oneshot_bytes=1 << 20, device=dev, rccl=rccl)
model = LLaMA(cfg, QuantConfig(5, 128), dev, tp=tp).load_state_dict(sd)
assert model.cfg.num_heads == cfg.num_heads // world and model.lm_head.shape[0] == cfg.vocab_size // world
batch, len_buf, steps = 2, 64, 3
tokens = torch.from_numpy(rng.integers(0, cfg.vocab_size, batch).astype(np.int32))
ref_ctx, ctx = ref_model.new_context(batch, len_buf, 0), model.new_context(batch, len_buf, 0)
ref_ctx.tokens.copy_(tokens)
ctx.tokens.copy_(tokens)
ok, detail = True, ""
# step 0 eagerly (allocates the step's buffers), then ONE captured step replayed for the rest: the exchange kernel's message
# numbers live on the device, so a replay is a new message
logits = model.encode(ctx)
dist.barrier()
torch.cuda.synchronize()
graph, captured = None, None
for step in range(steps):
    ref = ref_model.encode(ref_ctx).float()
    if step == 1:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = model.encode(ctx)
    if step >= 1:
        dist.barrier()                    # both ranks replay the same step
        graph.replay()
        logits = captured
    got = logits.float()
    torch.cuda.synchronize()
    tp.check()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    if not (torch.isfinite(got).all().item() and err <= 2e-3 * scale and torch.equal(got.argmax(dim=-1), ref.argmax(dim=-1))):
        ok, detail = False, f"step {step}: err {err:.3e} scale {scale:.3e}"
        break
    # every rank holds the full logits, identical bit for bit
    mine = got.cpu()
    others = [None] * world
    dist.all_gather_object(others, mine.numpy().tobytes())
    if any(o != others[0] for o in others):
        ok, detail = False, f"step {step}: ranks disagree"
        break
    nxt = ref.argmax(dim=-1)
    ref_model.advance(ref_ctx, nxt)
    model.advance(ctx, nxt)
dist.barrier()
print(f"RESULT {rank} {'ok' if ok else 'FAILED ' + detail} rccl_ranks={tp.rccl_ranks} captured={graph is not None}", flush=True)
dist.destroy_process_group()

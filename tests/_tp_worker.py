"""Worker of tests/test_gpu_comm.py::test_model_on_the_direct_transport_*: one rank of a TP model on its REAL exchange step --
LLaMA(tp=DirectTPGroup(...)): one-shot peer-read all-reduce over hipIpc-mapped buffers with the residual add fused, logits
all-gather -- decoding under hipGraph capture, checked against the unsharded model every rank builds from the same seed.
Then the PROMPT leg: a 2304-token prompt under DUAL_STREAM=1 -- EncoderLayer::dual_stream_encode
(src/nn/block/block.cpp:205-441): two halves, every row-parallel partial (1152 x 1024 halfs = 2.4 MB) all-reduced by the
one-shot exchange on the second, high-priority stream while the main stream computes the other half -- logits against the
unsharded model's single-stream encode AND against the CPU oracle's prompt encode, this rank's KV rows against the unsharded
model's, and a decode step on top of that KV.
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
sys.path.insert(0, os.path.join(ROOT, "oracle"))
rank, world, xdir, devi, rccl = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5] == "1"
torch.cuda.set_device(devi)
dev = torch.device("cuda", devi)
dist.init_process_group("gloo", init_method="file://" + os.path.join(xdir, "rdv"), rank=rank, world_size=world)
from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig  # noqa: E402
from zhilight_amd.parallel import DirectTPGroup  # noqa: E402
from test_gpu_model import OracleModel, _hf_state  # noqa: E402

rng = np.random.default_rng(41)
cfg = ModelConfig(num_layers=2, dim_model=1024, num_heads=8, dim_head=128, dim_ff=2048, vocab_size=512, num_kv_heads=2,
                  eps=1e-5, rope_theta=5e5)
sd_np = _hf_state(rng, cfg, 128)
sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
ref_model = LLaMA(cfg, QuantConfig(5, 128), dev).load_state_dict(sd)
tp = DirectTPGroup(oneshot_bytes=4 << 20, device=dev, rccl=rccl)
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
# ---- prompt leg: dual-stream encode of 2304 tokens over the direct transport
prompt_detail = ""
if ok:
    s_prompt, len_buf2 = 2304, 2368
    prompt = torch.from_numpy(np.random.default_rng(7).integers(0, cfg.vocab_size, s_prompt).astype(np.int64))
    ref_pctx = ref_model.new_context(1, len_buf2, 0)
    ref_logits = ref_model.prefill(ref_pctx, 0, prompt).float()
    os.environ["DUAL_STREAM"] = "1"
    os.environ["DUAL_STREAM_THRESHOLD"] = "1024"
    pctx = model.new_context(1, len_buf2, 0)
    dist.barrier()
    got = model.prefill(pctx, 0, prompt).float()
    torch.cuda.synchronize()
    tp.check()
    scale = ref_logits.abs().max().item()
    err_ref = (got - ref_logits).abs().max().item() / scale
    # the CPU oracle's prompt encode: rank 0 computes it (all host cores), the others receive it
    box = [None]
    if rank == 0:
        torch.set_num_threads(os.cpu_count() or 1)
        import zl_oracle
        om = OracleModel(zl_oracle, cfg, sd_np, 128, 1, len_buf2)
        om.rope_kind = "plain"
        box[0] = om.prefill(0, prompt.numpy())
    dist.broadcast_object_list(box, src=0)
    ora = box[0]
    err_ora = float(np.abs(got.cpu().numpy().astype(np.float64) - ora).max() / np.abs(ora).max())
    err_ref_ora = float(np.abs(ref_logits.cpu().numpy().astype(np.float64) - ora).max() / np.abs(ora).max())
    # this rank's KV heads against the unsharded model's (rank r holds kv heads [r * hkv_local, (r + 1) * hkv_local))
    hl = cfg.num_kv_heads // world
    mine_kv = pctx.kv[0][:, :, :s_prompt].float()
    full_kv = ref_pctx.kv[0][:, :, :s_prompt, rank * hl:(rank + 1) * hl].float()
    kv_err = (mine_kv - full_kv).abs().max().item() / full_kv.abs().max().item()
    # layer 0 sees identical inputs: its rows differ by the summation order of a differently shaped GEMM at most
    kv0 = (mine_kv[0] - full_kv[0]).abs().max().item() / full_kv[0].abs().max().item()
    kv0_equal = kv0 <= 2.0 ** -9
    # one decode step on top of the prompt's KV
    step_ref = ref_model.encode(ref_pctx).float()
    step_got = model.encode(pctx).float()
    torch.cuda.synchronize()
    tp.check()
    err_step = (step_got - step_ref).abs().max().item() / step_ref.abs().max().item()
    prompt_detail = (f"prompt: dual_stream_runs={getattr(model, 'dual_stream_runs', 0)} vs_unsharded={err_ref:.2e} vs_oracle={err_ora:.2e} "
                     f"unsharded_vs_oracle={err_ref_ora:.2e} kv_err={kv_err:.2e} kv_layer0_equal={kv0_equal} next_step={err_step:.2e}")
    if not (getattr(model, "dual_stream_runs", 0) == 1 and torch.isfinite(got).all().item() and err_ref <= 2e-3 and err_ora <= 2e-3
            and err_ref_ora <= 1e-3 and kv_err <= 1e-2 and kv0_equal and err_step <= 2e-3
            and torch.equal(got.argmax(dim=-1), ref_logits.argmax(dim=-1))):
        ok, detail = False, prompt_detail
dist.barrier()
print(f"RESULT {rank} {'ok' if ok else 'FAILED ' + detail} rccl_ranks={tp.rccl_ranks} captured={graph is not None} {prompt_detail}", flush=True)
dist.destroy_process_group()

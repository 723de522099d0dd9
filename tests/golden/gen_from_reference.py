"""Generate golden fixtures from the reference's OWN in-test PyTorch models.

The reference's tests compare its CUDA layers against small PyTorch re-implementations that live in
the test files (tests/test_attention.py:11-240, tests/test_feedforward.py:12-108,
tests/test_linear.py, tests/test_rotary_embedding.py:11-90).  Those models are the only numeric
ground truth the reference holds for this path, so this script IMPORTS them from
/root/reference (read-only, never copied), runs them on CPU with seeded inputs and stores
inputs/outputs as tests/golden/reference_models.npz.  It must be run in the build container
(where /root/reference exists); the GPU box only reads the committed .npz.

    python tests/golden/gen_from_reference.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_models.npz")


def _load(name):
    """Import a reference test module with the CUDA-only extension stubbed and `device="cuda"` ignored."""
    stub = types.ModuleType("zhilight")
    internals = types.ModuleType("zhilight.internals_")
    internals.layers = types.SimpleNamespace()
    internals.functions = types.SimpleNamespace()
    stub.internals_ = internals
    sys.modules.setdefault("zhilight", stub)
    sys.modules.setdefault("zhilight.internals_", internals)
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_orig_arange = torch.arange


def _arange_cpu(*a, **kw):
    kw.pop("device", None)
    return _orig_arange(*a, **kw)


def main():
    torch.arange = _arange_cpu
    torch.manual_seed(0)
    out = {}
    t_attn = _load("test_attention")
    t_ff = _load("test_feedforward")
    t_rope = _load("test_rotary_embedding")

    # ---- 1. neox rotary embedding (tests/test_attention.py RotaryEmbeddingESM: fp32 inv_freq) -----
    d, h, s = 128, 2, 9
    rope = t_attn.RotaryEmbeddingESM(dim=d, dtype=torch.float32)
    q = torch.randn(1, h, s, d, dtype=torch.float32)
    k = torch.randn(1, h, s, d, dtype=torch.float32)
    q16, k16 = q.half().float(), k.half().float()
    rq, rk = rope(q16, k16)
    out.update(rope_q=q16.half().numpy(), rope_k=k16.half().numpy(), rope_q_out=rq.numpy(), rope_k_out=rk.numpy(),
               rope_base=np.float32(10000.0))
    # the fp16 variant used by tests/test_rotary_embedding.py (cos/sin in the table dtype)
    rope16 = t_rope.RotaryEmbeddingESM(dim=d, dtype=torch.float32)
    rq2, _ = rope16(q16, k16)
    out.update(rope_q_out2=rq2.numpy())

    # ---- 2. attention: projections + rope + masked softmax + P.V + out projection ------------------
    dim_model, heads = 256, 2
    attn = t_attn.Attention(dim_model, heads, d, "rotary", dtype=torch.float32)
    with torch.no_grad():
        for p in attn.parameters():
            p.copy_(p.half().float())           # weights representable in fp16
        hidden = (torch.randn(1, s, dim_model) * 0.5).half().float()
        mask = (torch.arange(s) <= torch.arange(s).view(-1, 1)).to(torch.int8).view(1, s, s)
        full = attn(hidden, mask, None)
        # intermediates, recomputed with the module's own sub-layers (same code path as forward)
        hq = attn.project_q(hidden).view(1, s, heads, d).permute(0, 2, 1, 3)
        hk = attn.project_k(hidden).view(1, s, heads, d).permute(0, 2, 1, 3)
        hv = attn.project_v(hidden).view(1, s, heads, d).permute(0, 2, 1, 3)
        hq_r, hk_r = attn.position_bias(hq, hk)
        score = torch.matmul(hq_r, hk_r.transpose(-1, -2)) / (d ** 0.5)
        score = score.masked_fill(mask.view(1, 1, s, s) == 0, float("-inf")).softmax(dim=-1)
        ctxv = torch.matmul(score, hv).permute(0, 2, 1, 3).reshape(1, s, heads * d)
        assert torch.allclose(attn.attn_out(ctxv), full, atol=1e-5)
    out.update(attn_hidden=hidden.half().numpy(), attn_wq=attn.project_q.weight.detach().half().numpy(),
               attn_wk=attn.project_k.weight.detach().half().numpy(), attn_wv=attn.project_v.weight.detach().half().numpy(),
               attn_wo=attn.attn_out.weight.detach().half().numpy(), attn_q=hq.numpy(), attn_q_rot=hq_r.numpy(),
               attn_k_rot=hk_r.numpy(), attn_v=hv.numpy(), attn_ctx=ctxv.numpy(), attn_out=full.numpy(),
               attn_mask=mask.numpy())

    # ---- 3. gated-GELU feed-forward (tests/test_feedforward.py FeedForward) ---------------------------
    ff = t_ff.FeedForward(128, 256, dtype=torch.float32)
    with torch.no_grad():
        for p in ff.parameters():
            p.copy_((p * 0.1).half().float())
        x = (torch.randn(1, 5, 128) * 0.5).half().float()
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):   # the reference model prints its gate
            y = ff(x)
        gate = ff.w_in(x)
        up = ff.w_gated(x)
    out.update(ff_x=x.half().numpy(), ff_w_in=ff.w_in.weight.detach().half().numpy(),
               ff_w_gated=ff.w_gated.weight.detach().half().numpy(), ff_w_out=ff.w_out.weight.detach().half().numpy(),
               ff_gate=gate.numpy(), ff_up=up.numpy(), ff_out=y.numpy())
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

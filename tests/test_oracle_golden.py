"""Pins the CPU oracle against the golden fixtures produced by the reference's own in-test PyTorch
models (tests/golden/gen_from_reference.py).  These are the only numeric ground truth the reference
holds for this path: neox RoPE, masked-softmax attention, gated feed-forward, linear.
The fixtures are fp32 PyTorch results on fp16-representable inputs, so the oracle (fp16 outputs) must
agree to fp16 rounding."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_models.npz"))


def _close(a, b, ulps=2.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    tol = ulps * 2.0 ** -11 * np.maximum(np.abs(b), 2.0 ** -4) + 1e-6
    assert (np.abs(a - b) <= tol).all(), float(np.abs(a - b).max())


def test_rope_matches_reference_model(oracle):
    """RotaryEmbeddingESM (tests/test_attention.py:29-92): x*cos + rotate_half(x)*sin, inv_freq = base^(-2i/d)."""
    q, k = G["rope_q"][0], G["rope_k"][0]                     # (h, s, d) fp16
    h, s, d = q.shape
    pos = np.arange(s, dtype=np.int32)
    # fused-qkv layout expected by the kernel: (s, (H + 2Hkv) * D) with q heads, then k heads, then v heads
    qkv = np.concatenate([q.transpose(1, 0, 2).reshape(s, h * d), k.transpose(1, 0, 2).reshape(s, h * d),
                          np.zeros((s, h * d), np.float16)], axis=1)
    oq, ok, _ = oracle.rotary_embedding_qk(pos, oracle.h2u(qkv), h, h, d, float(G["rope_base"]))
    _close(oracle.u2h(oq).reshape(s, h, d).transpose(1, 0, 2), G["rope_q_out"][0])
    _close(oracle.u2h(ok).reshape(s, h, d).transpose(1, 0, 2), G["rope_k_out"][0])
    cs, sn = oracle.rope_cos_sin(pos, d, float(G["rope_base"]), True)
    oq2, ok2, _ = oracle.rope_qk_cache(cs, sn, oracle.h2u(qkv), h, h, d, True)
    _close(oracle.u2h(oq2).reshape(s, h, d).transpose(1, 0, 2), G["rope_q_out"][0])


def test_linear_matches_reference_model(oracle):
    x = G["attn_hidden"][0]                                   # (s, dim) fp16
    y = oracle.u2h(oracle.gemm_nt(oracle.h2u(x), oracle.h2u(G["attn_wq"])))
    s = x.shape[0]
    ref = G["attn_q"][0].transpose(1, 0, 2).reshape(s, -1)    # (s, h*d)
    _close(y, ref)


def test_decode_attention_matches_reference_model(oracle):
    """softmax(q.K^T/sqrt(d) masked) . V of the reference's Attention model, row by row as decode steps."""
    qr, kr, v = G["attn_q_rot"][0], G["attn_k_rot"][0], G["attn_v"][0]    # (h, s, d) fp32
    h, s, d = qr.shape
    kb = [oracle.h2u(kr.transpose(1, 0, 2).astype(np.float16))]            # BSHD (s, h, d)
    vb = [oracle.h2u(v.transpose(1, 0, 2).astype(np.float16))]
    # compare on the fp16-rounded operands the oracle actually sees
    kf, vf = oracle.u2h(kb[0]).astype(np.float64), oracle.u2h(vb[0]).astype(np.float64)
    for t in (0, 3, s - 1):
        q16 = qr[:, t, :].astype(np.float16)
        mask = G["attn_mask"][0, t].astype(np.int8)
        out = oracle.mqa_rag_buffer(oracle.h2u(q16).reshape(1, 1, h, d), np.array([s], np.int32), kb, vb, mask, h,
                                    1.0 / np.sqrt(d), True)
        out_split = oracle.mqa_rag_buffer(oracle.h2u(q16).reshape(1, 1, h, d), np.array([s], np.int32), kb, vb, mask, h,
                                          1.0 / np.sqrt(d), True, num_split=2)
        sc = np.einsum("hd,shd->hs", q16.astype(np.float64), kf) / np.sqrt(d)
        sc = np.where(mask[None, :] != 0, sc, -np.inf)
        p = np.exp(sc - sc.max(axis=1, keepdims=True))
        p /= p.sum(axis=1, keepdims=True)
        ref = np.einsum("hs,shd->hd", p, vf)
        _close(oracle.u2h(out).reshape(h, d), ref)
        _close(oracle.u2h(out_split).reshape(h, d), ref)
    # and the stored PyTorch context (computed from unrounded q/k/v): fp16-level agreement
    t = s - 1
    q16 = qr[:, t, :].astype(np.float16)
    out = oracle.mqa_rag_buffer(oracle.h2u(q16).reshape(1, 1, h, d), np.array([s], np.int32), kb, vb,
                                G["attn_mask"][0, t].astype(np.int8), h, 1.0 / np.sqrt(d), True)
    _close(oracle.u2h(out).reshape(-1), G["attn_ctx"][0, t], ulps=40)


def test_feedforward_matches_reference_model(oracle):
    """gated GELU FF (tests/test_feedforward.py:44-108): w_out(gelu_tanh(w_in x) * w_gated x)."""
    x = G["ff_x"][0]
    gate = oracle.gemm_nt(oracle.h2u(x), oracle.h2u(G["ff_w_in"]))
    up = oracle.gemm_nt(oracle.h2u(x), oracle.h2u(G["ff_w_gated"]))
    _close(oracle.u2h(gate), G["ff_gate"][0])
    _close(oracle.u2h(up), G["ff_up"][0])
    act = oracle.gelu_mul(gate, up)
    y = oracle.u2h(oracle.gemm_nt(act, oracle.h2u(G["ff_w_out"])))
    # three fp16 roundings (gate, up, act) feed a 256-term sum on the oracle side vs fp32 PyTorch:
    # compare against the magnitude of the output vector (reference test bar: atol 1e-2, test_feedforward.py:115)
    ref = G["ff_out"][0].astype(np.float64)
    assert np.abs(y.astype(np.float64) - ref).max() <= 2e-3 * np.abs(ref).max()

"""CPU checks of the oracle's scaled-rope restatements (rotary_embedding.cu "dynamic" and YarnImpl) and q / k head norms
against independent formulas: transformers' published YaRN init (when importable) and plain fp64 numpy."""
import numpy as np
import pytest


def test_yarn_tables_agree_with_published_yarn(oracle):
    d, theta, factor, orig = 128, 1e6, 4.0, 32768
    low, high, msc = oracle.yarn_params(theta, d, orig, factor)
    pos = np.array([0, 1, 77, 1000, 32768, 100000], np.int32)
    cs, sn = oracle.rope_cos_sin_yarn(pos, d, theta, factor, low, high, msc)
    # fp64 restatement of the published algorithm (arXiv 2309.00071 / HF _compute_yarn_parameters)
    i = np.arange(d // 2, dtype=np.float64)
    pos_freq = theta ** (2 * i / d)
    corr = lambda rot: d * np.log(orig / (rot * 2 * np.pi)) / (2 * np.log(theta))   # noqa: E731
    lo, hi = max(np.floor(corr(32)), 0), min(np.ceil(corr(1)), d - 1)
    assert (low, high) == (lo, hi)
    ramp = np.clip((i - lo) / (hi - lo), 0, 1)
    inv = (1 / (factor * pos_freq)) * ramp + (1 / pos_freq) * (1 - ramp)
    att = 0.1 * np.log(factor) + 1.0
    assert abs(msc - att) < 1e-6
    ang = pos[:, None] * inv[None, :]
    bound = 1e-5 + 2.0 ** -22 * np.maximum(pos[:, None], 1)
    for half in (slice(0, d // 2), slice(d // 2, d)):                           # neox: both halves carry the same angles
        assert (np.abs(cs[:, half] - np.cos(ang) * att) <= bound).all()
        assert (np.abs(sn[:, half] - np.sin(ang) * att) <= bound).all()
    try:
        from transformers import LlamaConfig
        from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
        cfg = LlamaConfig(hidden_size=32 * d, num_attention_heads=32, rope_theta=theta, max_position_embeddings=4 * orig,
                          rope_scaling={"rope_type": "yarn", "factor": factor, "original_max_position_embeddings": orig})
        hf_inv, hf_att = ROPE_INIT_FUNCTIONS["yarn"](cfg, "cpu")
    except Exception as e:                                                          # API drift across versions
        pytest.skip(f"transformers yarn init not usable here: {e}")
    assert np.abs(hf_inv.double().numpy() - inv).max() <= 1e-6 * inv.max() and abs(hf_att - att) < 1e-6


def test_yarn_deepseek_mscale(oracle):
    low, high, m = oracle.yarn_params(1e4, 64, 4096, 40.0, 32, 1, 1.0, True, 0.707, 0.707)
    assert m == pytest.approx(1.0)                   # mscale == mscale_all_dim: the two factors cancel (deepseek_v2 configs)
    _, _, m2 = oracle.yarn_params(1e4, 64, 4096, 40.0, 32, 1, 1.0, True, 1.0, 0.0)
    assert m2 == pytest.approx(0.1 * np.log(40.0) + 1.0, rel=1e-6)
    assert oracle.yarn_params(1e4, 64, 4096, 1.0)[2] == 1.0


def test_dynamic_ntk_keeps_the_reference_integer_exponent(oracle):
    d, theta, factor, maxp = 128, 1e4, 2.0, 4096
    pos = np.array([5, 4096, 5000, 9000], np.int32)
    cs, sn = oracle.rope_cos_sin_dynamic(pos, d, theta, factor, maxp)
    i = np.arange(d // 2, dtype=np.float64)
    for r, p in enumerate(pos):
        th = theta
        if p > maxp:
            th = theta * ((factor * p / maxp) - (factor - 1)) ** (d // (d - 2))        # int / int == 1 (rotary_embedding.cu:38)
        ang = p * th ** (-2 * i / d)
        bound = 1e-5 + 2.0 ** -21 * max(p, 1)
        assert (np.abs(cs[r, :d // 2] - np.cos(ang)) <= bound).all() and (np.abs(sn[r, d // 2:] - np.sin(ang)) <= bound).all()
    # one sequence length for the whole call (the reference reads the last row's position): rows below the threshold
    # are rotated with the scaled base too
    seq = np.full(pos.size, 9000, np.int32)
    cs2, _ = oracle.rope_cos_sin_dynamic(pos, d, theta, factor, maxp, seq)
    assert not np.allclose(cs2[0], cs[0]) and np.array_equal(cs2[3], cs[3])


@pytest.mark.parametrize("mode", [0, 1])
def test_head_norm_against_fp64(oracle, mode):
    rng = np.random.default_rng(mode)
    rows, heads, d = 5, 6, 128
    x = (rng.standard_normal((rows, heads * d)) * 3 + 0.5).astype(np.float16)
    w = (1 + 0.2 * rng.standard_normal(d if mode == 0 else heads * d)).astype(np.float16)
    got = oracle.u2h(oracle.head_norm(oracle.h2u(x), oracle.h2u(w), heads, d, 1e-6, mode)).astype(np.float64)
    v = x.astype(np.float64).reshape(rows, heads, d)
    if mode == 1:
        v = v - v.mean(axis=2, keepdims=True)
    wv = w.astype(np.float64).reshape((1, 1, d) if mode == 0 else (1, heads, d))
    ref = (v / np.sqrt((v * v).mean(axis=2, keepdims=True) + 1e-6) * wv).reshape(rows, heads * d)
    assert np.abs(got - ref).max() <= 2.0 ** -10 * np.abs(ref).max()

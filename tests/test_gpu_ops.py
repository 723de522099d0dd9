"""GPU parity for the non-GEMM rows of SURVEY 8a (a8-a11, a13-a18, a21, a22) through the C ABI vs the
CPU oracle.  Integer / index / copy work is bit-exact; fp work within the stated ulp / relative bars."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.view(dtype)


def _np(t):
    return t.detach().cpu().numpy()


def _bits(t):
    return _np(t.view(torch.int16)).view(np.uint16)


def _to_bits(x, dtype, oracle):
    return oracle.f32_to_bf16(x.astype(np.float32)) if dtype else oracle.h2u(x.astype(np.float16))


def _tt(bits, dev, dtype):
    return _t(bits.view(np.int16), dev, torch.bfloat16 if dtype else torch.float16)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("dim,eps", [(4096, 1e-5), (2304, 1e-6), (8192, 1e-5)])
def test_rmsnorm(oracle, dev, dim, eps, dtype):
    from zhilight_amd import ops
    rng = np.random.default_rng(0)
    x = _to_bits(rng.standard_normal((5, dim)) * 2, dtype, oracle)
    x2 = _to_bits(rng.standard_normal((5, dim)), dtype, oracle)
    w = _to_bits(1 + 0.1 * rng.standard_normal(dim), dtype, oracle)
    ref = oracle.rmsnorm(x, w, eps, dtype=dtype)
    got = _bits(ops.rmsnorm(_tt(x, dev, dtype), _tt(w, dev, dtype), eps))
    ulp = synth.ulp_diff_f16(got, ref)
    assert ulp.max() <= 1 and (ulp > 0).mean() < 0.02, (ulp.max(), (ulp > 0).mean())   # bar: <= 1 ulp of T
    exact = oracle.rmsnorm_exact(x, w, eps, dtype=dtype)
    assert np.abs(oracle.to_f32(got, dtype) - exact).max() <= np.abs(exact).max() * (2 ** -8 if dtype else 2 ** -11) * 1.01
    ref2, refsum = oracle.rmsnorm(x, w, eps, x2=x2, dtype=dtype)
    o, osum = ops.rmsnorm(_tt(x, dev, dtype), _tt(w, dev, dtype), eps, x2=_tt(x2, dev, dtype))
    assert np.array_equal(_bits(osum), refsum)   # T(f32(a)+f32(b)): exact
    ulp = synth.ulp_diff_f16(_bits(o), ref2)
    assert ulp.max() <= 1 and (ulp > 0).mean() < 0.02


@pytest.mark.parametrize("theta", [1e4, 5e5])
def test_rope_tables_and_fused_qk(oracle, dev, theta):
    from zhilight_amd import ops
    rng = np.random.default_rng(1)
    h, hkv, d = 8, 2, 128
    pos = np.concatenate([np.arange(0, 70), [1024, 4095, 8191, 131071]]).astype(np.int32)
    s = pos.size
    for llama3 in (None, (8.0, 1.0, 4.0, 8192.0)):
        rc, rs = oracle.rope_cos_sin(pos, d, theta, True, llama3)
        gc, gs = ops.rope_cos_sin(_t(pos, dev), d, theta, True, llama3)
        # freq = pos * powf(theta, -2i/D): device and glibc powf may differ by an ulp, which the
        # multiplication by pos amplifies -> bound the table error by 4 ulp of the ANGLE (+ 1e-6)
        ang = pos[:, None].astype(np.float64) * 1.0  # |freq| <= pos since inv_freq <= 1
        bound = 1e-6 + 4 * 2.0 ** -24 * np.maximum(ang, 1.0)
        assert (np.abs(_np(gc) - rc) <= bound).all() and (np.abs(_np(gs) - rs) <= bound).all()
    x = synth.act(rng, s, (h + 2 * hkv) * d)
    rq, rk, rv = oracle.rotary_embedding_qk(pos, oracle.h2u(x), h, hkv, d, theta)
    gq, gk, gv = ops.rotary_embedding_qk(_t(pos, dev), _t(x, dev), h, hkv, d, theta)
    assert np.array_equal(_bits(gv), rv)
    # bar: 2 fp16 ulp at the magnitude of the rotated pair, plus the angle's own fp32 uncertainty
    # (freq = pos * powf(...) carries ~2 ulp of a value up to `pos` radians)
    for g_, r_, nh in ((gq, rq, h), (gk, rk, hkv)):
        gv_ = oracle.u2h(_bits(g_)).astype(np.float64).reshape(s, nh, d)
        rv_ = oracle.u2h(r_).astype(np.float64).reshape(s, nh, d)
        mag = np.sqrt(rv_[..., : d // 2] ** 2 + rv_[..., d // 2:] ** 2)
        mag = np.concatenate([mag, mag], axis=-1)
        tol = 2 * 2.0 ** -10 * np.maximum(mag, 2.0 ** -10) + mag * (4 * 2.0 ** -24 * pos[:, None, None])
        assert (np.abs(gv_ - rv_) <= tol + 1e-7).all(), np.abs(gv_ - rv_).max()
    # cached variant: identical cos/sin input -> bit-exact except fma association (<= 1 ulp)
    rc, rs = oracle.rope_cos_sin(pos, d, theta, True, (8.0, 1.0, 4.0, 8192.0))
    for neox in (True, False):
        rq, rk, rv = oracle.rope_qk_cache(rc, rs, oracle.h2u(x), h, hkv, d, neox)
        gq, gk, gv = ops.rope_qk_cache(_t(rc, dev), _t(rs, dev), _t(x, dev), h, hkv, d, neox)
        assert np.array_equal(_bits(gv), rv)
        assert np.array_equal(_bits(gq), rq) and np.array_equal(_bits(gk), rk)
        # a few rows only: the element-per-thread kernel (>= 16 rows run the 16-byte-per-thread one above)
        gq5, gk5, gv5 = ops.rope_qk_cache(_t(rc[:5], dev), _t(rs[:5], dev), _t(x[:5], dev), h, hkv, d, neox)
        assert np.array_equal(_bits(gq5), rq[:5]) and np.array_equal(_bits(gk5), rk[:5]) and np.array_equal(_bits(gv5), rv[:5])


def _make_kv(rng, lens, hkv, d, bshd, dev, dtype=0, oracle=None):
    kb, vb = [], []
    for L in lens:
        shape = (L, hkv, d) if bshd else (hkv, L, d)
        kb.append(_to_bits(rng.standard_normal(shape), dtype, oracle))
        vb.append(_to_bits(rng.standard_normal(shape), dtype, oracle))
    tdt = torch.bfloat16 if dtype else torch.float16
    dk = [_t(a.view(np.int16), dev, tdt) for a in kb]
    dv = [_t(a.view(np.int16), dev, tdt) for a in vb]
    return kb, vb, dk, dv


@pytest.mark.parametrize("bshd", [True, False])
def test_kv_scatter_bit_exact(oracle, dev, bshd):
    from zhilight_amd import ops
    rng = np.random.default_rng(2)
    hkv, d, len_q = 8, 128, 3
    lens = [64, 128, 192, 65]
    kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, bshd, dev, oracle=oracle)
    b = len(lens)
    placement = np.stack([rng.choice(L, len_q, replace=False) for L in lens]).astype(np.int32)
    placement[1, 1] = -1
    ks, vs = synth.act(rng, b * len_q, hkv * d).reshape(b, len_q, hkv, d), synth.act(rng, b * len_q, hkv * d).reshape(b, len_q, hkv, d)
    oracle.copy_to_rag_buffer2(placement, np.array(lens, np.int32), oracle.h2u(ks), oracle.h2u(vs), kb, vb, bshd)
    ops.copy_to_rag_buffer2(_t(placement, dev), _t(np.array(lens, np.int32), dev), _t(ks, dev), _t(vs, dev),
                            ops.make_ptr_table(dk), ops.make_ptr_table(dv), bshd)
    for i in range(b):
        assert np.array_equal(_bits(dk[i]), kb[i]) and np.array_equal(_bits(dv[i]), vb[i])


def _mask_for(lens, len_q, rng, mode):
    parts = []
    for L in lens:
        if mode == "prefix":
            vis = rng.integers(1, L + 1)
            m = np.zeros((len_q, L), np.int8)
            m[:, :vis] = 1
        else:
            m = (rng.random((len_q, L)) < 0.7).astype(np.int8)
            m[:, 0] = 1
        parts.append(m.reshape(-1))
    return np.concatenate(parts)


@pytest.mark.parametrize("h,hkv,d,len_q", [(32, 8, 128, 1), (32, 32, 128, 1), (16, 2, 64, 1), (8, 2, 128, 4), (4, 4, 256, 2)])
@pytest.mark.parametrize("bshd", [True, False])
def test_decode_attention(oracle, dev, h, hkv, d, len_q, bshd):
    """A.6 oracle (fp32, reference association) and fp64 oracle; bar 1e-3 relative to max|out| (the
    reference's own test uses atol = max|out|/100, tests/test_attention.py:320-322)."""
    from zhilight_amd import ops
    rng = np.random.default_rng(3)
    lens = [1, 63, 64, 130, 1088, 517]
    b = len(lens)
    kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, bshd, dev, oracle=oracle)
    q = synth.act(rng, b * len_q * h, d).reshape(b, len_q, h, d)
    scale = 1.0 / np.sqrt(d)
    for mode in ("prefix", "random"):
        mask = _mask_for(lens, len_q, rng, mode)
        ref = oracle.u2h(oracle.mqa_rag_buffer(oracle.h2u(q), np.array(lens, np.int32), kb, vb, mask, hkv, scale, bshd)).astype(np.float64)
        exact = oracle.mqa_rag_buffer(oracle.h2u(q), np.array(lens, np.int32), kb, vb, mask, hkv, scale, bshd, exact=True)
        got = ops.multi_query_attention_rag_buffer(_t(q, dev), _t(np.array(lens, np.int32), dev), ops.make_ptr_table(dk),
                                                   ops.make_ptr_table(dv), _t(mask, dev), scale, max(lens), hkv, bshd=bshd)
        g = _np(got).astype(np.float64)
        assert np.isfinite(g).all()
        tol = 1e-3 * max(1.0, np.abs(exact).max())
        assert np.abs(g - exact).max() < tol, np.abs(g - exact).max()
        assert np.abs(g - ref).max() < 2 * tol


def test_decode_attention_valid_lens_and_bf16(oracle, dev):
    from zhilight_amd import ops
    rng = np.random.default_rng(4)
    h, hkv, d = 32, 8, 128
    lens = [1088, 1088, 256]
    valid = [1025, 7, 256]
    for dtype in (0, 1):
        kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, True, dev, dtype, oracle)
        q = _to_bits(rng.standard_normal((3, 1, h, d)), dtype, oracle)
        mask = np.concatenate([(np.arange(L) < v).astype(np.int8) for L, v in zip(lens, valid)])
        exact = oracle.mqa_rag_buffer(q, np.array(lens, np.int32), kb, vb, mask, hkv, 0.088, True, dtype=dtype, exact=True)
        got = ops.multi_query_attention_rag_buffer(_tt(q, dev, dtype), _t(np.array(lens, np.int32), dev),
                                                   ops.make_ptr_table(dk), ops.make_ptr_table(dv), None, 0.088, max(lens), hkv,
                                                   valid_lens=_t(np.array(valid, np.int32), dev))
        g = oracle.to_f32(_bits(got), dtype).astype(np.float64)
        rel = 1e-3 if dtype == 0 else 5e-3   # output rounding of bf16 is 2^-9 relative
        assert np.abs(g - exact).max() < rel * max(1.0, np.abs(exact).max())


@pytest.mark.parametrize("h,hkv,len_q", [(32, 8, 1), (32, 32, 1), (16, 1, 1), (8, 2, 2), (16, 4, 4), (28, 4, 1), (24, 8, 5)])
@pytest.mark.parametrize("bshd", [True, False])
@pytest.mark.parametrize("dtype", [0, 1])
def test_decode_attention_matrix_core_path(oracle, dev, h, hkv, len_q, bshd, dtype):
    """prefix visibility (valid_lens), fp16 / bf16, D = 128, up to 16 query rows per kv head: k_decode_attn_mfma
    (probabilities as a hi + lo pair, fp32 accumulation) vs the fp64 oracle; ragged lengths around the 32-key chunk and
    128-key split boundaries; NaN in the never-visible tail of V must not leak."""
    from zhilight_amd import ops
    rng = np.random.default_rng(33)
    d = 128
    lens = [64, 64, 64, 160, 160, 160, 1088, 1088, 640]
    valid = [1, 31, 33, 127, 128, 129, 1025, 517, 640]
    b = len(lens)
    kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, bshd, dev, dtype, oracle)
    tdt = torch.bfloat16 if dtype else torch.float16
    for bi, (L, v) in enumerate(zip(lens, valid)):       # poison (device copy only) what must never reach the result
        if v < L:
            pv = vb[bi].copy()
            if bshd:
                pv[v:] = np.uint16(0x7fc0 if dtype else 0x7e00)
            else:
                pv[:, v:] = np.uint16(0x7fc0 if dtype else 0x7e00)
            dv[bi].copy_(_t(pv.view(np.int16), dev, tdt))
    q = _to_bits(rng.standard_normal((b, len_q, h, d)), dtype, oracle)
    scale = 1.0 / np.sqrt(d)
    mask = np.concatenate([np.tile((np.arange(L) < v).astype(np.int8), len_q) for L, v in zip(lens, valid)])
    exact = oracle.mqa_rag_buffer(q, np.array(lens, np.int32), kb, vb, mask, hkv, scale, bshd, dtype=dtype, exact=True)
    got = ops.multi_query_attention_rag_buffer(_tt(q, dev, dtype), _t(np.array(lens, np.int32), dev), ops.make_ptr_table(dk),
                                               ops.make_ptr_table(dv), None, scale, max(lens), hkv,
                                               valid_lens=_t(np.array(valid, np.int32), dev), bshd=bshd)
    g = oracle.to_f32(_bits(got), dtype).astype(np.float64)
    assert np.isfinite(g).all()
    rel = 5e-3 if dtype else 1e-3                        # bf16 output rounding is 2^-9 relative
    assert np.abs(g - exact).max() < rel * max(1.0, np.abs(exact).max()), np.abs(g - exact).max()


@pytest.mark.parametrize("h,hkv", [(32, 8), (32, 32), (16, 1), (28, 4)])
@pytest.mark.parametrize("bshd", [True, False])
@pytest.mark.parametrize("dtype", [0, 1])
def test_decode_attention_matrix_core_mask_form(oracle, dev, h, hkv, bshd, dtype):
    """the reference's int8 visibility mask (what multi_query_attention_rag_buffer is handed, attention_kernel.cu:1252-1457), one
    query row per task, D = 128: k_decode_attn_mfma's mask form vs the fp64 oracle -- masks with holes (beam hypotheses), prefixes,
    a split without a visible key, a task without one; NaN in K and V of every invisible key must not leak."""
    from zhilight_amd import ops
    rng = np.random.default_rng(34)
    d = 128
    lens = [64, 33, 160, 1088, 1088, 640, 300, 1]
    b = len(lens)
    kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, bshd, dev, dtype, oracle)
    tdt = torch.bfloat16 if dtype else torch.float16
    masks = []
    for bi, L in enumerate(lens):
        if bi == 3:
            m = (np.arange(L) < 1025).astype(np.int8)            # a decode row's prefix
        elif bi == 4:
            m = (rng.random(L) < 0.5).astype(np.int8)
            m[:300] = 0                                          # the first splits see nothing
            m[700] = 1
        elif bi == 6:
            m = np.zeros(L, np.int8)                             # nothing visible: zeros (Z = 1e-20)
        else:
            m = (rng.random(L) < 0.7).astype(np.int8)
            m[0] = 1
        masks.append(m)
        nan = np.uint16(0x7fc0 if dtype else 0x7e00)
        pk, pv = kb[bi].copy(), vb[bi].copy()
        if bshd:
            pk[m == 0] = nan
            pv[m == 0] = nan
        else:
            pk[:, m == 0] = nan
            pv[:, m == 0] = nan
        dk[bi].copy_(_t(pk.view(np.int16), dev, tdt))
        dv[bi].copy_(_t(pv.view(np.int16), dev, tdt))
    mask = np.concatenate(masks)
    q = _to_bits(rng.standard_normal((b, 1, h, d)), dtype, oracle)
    scale = 1.0 / np.sqrt(d)
    exact = oracle.mqa_rag_buffer(q, np.array(lens, np.int32), kb, vb, mask, hkv, scale, bshd, dtype=dtype, exact=True)
    got = ops.multi_query_attention_rag_buffer(_tt(q, dev, dtype), _t(np.array(lens, np.int32), dev), ops.make_ptr_table(dk),
                                               ops.make_ptr_table(dv), _t(mask, dev), scale, max(lens), hkv, bshd=bshd)
    g = oracle.to_f32(_bits(got), dtype).astype(np.float64)
    assert np.isfinite(g).all()
    assert not g[6].any()
    rel = 5e-3 if dtype else 1e-3                        # bf16 output rounding is 2^-9 relative
    assert np.abs(g - exact).max() < rel * max(1.0, np.abs(exact).max()), np.abs(g - exact).max()


def test_decode_attention_long_split(oracle, dev):
    """L = 32768 exercises many splits + combine; checked against the fp64 oracle and against the
    reference's split-KV + combine restatement."""
    from zhilight_amd import ops
    rng = np.random.default_rng(5)
    h, hkv, d = 8, 2, 128
    lens = [32768, 4096]
    kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, True, dev, oracle=oracle)
    q = synth.act(rng, 2 * h, d).reshape(2, 1, h, d)
    mask = np.ones(sum(lens), np.int8)
    exact = oracle.mqa_rag_buffer(oracle.h2u(q), np.array(lens, np.int32), kb, vb, mask, hkv, 0.088, True, exact=True)
    ref_split = oracle.u2h(oracle.mqa_rag_buffer(oracle.h2u(q), np.array(lens, np.int32), kb, vb, mask, hkv, 0.088, True,
                                                 num_split=8)).astype(np.float64)
    got = _np(ops.multi_query_attention_rag_buffer(_t(q, dev), _t(np.array(lens, np.int32), dev), ops.make_ptr_table(dk),
                                                   ops.make_ptr_table(dv), _t(mask, dev), 0.088, max(lens), hkv)).astype(np.float64)
    tol = 1e-3 * max(1.0, np.abs(exact).max())
    assert np.abs(got - exact).max() < tol and np.abs(got - ref_split).max() < 2 * tol


def test_rope_scatter_decode_matches_unfused(oracle, dev):
    from zhilight_amd import ops
    rng = np.random.default_rng(6)
    h, hkv, d, b = 32, 8, 128, 4
    lens = [128, 192, 64, 256]
    kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, True, dev, oracle=oracle)
    pos = np.array([100, 150, 3, 255], np.int32)
    placement = pos.copy()
    qkv = synth.act(rng, b, (h + 2 * hkv) * d)
    cs, sn = oracle.rope_cos_sin(pos, d, 5e5, True, (8.0, 1.0, 4.0, 8192.0))
    rq, rk, rv = oracle.rope_qk_cache(cs, sn, oracle.h2u(qkv), h, hkv, d, True)
    oracle.copy_to_rag_buffer2(placement.reshape(b, 1), np.array(lens, np.int32), rk.reshape(b, 1, hkv, d), rv.reshape(b, 1, hkv, d), kb, vb, True)
    gq = ops.rope_scatter_decode(_t(cs, dev), _t(sn, dev), _t(qkv, dev), _t(placement, dev), _t(np.array(lens, np.int32), dev),
                                 ops.make_ptr_table(dk), ops.make_ptr_table(dv), h, hkv, d)
    assert np.array_equal(_bits(gq), rq)
    for i in range(b):
        assert np.array_equal(_bits(dk[i]), kb[i]) and np.array_equal(_bits(dv[i]), vb[i])


@pytest.mark.parametrize("dtype", [0, 1])
def test_elementwise_and_embedding(oracle, dev, dtype):
    from zhilight_amd import ops
    rng = np.random.default_rng(7)
    n = 14336 * 2 + 3
    a, b_ = _to_bits(rng.standard_normal(n) * 3, dtype, oracle), _to_bits(rng.standard_normal(n) * 3, dtype, oracle)
    for scale, sr in ((1.0, True), (0.3, True), (1.4, False)):
        ref = oracle.element_add_scale(a, b_, scale, sr, dtype)
        got = _bits(ops.element_add_scale(_tt(a, dev, dtype), _tt(b_, dev, dtype), scale, sr))
        assert np.array_equal(got, ref)   # T arithmetic, one rounding per op: bit-exact
    for act, fn in (("silu", oracle.silu_mul), ("gelu", oracle.gelu_mul)):
        ref = fn(a, b_, dtype)
        got = _bits(ops.gate_mul(_tt(a, dev, dtype).clone(), _tt(b_, dev, dtype), act))
        ulp = synth.ulp_diff_f16(got, ref)
        assert ulp.max() <= 1 and (ulp > 0).mean() < 5e-3, (act, ulp.max(), (ulp > 0).mean())  # device expf/tanhf vs glibc
    vocab, dim = 1000, 2304
    w = _to_bits(rng.standard_normal((vocab, dim)), dtype, oracle)
    ids = rng.integers(0, vocab, 17).astype(np.int32)
    for scale, begin, end in ((1.0, 0, vocab), (12.0, 0, vocab), (1.0, 100, 600)):
        wv = w[begin:end] if (begin, end) != (0, vocab) else w
        ref = oracle.embedding(ids, wv, scale, begin, end, dtype)
        got = _bits(ops.embedding(_t(ids, dev), _tt(wv, dev, dtype), scale, begin, end))
        assert np.array_equal(got, ref)   # indexing: bit-exact


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("m", [1, 3, 8])
def test_dense_gemm_small_m(oracle, dev, dtype, m):
    from zhilight_amd import ops
    rng = np.random.default_rng(8)
    n, k = 1000, 2304
    x = _to_bits(rng.standard_normal((m, k)), dtype, oracle)
    w = _to_bits(rng.standard_normal((n, k)) * 0.05, dtype, oracle)
    bias = _to_bits(rng.standard_normal(n), dtype, oracle)
    exact = oracle.gemm_nt(x, w, bias, 0.5, dtype, exact=True)
    got = oracle.to_f32(_bits(ops.gemm_nt_small_m(_tt(x, dev, dtype), _tt(w, dev, dtype), _tt(bias, dev, dtype), 0.5)), dtype)
    rel = 1e-3 if dtype == 0 else 8e-3
    assert np.abs(got - exact).max() < rel * np.abs(exact).max()
    # fused final-norm prologue
    nw = _to_bits(1 + 0.1 * rng.standard_normal(k), dtype, oracle)
    xn = oracle.rmsnorm(x, nw, 1e-5, dtype=dtype)
    exact = oracle.gemm_nt(xn, w, None, 1.0, dtype, exact=True)
    got = oracle.to_f32(_bits(ops.gemm_nt_small_m(_tt(x, dev, dtype), _tt(w, dev, dtype), norm_weight=_tt(nw, dev, dtype), norm_eps=1e-5)), dtype)
    assert np.abs(got - exact).max() < 2 * rel * np.abs(exact).max()


@pytest.mark.parametrize("dtype", [0, 1])
def test_int8_path_bit_exact(oracle, dev, dtype):
    from zhilight_amd import ops
    rng = np.random.default_rng(9)
    m, k, n = 33, 4096, 512
    x = _to_bits(rng.standard_normal((m, k)) * rng.uniform(0.1, 10, (m, 1)), dtype, oracle)
    rq, rs = oracle.quant_calc_scale(x, dtype)
    gq, gs = ops.quant_calc_scale(_tt(x, dev, dtype))
    assert np.array_equal(_np(gq), rq) and np.array_equal(_np(gs), rs)
    w8 = rng.integers(-127, 128, (n, k)).astype(np.int8)
    rc = oracle.int8_gemm_nt(rq, w8)
    gc = ops.int8_gemm_nt(gq, _t(w8, dev))
    assert np.array_equal(_np(gc), rc)
    sy = _to_bits(np.abs(rng.standard_normal(n)) * 0.01, dtype, oracle)
    tdt = torch.bfloat16 if dtype else torch.float16
    assert np.array_equal(_bits(ops.quant_scale_back(gc, gs, _tt(sy, dev, dtype), tdt)), oracle.quant_scale_back(rc, rs, sy, dtype))
    rc2 = oracle.int8_gemm_nt(rq, w8[::-1].copy())
    gc2 = ops.int8_gemm_nt(gq, _t(w8[::-1].copy(), dev))
    ref = oracle.quant_back_act_mul(rc, rs, sy, rc2, rs, sy, "silu", dtype)
    got = _bits(ops.quant_back_act_mul(gc, gs, _tt(sy, dev, dtype), gc2, gs, _tt(sy, dev, dtype), "silu", tdt))
    ulp = synth.ulp_diff_f16(got, ref)
    assert ulp.max() <= 1 and (ulp > 0).mean() < 5e-3
    # fused rmsnorm + quant
    nw = _to_bits(1 + 0.1 * rng.standard_normal(k), dtype, oracle)
    ro, rq2, rs2 = oracle.rmsnorm_quant(x, nw, 1e-5, 1.0, dtype)
    go, gq2, gs2 = ops.layernorm_quant(_tt(x, dev, dtype), _tt(nw, dev, dtype), 1e-5)
    assert np.array_equal(_np(gq2), rq2)            # int8 codes: bit-exact (independent of the rsqrt)
    assert np.allclose(_np(gs2), rs2, rtol=3e-7, atol=0)   # scale carries the block-sum association
    ulp = synth.ulp_diff_f16(_bits(go), ro)
    assert ulp.max() <= 1 and (ulp > 0).mean() < 0.02


@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("bshd", [True, False])
def test_decode_attention_fused_equals_three_kernel_sequence(oracle, dev, neox, bshd):
    """zl_decode_attn_fused == rope_qk_cache + copy_to_rag_buffer2 + multi_query_attention_rag_buffer:
    KV buffers bit-identical, attention output within the attention bar (same rounded q/k/v feed it)."""
    from zhilight_amd import ops
    rng = np.random.default_rng(11)
    h, hkv, d, b = 32, 8, 128, 5
    lens = [1088, 192, 64, 256, 128]
    pos = np.array([1024, 150, 0, 255, 127], np.int32)     # incl. first token and last slot of a buffer
    kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, bshd, dev, oracle=oracle)
    kb2, vb2 = [a.copy() for a in kb], [a.copy() for a in vb]
    dk2, dv2 = [t.clone() for t in dk], [t.clone() for t in dv]
    qkv = synth.act(rng, b, (h + 2 * hkv) * d)
    cs, sn = oracle.rope_cos_sin(pos, d, 5e5, neox, (8.0, 1.0, 4.0, 8192.0))
    lens_np, valid = np.array(lens, np.int32), (pos + 1).astype(np.int32)
    # oracle: three steps
    rq, rk, rv = oracle.rope_qk_cache(cs, sn, oracle.h2u(qkv), h, hkv, d, neox)
    oracle.copy_to_rag_buffer2(pos.reshape(b, 1), lens_np, rk.reshape(b, 1, hkv, d), rv.reshape(b, 1, hkv, d), kb, vb, bshd)
    mask = np.concatenate([(np.arange(L) < v).astype(np.int8) for L, v in zip(lens, valid)])
    exact = oracle.mqa_rag_buffer(rq.reshape(b, 1, h, d), lens_np, kb, vb, mask, hkv, 0.088, bshd, exact=True).reshape(b, -1)
    # device: fused
    got = ops.decode_attention_fused(_t(cs, dev), _t(sn, dev), _t(qkv, dev), _t(pos, dev), _t(lens_np, dev), _t(valid, dev),
                                     ops.make_ptr_table(dk), ops.make_ptr_table(dv), h, hkv, d, 0.088, max(lens), neox, bshd)
    for i in range(b):
        assert np.array_equal(_bits(dk[i]), kb[i]) and np.array_equal(_bits(dv[i]), vb[i])
    g = _np(got).astype(np.float64)
    assert np.abs(g - exact).max() < 1e-3 * max(1.0, np.abs(exact).max())
    # device: the unfused sequence on fresh copies (its attention is the matrix-core kernel for this shape: other
    # arithmetic, same bar)
    gq = ops.rope_scatter_decode(_t(cs, dev), _t(sn, dev), _t(qkv, dev), _t(pos, dev), _t(lens_np, dev),
                                 ops.make_ptr_table(dk2), ops.make_ptr_table(dv2), h, hkv, d, neox, bshd)
    ref = ops.multi_query_attention_rag_buffer(gq.view(b, 1, h, d), _t(lens_np, dev), ops.make_ptr_table(dk2),
                                               ops.make_ptr_table(dv2), None, 0.088, max(lens), hkv,
                                               valid_lens=_t(valid, dev), bshd=bshd)
    for i in range(b):
        assert np.array_equal(_bits(dk2[i]), kb[i]) and np.array_equal(_bits(dv2[i]), vb[i])
    assert np.abs(_np(ref.view(b, -1)).astype(np.float64) - exact).max() < 1e-3 * max(1.0, np.abs(exact).max())


@pytest.mark.parametrize("dtype", [0, 1])
def test_int8_scale_back_variants_bit_exact(oracle, dev, dtype):
    """quant_scale_back3 / quant_back_element_add_scale / quant_back_transpose / quant_back_copy_to_buffer
    (src/nn/quant/int8/quant_kernel.cu:311-583): same single-rounding arithmetic as quant_scale_back."""
    from zhilight_amd import ops
    rng = np.random.default_rng(21)
    b, t, h, hk, d = 2, 5, 6, 2, 64
    m, dim_q, dim_kv = b * t, h * d, hk * d
    n = dim_q + 2 * dim_kv
    c = rng.integers(-2 ** 20, 2 ** 20, (m, n)).astype(np.int32)
    sx = np.abs(rng.standard_normal(m)).astype(np.float32) * 0.01 + 1e-4
    sy = _to_bits(np.abs(rng.standard_normal(n)) * 0.01 + 1e-4, dtype, oracle)
    gq, gk, gv = ops.quant_scale_back3(_t(c, dev), _t(sx, dev), _tt(sy, dev, dtype), dim_q, dim_kv)
    rq, rk, rv = oracle.quant_scale_back3(c, sx, sy, dim_q, dim_kv, dtype)
    assert np.array_equal(_bits(gq), rq) and np.array_equal(_bits(gk), rk) and np.array_equal(_bits(gv), rv)
    with pytest.raises(ops.ZLError):
        ops.quant_scale_back3(_t(c, dev), _t(sx, dev), _tt(sy, dev, dtype), dim_q, dim_kv + 1)

    badd = _to_bits(rng.standard_normal((m, n)), dtype, oracle)
    got = ops.quant_back_element_add_scale(_t(c, dev), _t(sx, dev), _tt(sy, dev, dtype), _tt(badd, dev, dtype), 0.5)
    assert np.array_equal(_bits(got), oracle.quant_back_element_add_scale(c, sx, sy, badd, 0.5, dtype))

    c4 = c[:, :dim_q].copy().reshape(b, t, h, d)
    syq = sy[:dim_q].copy()
    got = ops.quant_back_transpose(_t(c4, dev), _t(sx.reshape(b, t), dev), _tt(syq, dev, dtype))
    assert np.array_equal(_bits(got), oracle.quant_back_transpose(c4, sx, syq, dtype))

    len_buf = 16
    place = np.stack([rng.permutation(len_buf)[:t] for _ in range(b)]).astype(np.int32)
    place[1, 2] = -1                                   # padded row: must be left untouched
    init = _to_bits(rng.standard_normal((b, h, len_buf, d)), dtype, oracle)
    ref = oracle.quant_back_copy_to_buffer(c4, sx, syq, place, init.copy(), dtype)
    dst = _tt(init.copy(), dev, dtype)
    ops.quant_back_copy_to_buffer(_t(c4, dev), _t(sx.reshape(b, t), dev), _tt(syq, dev, dtype), _t(place, dev), dst)
    assert np.array_equal(_bits(dst), ref)
    # 3-d form without placement
    ref3 = oracle.quant_back_copy_to_buffer(c4[:1], sx[:t], syq, None, init[:1].copy(), dtype)[0]
    dst3 = _tt(init[0].copy(), dev, dtype)
    ops.quant_back_copy_to_buffer(_t(c4[0], dev), _t(sx[:t], dev), _tt(syq, dev, dtype), None, dst3)
    assert np.array_equal(_bits(dst3), ref3)


@pytest.mark.parametrize("s_q,pos0,h,hkv,bshd", [(70, 0, 8, 2, True), (200, 37, 4, 4, True), (64, 0, 4, 1, False), (1, 5, 8, 2, True),
                                                 (129, 300, 8, 8, True)])
def test_prefill_attention(oracle, dev, s_q, pos0, h, hkv, bshd):
    """zl_prefill_attn (causal MFMA attention of one task's chunk) vs the oracle's exact attention with a causal
    mask; probabilities are rounded to fp16 before P.V like in flash attention -> ~1e-3 of the output scale."""
    from zhilight_amd import ops
    rng = np.random.default_rng(s_q + pos0)
    d = 128
    len_buf = (pos0 + s_q + 63) // 64 * 64 + 64
    q = (rng.standard_normal((s_q, h, d)) * 1.5).astype(np.float16)
    kb = rng.standard_normal((len_buf, hkv, d)).astype(np.float16)
    vb = rng.standard_normal((len_buf, hkv, d)).astype(np.float16)
    if not bshd:
        kb, vb = np.ascontiguousarray(kb.transpose(1, 0, 2)), np.ascontiguousarray(vb.transpose(1, 0, 2))
    mask = (np.arange(len_buf)[None, :] <= (pos0 + np.arange(s_q))[:, None]).astype(np.int8)
    ref = oracle.mqa_rag_buffer(oracle.h2u(q)[None], np.array([len_buf], np.int32), [oracle.h2u(kb)], [oracle.h2u(vb)], mask, hkv,
                                1.0 / np.sqrt(d), bshd, exact=True)[0]
    got = _np(ops.prefill_attention(_t(q, dev), _t(kb, dev), _t(vb, dev), pos0, hkv, 1.0 / np.sqrt(d), bshd)).astype(np.float64)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
    # bf16: the same kernel on v_mfma_f32_16x16x32_bf16; probabilities rounded to bf16 (2^-9): 1.5e-2 of the output scale
    qb, kbb, vbb = (oracle.f32_to_bf16(a.astype(np.float32)) for a in (q, kb, vb))
    refb = oracle.mqa_rag_buffer(qb[None], np.array([len_buf], np.int32), [kbb], [vbb], mask, hkv, 1.0 / np.sqrt(d), bshd, dtype=1,
                                 exact=True)[0]
    gotb = ops.prefill_attention(_tt(qb, dev, 1), _tt(kbb, dev, 1), _tt(vbb, dev, 1), pos0, hkv, 1.0 / np.sqrt(d), bshd)
    gb = oracle.to_f32(_bits(gotb), 1).astype(np.float64)
    assert np.isfinite(gb).all()
    assert np.abs(gb - refb).max() <= 1.5e-2 * np.abs(refb).max(), np.abs(gb - refb).max() / np.abs(refb).max()


@pytest.mark.parametrize("groups", [1, 2, 4])
@pytest.mark.parametrize("s_q,pos0,h,hkv", [(1024, 0, 8, 2), (300, 517, 4, 4), (65, 0, 4, 1), (130, 64, 8, 8), (1024, 0, 32, 8)])
def test_prefill_attention_wave_groups(oracle, dev, s_q, pos0, h, hkv, groups):
    """zl_prefill_attn_ex: the key tiles of a query tile dealt to 1 / 2 / 4 wave groups, and the workgroup -> work item maps of the
    launcher (longest first; long / short pairs when the whole launch is resident: the 32-head 1 024-token case) -- every form
    against the oracle's exact causal attention (same bar as before), incl. query tiles with fewer key tiles than groups, a ragged
    last query tile, a chunk that starts inside the buffer, NaN behind the visible keys; and the launcher's own choice (groups = 0)."""
    from zhilight_amd import ops
    rng = np.random.default_rng(7 * s_q + pos0 + groups)
    d = 128
    len_buf = (pos0 + s_q + 63) // 64 * 64 + 64
    q = (rng.standard_normal((s_q, h, d)) * 1.5).astype(np.float16)
    kb = rng.standard_normal((len_buf, hkv, d)).astype(np.float16)
    vb = rng.standard_normal((len_buf, hkv, d)).astype(np.float16)
    vb_ref = vb.copy()
    vb[pos0 + s_q:] = np.float16(np.nan)                    # never visible: must not leak
    mask = (np.arange(len_buf)[None, :] <= (pos0 + np.arange(s_q))[:, None]).astype(np.int8)
    ref = oracle.mqa_rag_buffer(oracle.h2u(q)[None], np.array([len_buf], np.int32), [oracle.h2u(kb)], [oracle.h2u(vb_ref)], mask, hkv,
                                1.0 / np.sqrt(d), True, exact=True)[0]
    for g in ((groups, 0) if groups == 4 else (groups,)):
        got = _np(ops.prefill_attention(_t(q, dev), _t(kb, dev), _t(vb, dev), pos0, hkv, 1.0 / np.sqrt(d), True, groups=g)).astype(np.float64)
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max(), (g, np.abs(got - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("m,n,k", [(5, 1000, 256), (16, 128, 128), (33, 4100, 512), (100, 300, 1024), (8, 128256, 128)])
def test_dense_gemm_nt_mfma(oracle, dev, dtype, m, n, k):
    """zl_gemm_nt (a21: functions::Gemm NT, fp32 accumulation) vs the exact product, incl. ragged M / N and bias."""
    from zhilight_amd import ops
    rng = np.random.default_rng(m + n)
    x = _to_bits(rng.standard_normal((m, k)), dtype, oracle)
    w = _to_bits(rng.standard_normal((n, k)) * 0.05, dtype, oracle)
    bias = _to_bits(rng.standard_normal(n) * 0.1, dtype, oracle)
    rel = 2.0 ** -8 if dtype else 2.0 ** -11
    for b in (None, bias):
        exact = oracle.gemm_nt(x, w, b, 0.5, dtype, exact=True)
        got = oracle.to_f32(_bits(ops.gemm_nt(_tt(x, dev, dtype), _tt(w, dev, dtype), None if b is None else _tt(b, dev, dtype), 0.5)), dtype)
        assert (np.abs(got - exact) <= 1.01 * rel * np.abs(exact) + 1e-5 * np.abs(exact).max()).all()


@pytest.mark.parametrize("m,n,k", [(1, 128, 256), (5, 1000, 1024), (32, 4096, 4096), (33, 520, 2304), (100, 300, 512),
                                   (7, 64, 272)])
def test_int8_gemm_shapes_bit_exact(oracle, dev, m, n, k):
    """zl_int8_gemm_nt: the tiled kernel (K % 256 == 0; split-K with int32 atomics = exact) and the simple one
    (other K) against the oracle, ragged M / N."""
    from zhilight_amd import ops
    rng = np.random.default_rng(m * 7 + n)
    a = rng.integers(-127, 128, (m, k)).astype(np.int8)
    w = rng.integers(-127, 128, (n, k)).astype(np.int8)
    got = _np(ops.int8_gemm_nt(_t(a, dev), _t(w, dev)))
    assert np.array_equal(got, oracle.int8_gemm_nt(a, w))


@pytest.mark.parametrize("rounds", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("m", [1, 7, 16, 17, 32])
def test_w8a8_streaming_gemm_bit_identical(oracle, dev, m, rounds, monkeypatch):
    """zl_w8a8_gemm_phase (every tiles-per-workgroup x row-block instantiation) == zl_int8_gemm_nt + the scale-back
    launchers, bit for bit, for the three epilogues the Int8 layer uses; ragged N, K with a partial last phase and K
    beyond one unrolled body; the int32 product itself is checked against the oracle."""
    from zhilight_amd import ops
    monkeypatch.setenv("ZL_W8_PHASE_ROUNDS", str(rounds))
    rng = np.random.default_rng(200 + m + rounds)
    for k, n in ((1152, 16 * (2 * rounds) + 8), (4096, 16 * rounds), (9 * 1024 + 256, 16 * rounds + 16)):
        a = rng.integers(-127, 128, (m, k), dtype=np.int8)
        w = rng.integers(-127, 128, (n, k), dtype=np.int8)
        sx = rng.uniform(0.01, 0.05, m).astype(np.float32)
        sy = rng.uniform(0.001, 0.01, n).astype(np.float16)
        add = synth.act(rng, m, n)
        ta, tw, tsx, tsy, tadd = _t(a, dev), _t(w, dev), _t(sx, dev), _t(sy, dev), _t(add, dev)
        c = ops.int8_gemm_nt(ta, tw)
        assert np.array_equal(_np(c), oracle.int8_gemm_nt(a, w))
        w8 = ops.W8MWeight.from_rows(tw, tsy)
        got = ops.w8a8_gemm_phase(ta, tsx, w8, ops.W8_BACK)
        assert torch.equal(got, ops.quant_scale_back(c, tsx, tsy, torch.float16))
        got = ops.w8a8_gemm_phase(ta, tsx, w8, ops.W8_BACK_ADD, addend=tadd, scale=1.0)
        assert torch.equal(got, ops.quant_back_element_add_scale(c, tsx, tsy, tadd, 1.0))
        if n % 2 == 0:
            w8g = ops.W8MWeight.from_rows(tw, tsy, row_interleave=True)
            got = ops.w8a8_gemm_phase(ta, tsx, w8g, ops.W8_ACT_SILU)
            h2 = n // 2
            ref = ops.quant_back_act_mul(c[:, :h2].contiguous(), tsx, tsy[:h2].contiguous(), c[:, h2:].contiguous(), tsx,
                                         tsy[h2:].contiguous(), "silu", torch.float16)
            assert torch.equal(got, ref)


@pytest.mark.parametrize("m", [1, 8, 16, 17, 32])
@pytest.mark.parametrize("bshd", [True, False])
def test_w8a8_fused_qkv_rotary_scatter_bit_identical(oracle, dev, m, bshd):
    """zl_w8a8_qkv_rope_scatter == zl_w8a8_gemm_phase(BACK) + zl_rope_scatter_decode, bit for bit (q, K / V buffers),
    incl. a task whose placement is -1"""
    from zhilight_amd import ops
    rng = np.random.default_rng(400 + m)
    h, hkv, d, k = 8, 2, 128, 1024 + 256
    n = (h + 2 * hkv) * d
    a = _t(rng.integers(-127, 128, (m, k), dtype=np.int8), dev)
    w = _t(rng.integers(-127, 128, (n, k), dtype=np.int8), dev)
    sx = _t(rng.uniform(0.002, 0.01, m).astype(np.float32), dev)
    sy = _t(rng.uniform(0.001, 0.004, n).astype(np.float16), dev)
    w8 = ops.W8MWeight.from_rows(w, sy)
    lens = [int(v) for v in rng.integers(2, 6, m) * 32]
    pos = np.array([int(rng.integers(0, L)) for L in lens], np.int32)
    placement = pos.copy()
    if m > 1:
        placement[1] = -1
    cs, sn = oracle.rope_cos_sin(pos, d, 5e5, True, (8.0, 1.0, 4.0, 8192.0))
    shape = (lambda L: (L, hkv, d)) if bshd else (lambda L: (hkv, L, d))
    mk = lambda: [torch.full(shape(L), 3.0, dtype=torch.float16, device=dev) for L in lens]
    k1, v1, k2, v2 = mk(), mk(), mk(), mk()
    lens_t, place_t = _t(np.array(lens, np.int32), dev), _t(placement, dev)
    qkv = ops.w8a8_gemm_phase(a, sx, w8, ops.W8_BACK)
    q_ref = ops.rope_scatter_decode(_t(cs, dev), _t(sn, dev), qkv, place_t, lens_t, ops.make_ptr_table(k1), ops.make_ptr_table(v1),
                                    h, hkv, d, True, bshd)
    q_got = ops.w8a8_qkv_rope_scatter(a, sx, w8, _t(cs, dev), _t(sn, dev), place_t, lens_t, ops.make_ptr_table(k2),
                                      ops.make_ptr_table(v2), h, hkv, d, bshd=bshd)
    assert torch.equal(q_got, q_ref)
    for x1, x2 in zip(k1 + v1, k2 + v2):
        assert torch.equal(x1, x2)
    assert not torch.equal(k1[0], torch.full_like(k1[0], 3.0))


@pytest.mark.parametrize("bshd", [True, False])
def test_kv_scatter_drops_slots_outside_the_buffer(oracle, dev, bshd):
    """a placement >= len_buf (one decode step too many: the device-side bookkeeping bumps it without a bound)
    must not write anywhere -- every scatter flavour (copy_to_rag_buffer2, the fused rope + scatter, the qkv-GEMV epilogue,
    the INT8-cache quantising scatter) leaves all buffers untouched; the reference asserts pos_buf < len_buf
    (ragged_buffer_kernel.cu:194-222)."""
    from zhilight_amd import ops
    rng = np.random.default_rng(21)
    h, hkv, d = 8, 2, 128
    lens = [64, 128]
    b = len(lens)
    # the two tasks' buffers are carved from ONE allocation with a guard tensor right behind: a stray write shows there
    def carve():
        tot = sum(L * hkv * d for L in lens)
        blob = torch.zeros(tot + 4096, dtype=torch.float16, device=dev)
        outs, off = [], 0
        for L in lens:
            outs.append(blob[off:off + L * hkv * d].view((L, hkv, d) if bshd else (hkv, L, d)))
            off += L * hkv * d
        return blob, outs
    kblob, dk = carve()
    vblob, dv = carve()
    place = torch.tensor([lens[0], lens[1] + 5], dtype=torch.int32, device=dev)          # both past the end
    lens_t = _t(np.array(lens, np.int32), dev)
    ks, vs = _t(synth.act(rng, b, hkv * d).reshape(b, 1, hkv, d), dev), _t(synth.act(rng, b, hkv * d).reshape(b, 1, hkv, d), dev)
    kt, vt = ops.make_ptr_table(dk), ops.make_ptr_table(dv)
    ops.copy_to_rag_buffer2(place.view(b, 1), lens_t, ks, vs, kt, vt, bshd)
    pos = np.array([3, 4], np.int32)
    cs, sn = oracle.rope_cos_sin(pos, d, 1e4, True)
    qkv = _t(synth.act(rng, b, (h + 2 * hkv) * d), dev)
    ops.rope_scatter_decode(_t(cs, dev), _t(sn, dev), qkv, place, lens_t, kt, vt, h, hkv, d, bshd=bshd)
    w = ops.W4MWeight.random((h + 2 * hkv) * d, 1024, 128, dev)
    x = _t(synth.act(rng, b, 1024), dev)
    ops.w4_qkv_rope_scatter(x, w, _t(cs, dev), _t(sn, dev), place, lens_t, kt, vt, h, hkv, d, bshd=bshd)
    torch.cuda.synchronize()
    assert not kblob.any() and not vblob.any()


def test_decode_attention_single_kv_head_very_long_buffer(oracle, dev):
    """batch 1 with ONE local kv head (the ATTN_KV_REP_TP geometry) and a 131072-slot buffer wanted 1024 splits
    of 128 keys; the merge holds 512 -> keys beyond 65536 were dropped silently.  The split length now grows to fit."""
    from zhilight_amd import ops
    rng = np.random.default_rng(22)
    h, hkv, d, L = 4, 1, 128, 131072
    assert ops.decode_attn_split_len(1, hkv, L) * 512 >= L
    q = synth.act(rng, h, d).reshape(1, 1, h, d)
    kb = (rng.standard_normal((L, hkv, d)) * 0.5).astype(np.float16)
    vb = np.zeros((L, hkv, d), np.float16)
    vb[L - 1000:] = 1.0                                     # only the LAST keys carry value mass: dropping them shows
    kb[L - 1000:] = kb[L - 1000:] + (q[0, 0, 0] * 0.5).astype(np.float16)   # ... and they attract the attention
    dk, dv = [_t(kb, dev)], [_t(vb, dev)]
    valid = torch.tensor([L], dtype=torch.int32, device=dev)
    out = ops.multi_query_attention_rag_buffer(_t(q, dev), _t(np.array([L], np.int32), dev), ops.make_ptr_table(dk), ops.make_ptr_table(dv),
                                               None, 1.0 / np.sqrt(d), L, hkv, valid_lens=valid)
    got = _np(out).astype(np.float64)
    mask = np.ones(L, np.int8)
    ref = oracle.mqa_rag_buffer(oracle.h2u(q), np.array([L], np.int32), [oracle.h2u(kb)], [oracle.h2u(vb)], mask, hkv, 1.0 / np.sqrt(d),
                                True, exact=True)
    assert np.abs(got - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-3)
    assert ref[0, 0, 0].mean() > 0.5                         # the tail really dominates (else the test proves nothing)


@pytest.mark.gpu
def test_rope_tables_dynamic_ntk_and_yarn(oracle, dev):
    """RotaryEmbedding "dynamic" and YarnImpl angles (rotary_embedding.cu:19-61, 398-553) as cos / sin tables: device vs
    oracle within the angle's fp32 uncertainty (pos x a few ulp of inv_freq), the host-side yarn constants bit-equal."""
    from zhilight_amd import ops
    d = 128
    pos = np.concatenate([np.arange(0, 70), [1024, 4095, 8191, 40000, 131071]]).astype(np.int32)
    bound = 1e-6 + 6 * 2.0 ** -24 * np.maximum(pos[:, None].astype(np.float64), 1.0)
    for theta in (1e4, 1e6):
        for seq in (None, np.full(pos.size, 131071, np.int32), pos[::-1].copy()):
            rc, rs = oracle.rope_cos_sin_dynamic(pos, d, theta, 2.0, 4096.0, seq)
            gc, gs = ops.rope_cos_sin_dynamic(_t(pos, dev), d, theta, 2.0, 4096.0, None if seq is None else _t(seq, dev))
            assert (np.abs(_np(gc) - rc) <= bound).all() and (np.abs(_np(gs) - rs) <= bound).all()
        for deepseek, kw in ((False, {}), (True, dict(mscale=1.0, mscale_all_dim=0.707))):
            ref_p = oracle.yarn_params(theta, d, 4096, 40.0, 32, 1, 1.0, deepseek, **kw)
            got_p = ops.yarn_params(theta, d, 4096, 40.0, 32, 1, 1.0, deepseek, **kw)
            assert ref_p == got_p, (ref_p, got_p)
            rc, rs = oracle.rope_cos_sin_yarn(pos, d, theta, 40.0, *ref_p)
            gc, gs = ops.rope_cos_sin_yarn(_t(pos, dev), d, theta, 40.0, *got_p)
            b2 = bound * max(1.0, ref_p[2])
            assert (np.abs(_np(gc) - rc) <= b2).all() and (np.abs(_np(gs) - rs) <= b2).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("mode", [0, 1])
def test_head_norm_matches_oracle(oracle, dev, mode, dtype):
    """q_norm / k_norm over dim_head (mode 0: Qwen3 RMSNorm, one weight; mode 1: KERNEL_layernorm_multi_head), in place on a
    column window of a wider fused-projection row."""
    from zhilight_amd import ops
    rng = np.random.default_rng(5 + mode)
    tdt = torch.float16 if dtype == 0 else torch.bfloat16
    for rows, heads, d in ((1, 32, 128), (7, 8, 64), (33, 2, 256)):
        wide = torch.from_numpy(rng.standard_normal((rows, heads * d + 2 * d)).astype(np.float32) * 2 + 0.3).to(dev).to(tdt)
        w = torch.from_numpy(1 + 0.2 * rng.standard_normal(d if mode == 0 else heads * d).astype(np.float32)).to(dev).to(tdt)
        x = wide[:, d:d + heads * d]
        xb = x.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
        ref = oracle.head_norm(xb, w.view(torch.int16).cpu().numpy().view(np.uint16), heads, d, 1e-6, mode, dtype)
        keep = wide.clone()
        out = ops.head_norm(x, w, heads, d, 1e-6, mode)
        got = out.view(torch.int16).cpu().numpy().view(np.uint16)
        ulp = synth.ulp_diff_f16(got, ref) if dtype == 0 else np.abs(got.astype(np.int64) - ref.astype(np.int64))
        assert ulp.max() <= 1 and (ulp > 0).mean() < 0.02, (rows, heads, d, ulp.max(), (ulp > 0).mean())
        ops.head_norm(x, w, heads, d, 1e-6, mode, out=x)       # in place on the window; the columns around it untouched
        assert torch.equal(x.contiguous().view(torch.int16), out.view(torch.int16))
        assert torch.equal(wide[:, :d], keep[:, :d]) and torch.equal(wide[:, d + heads * d:], keep[:, d + heads * d:])


@pytest.mark.gpu
def test_caller_supplied_outputs_are_checked(dev):
    """a wrong `out` (shape / dtype / non-contiguous) is refused by the host wrappers instead of being overrun by the launcher"""
    from zhilight_amd import ops
    from zhilight_amd._lib import ZLError
    w = ops.W4MWeight.random(256, 1024, 128, dev)
    x = torch.randn(3, 1024, device=dev).half()
    for bad in (torch.empty(3, 128, dtype=torch.float16, device=dev), torch.empty(3, 256, dtype=torch.float32, device=dev),
                torch.empty(3, 512, dtype=torch.float16, device=dev)[:, ::2]):
        with pytest.raises(ZLError):
            ops.w4a16_gemm_mfma(x, w, out=bad)
        with pytest.raises(ZLError):
            ops.w4a16_gemm_tiled(x, w, out=bad)
    dw = torch.randn(256, 1024, device=dev).half()
    with pytest.raises(ZLError):
        ops.gemm_nt(x, dw, out=torch.empty(3, 255, dtype=torch.float16, device=dev))
    with pytest.raises(ZLError):
        ops.gemm_nt_small_m(x, dw, out=torch.empty(2, 256, dtype=torch.float16, device=dev))
    ok = torch.empty(3, 256, dtype=torch.float16, device=dev)
    assert ops.w4a16_gemm_mfma(x, w, out=ok) is ok and ops.gemm_nt(x, dw, out=ok) is ok


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_int8_compressed_reduce_kernels_bit_exact(oracle, dev, world, dtype):
    """int8_op::quant_group_32 / dequant_sum_quant_g32 / dequant_group_32 (quant_reduce_kernel.cu:13-105, 270-330, 107-150): codes
    and T scales bit for bit against the oracle's restatement, incl. a group of zeros and magnitudes that round up in T."""
    from zhilight_amd import ops
    rng = np.random.default_rng(90 + world + dtype)
    groups = 8 * world * 5 + 3
    x = (rng.standard_normal((groups, 32)) * rng.uniform(0.01, 30.0, (groups, 1))).astype(np.float32)
    x[1] = 0.0
    xb = _to_bits(x, dtype, oracle)
    q_ref, s_ref = oracle.quant_group_32(xb, dtype)
    q, s = ops.quant_group_32(_tt(xb, dev, dtype))
    assert np.array_equal(_np(q), q_ref) and np.array_equal(_bits(s), s_ref)
    qo = rng.integers(-127, 128, (world - 1, groups, 32), dtype=np.int8)
    so = _to_bits(rng.uniform(1e-3, 0.3, (world - 1, groups)), dtype, oracle)
    qs_ref, ss_ref = oracle.dequant_sum_quant_g32(xb, qo, so, dtype)
    qs, ss = ops.dequant_sum_quant_g32(_tt(xb, dev, dtype), _t(qo, dev), _tt(so, dev, dtype))
    assert np.array_equal(_np(qs), qs_ref) and np.array_equal(_bits(ss), ss_ref)
    out = ops.dequant_group_32(qs, ss)
    assert np.array_equal(_bits(out), oracle.dequant_group_32(qs_ref, ss_ref, dtype))


@pytest.mark.parametrize("world", [2, 4])
def test_reduce_tp_int8_composition(oracle, dev, world):
    """DirectTPGroup.reduce_tp_int8 (ModelContext::reduce_tp_int8, model_context.cpp:244-326) with the RCCL send / recv rounds
    replaced by an in-process mailbox (all ranks on this one GPU, run in lock step): every rank ends with the oracle's
    composition of the reference's steps, bit for bit, and within int8 noise of the exact sum."""
    import threading
    from zhilight_amd.parallel import DirectTPGroup
    rng = np.random.default_rng(5 + world)
    rows, n = 4, 32 * world * 6
    parts = [_to_bits(rng.standard_normal((rows, n)), 0, oracle) for _ in range(world)]
    want = oracle.reduce_tp_int8([p.reshape(-1) for p in parts], 0).reshape(rows, n)
    box, lock, bar = {}, threading.Lock(), threading.Barrier(world)

    class MailComm:
        def __init__(self, rank):
            self.rank, self.pending = rank, []

        def send(self, t, peer):
            with lock:
                box.setdefault((self.rank, peer), []).append(t.clone())

        def recv(self, t, peer):
            self.pending.append((t, peer))

        def group_start(self):
            self.pending = []

        def group_end(self):
            torch.cuda.synchronize()
            bar.wait()                                   # every send of this round is posted
            for t, peer in self.pending:
                with lock:
                    t.copy_(box[(peer, self.rank)].pop(0))
            torch.cuda.synchronize()
            bar.wait()

    outs, errs = [None] * world, []

    def run(r):
        try:
            g = DirectTPGroup.__new__(DirectTPGroup)
            g.rank, g.size, g.comm = r, world, MailComm(r)
            outs[r] = _bits(g.reduce_tp_int8(_tt(parts[r], dev, 0)))
        except Exception as e:       # pragma: no cover
            errs.append(e)
            bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs
    for r in range(world):
        assert np.array_equal(outs[r], want), r
    exact = sum(oracle.u2h(p).astype(np.float64) for p in parts)
    assert np.abs(oracle.u2h(want).astype(np.float64) - exact).max() <= 0.02 * np.abs(exact).max()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("llama3", [None, (8.0, 1.0, 4.0, 8192)])
def test_embedding_rope_one_launch_equals_the_two_calls(dev, dtype, llama3):
    """zl_embedding_rope = zl_embedding + zl_rope_cos_sin(_llama3), the first two launches of a decode step, bit for bit"""
    from zhilight_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    vocab, dim, d = 1000, 4096, 128
    w = (torch.randn(vocab, dim, generator=g) * 0.5).to(dtype).to(dev)
    for b in (1, 7, 32):
        ids = torch.randint(0, vocab, (b,), generator=g, dtype=torch.int32).to(dev)
        pos = torch.randint(0, 9000, (b,), generator=g, dtype=torch.int32).to(dev)
        h, cs, sn = ops.embedding_rope(ids, w, 12.0, pos, d, 5e5, True, llama3)
        assert torch.equal(h, ops.embedding(ids, w, 12.0))
        cs2, sn2 = ops.rope_cos_sin(pos, d, 5e5, True, llama3)
        assert torch.equal(cs, cs2) and torch.equal(sn, sn2)
    # odd widths keep the element-wise path of the gather
    w2 = (torch.randn(50, 100, generator=g)).to(dtype).to(dev)
    ids = torch.tensor([3, 49, 0], dtype=torch.int32, device=dev)
    assert torch.equal(ops.embedding(ids, w2, 1.0), w2[ids.long()])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,n,k", [(5, 1000, 1024), (8, 128256 // 8, 4096), (16, 272, 128), (17, 2048 + 16, 2048), (32, 4096, 4096)])
def test_gemm_nt_on_the_packed_dense_layout(dev, dtype, m, n, k):
    """ZLD16M (zl_dense_pack_m + zl_gemm_nt_packed): the lm_head of a 5..32-row decode batch read in 1 KiB contiguous fragment loads --
    the same arithmetic in the same order as zl_gemm_nt on the row-major matrix, so the same bits; ragged N (a partial last tile, a
    partial last workgroup), bias, one and two row blocks"""
    from zhilight_amd import ops
    g = torch.Generator(device=dev).manual_seed(m * 7 + n)
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).to(dtype)
    x = torch.randn(m, k, generator=g, device=dev).to(dtype)
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).to(dtype)
    wp = ops.DenseMWeight(w)
    assert torch.equal(ops.gemm_nt_packed(x, wp), ops.gemm_nt(x, w))
    assert torch.equal(ops.gemm_nt_packed(x, wp, bias=bias, alpha=0.5), ops.gemm_nt(x, w, bias=bias, alpha=0.5))

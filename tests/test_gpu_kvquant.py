"""GPU parity for the INT8 KV cache (quantised variant): the write side is bit-exact
(codes u8 and fp32 scales), the attention is checked against the fp64 oracle at 1e-3 of max|out|."""
import numpy as np
import pytest
import torch

import synth
from test_gpu_ops import _bits, _mask_for, _np, _t, _to_bits, _tt

pytestmark = pytest.mark.gpu


def _empty_cache(lens, hkv, d, bshd, dev, rng=None):
    """per-task u8 code buffers + fp32 scale buffers (host mirrors and device tensors)"""
    kc, vc, ks, vs = [], [], [], []
    for L in lens:
        cshape, sshape = ((L, hkv, d), (L, hkv)) if bshd else ((hkv, L, d), (hkv, L))
        if rng is None:
            kc.append(np.full(cshape, 7, np.uint8)); vc.append(np.full(cshape, 9, np.uint8))
            ks.append(np.full(sshape, -1.0, np.float32)); vs.append(np.full(sshape, -2.0, np.float32))
        else:
            kc.append(rng.integers(0, 256, cshape, dtype=np.uint8)); vc.append(rng.integers(0, 256, cshape, dtype=np.uint8))
            ks.append(rng.uniform(0.005, 0.04, sshape).astype(np.float32)); vs.append(rng.uniform(0.005, 0.04, sshape).astype(np.float32))
    dev_of = lambda arrs: [_t(a, dev) for a in arrs]
    return (kc, vc, ks, vs), (dev_of(kc), dev_of(vc), dev_of(ks), dev_of(vs))


def _scatter_host(host, b, place, hk, bshd, kq, ksc, vq, vsc):
    kc, vc, ks, vs = host
    idx = (place, hk) if bshd else (hk, place)
    kc[b][idx] = kq; vc[b][idx] = vq
    ks[b][idx] = ksc; vs[b][idx] = vsc


@pytest.mark.parametrize("dtype", [0, 1])
def test_quant_calc_scale_zp_bit_exact(oracle, dev, dtype):
    from zhilight_amd import ops
    rng = np.random.default_rng(40)
    x = rng.standard_normal((37, 128)) * rng.uniform(0.01, 30, (37, 1))
    x[5] = 0.0                                   # all-zero row: codes 128, scale 0 (documented)
    xb = _to_bits(x, dtype, oracle)
    q, s = ops.quant_calc_scale_zp(_tt(xb, dev, dtype))
    rq, rs = oracle.quant_calc_scale_zp(xb, 128, dtype)
    assert np.array_equal(_np(q), rq) and np.array_equal(_np(s), rs)


@pytest.mark.parametrize("bshd", [True, False])
@pytest.mark.parametrize("dtype", [0, 1])
def test_quant_copy_to_rag_buffer_bit_exact(oracle, dev, bshd, dtype):
    from zhilight_amd import ops
    rng = np.random.default_rng(41)
    hkv, d, len_q = 8, 128, 3
    lens = [64, 128, 192, 65]
    b = len(lens)
    host, devt = _empty_cache(lens, hkv, d, bshd, dev)
    placement = np.stack([rng.choice(L, len_q, replace=False) for L in lens]).astype(np.int32)
    placement[1, 1] = -1
    ksrc = _to_bits(rng.standard_normal((b * len_q, hkv, d)) * 3, dtype, oracle)
    vsrc = _to_bits(rng.standard_normal((b * len_q, hkv, d)), dtype, oracle)
    kq, ksc = oracle.quant_calc_scale_zp(ksrc.reshape(-1, d), 128, dtype)
    vq, vsc = oracle.quant_calc_scale_zp(vsrc.reshape(-1, d), 128, dtype)
    for bi in range(b):
        for qi in range(len_q):
            if placement[bi, qi] < 0:
                continue
            for hk in range(hkv):
                r = ((bi * len_q) + qi) * hkv + hk
                _scatter_host(host, bi, placement[bi, qi], hk, bshd, kq[r], ksc[r], vq[r], vsc[r])
    ops.quant_copy_to_rag_buffer(_t(placement.reshape(-1), dev), _t(np.array(lens, np.int32), dev), _tt(ksrc, dev, dtype),
                                 _tt(vsrc, dev, dtype), *[ops.make_ptr_table(x) for x in devt], len_q=len_q, bshd=bshd)
    for bi in range(b):
        for h_arr, d_arr in zip(host, devt):
            assert np.array_equal(_np(d_arr[bi]), h_arr[bi])


@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("bshd", [True, False])
def test_rope_quant_scatter_decode_bit_exact(oracle, dev, neox, bshd):
    from zhilight_amd import ops
    rng = np.random.default_rng(42)
    h, hkv, d, b = 32, 8, 128, 4
    lens = [128, 192, 64, 256]
    host, devt = _empty_cache(lens, hkv, d, bshd, dev)
    pos = np.array([100, 150, 3, 255], np.int32)
    placement = pos.copy()
    placement[2] = -1
    qkv = synth.act(rng, b, (h + 2 * hkv) * d)
    cs, sn = oracle.rope_cos_sin(pos, d, 5e5, neox, (8.0, 1.0, 4.0, 8192.0))
    rq, rk, rv = oracle.rope_qk_cache(cs, sn, oracle.h2u(qkv), h, hkv, d, neox)
    kq, ksc = oracle.quant_calc_scale_zp(rk.reshape(-1, d), 128, 0)
    vq, vsc = oracle.quant_calc_scale_zp(rv.reshape(-1, d), 128, 0)
    for bi in range(b):
        if placement[bi] < 0:
            continue
        for hk in range(hkv):
            r = bi * hkv + hk
            _scatter_host(host, bi, placement[bi], hk, bshd, kq[r], ksc[r], vq[r], vsc[r])
    gq = ops.rope_quant_scatter_decode(_t(cs, dev), _t(sn, dev), _t(qkv, dev), _t(placement, dev),
                                       _t(np.array(lens, np.int32), dev), *[ops.make_ptr_table(x) for x in devt], h, hkv, d,
                                       neox=neox, bshd=bshd)
    assert np.array_equal(_bits(gq), rq)
    for bi in range(b):
        for h_arr, d_arr in zip(host, devt):
            assert np.array_equal(_np(d_arr[bi]), h_arr[bi])


@pytest.mark.parametrize("h,hkv,d,len_q", [(32, 8, 128, 1), (32, 32, 128, 1), (16, 2, 64, 1), (8, 2, 128, 4), (4, 4, 256, 2),
                                           (28, 4, 128, 1)])
@pytest.mark.parametrize("bshd", [True, False])
def test_decode_attention_quant(oracle, dev, h, hkv, d, len_q, bshd):
    """fp64 statement of KERNEL_mqa_rag_buffer_split_kv_quant; bar 1e-3 of max|out| (the kernel accumulates in
    fp32; the reference forms q.(K-128) in fp16, which is noisier than this bar -- quant_attention.cuh:39-75)"""
    from zhilight_amd import ops
    rng = np.random.default_rng(43)
    lens = [1, 63, 64, 130, 1088, 517]
    b = len(lens)
    host, devt = _empty_cache(lens, hkv, d, bshd, dev, rng)
    q = synth.act(rng, b * len_q * h, d).reshape(b, len_q, h, d)
    scale = 1.0 / np.sqrt(d)
    lens_np = np.array(lens, np.int32)
    for mode in ("prefix", "random"):
        mask = _mask_for(lens, len_q, rng, mode)
        exact = oracle.mqa_rag_buffer_quant(oracle.h2u(q), lens_np, *host, mask, hkv, scale, bshd)
        got = ops.multi_query_attention_rag_buffer_quant(_t(q, dev), _t(lens_np, dev), *[ops.make_ptr_table(x) for x in devt],
                                                         _t(mask, dev), scale, max(lens), hkv, bshd=bshd)
        g = _np(got).astype(np.float64)
        assert np.isfinite(g).all()
        tol = 1e-3 * max(1.0, np.abs(exact).max())
        assert np.abs(g - exact).max() < tol, np.abs(g - exact).max()


def test_decode_attention_quant_valid_lens_bf16_and_garbage_tail(oracle, dev):
    """prefix visibility through valid_lens; slots past the visible prefix hold NaN scales (never-written
    memory) and must not leak"""
    from zhilight_amd import ops
    rng = np.random.default_rng(44)
    h, hkv, d = 32, 8, 128
    lens = [1088, 1088, 256]
    valid = [1025, 7, 256]
    lens_np = np.array(lens, np.int32)
    for dtype in (0, 1):
        host, _ = _empty_cache(lens, hkv, d, True, dev, rng)
        for bi, v in enumerate(valid):
            host[2][bi][v:] = np.nan
            host[3][bi][v:] = np.inf
        devt = [[_t(a, dev) for a in arrs] for arrs in host]
        q = _to_bits(rng.standard_normal((3, 1, h, d)), dtype, oracle)
        mask = np.concatenate([(np.arange(L) < v).astype(np.int8) for L, v in zip(lens, valid)])
        exact = oracle.mqa_rag_buffer_quant(q, lens_np, *host, mask, hkv, 0.088, True, dtype=dtype)
        got = ops.multi_query_attention_rag_buffer_quant(_tt(q, dev, dtype), _t(lens_np, dev),
                                                         *[ops.make_ptr_table(x) for x in devt], None, 0.088, max(lens), hkv,
                                                         valid_lens=_t(np.array(valid, np.int32), dev))
        g = oracle.to_f32(_bits(got), dtype).astype(np.float64)
        assert np.isfinite(g).all()
        rel = 1e-3 if dtype == 0 else 5e-3
        assert np.abs(g - exact).max() < rel * max(1.0, np.abs(exact).max())
        # the mask form over the same visibility (other split count: not bit-identical)
        got_m = ops.multi_query_attention_rag_buffer_quant(_tt(q, dev, dtype), _t(lens_np, dev),
                                                           *[ops.make_ptr_table(x) for x in devt], _t(mask, dev), 0.088,
                                                           max(lens), hkv)
        gm = oracle.to_f32(_bits(got_m), dtype).astype(np.float64)
        assert np.isfinite(gm).all() and np.abs(gm - exact).max() < rel * max(1.0, np.abs(exact).max())


@pytest.mark.parametrize("h,hkv,len_q", [(32, 8, 1), (32, 32, 1), (16, 1, 1), (8, 2, 2), (16, 4, 4), (28, 4, 1)])
@pytest.mark.parametrize("bshd", [True, False])
def test_decode_attention_quant_matrix_core_path(oracle, dev, h, hkv, len_q, bshd):
    """prefix visibility, fp16, D = 128, <= 16 query rows per kv head: k_decode_attn_mfma_q8 vs the fp64 oracle; ragged
    lengths around the 32-key chunk and 128-key split boundaries; NaN scales / arbitrary codes past the visible prefix"""
    from zhilight_amd import ops
    rng = np.random.default_rng(47)
    d = 128
    lens = [64, 64, 64, 160, 160, 160, 1088, 1088, 640]
    valid = [1, 31, 33, 127, 128, 129, 1025, 517, 640]
    b = len(lens)
    host, _ = _empty_cache(lens, hkv, d, bshd, dev, rng)
    dev_host = [[a.copy() for a in arrs] for arrs in host]
    for bi, v in enumerate(valid):
        for arr, poison in ((dev_host[2][bi], np.nan), (dev_host[3][bi], np.inf)):
            if bshd:
                arr[v:] = poison
            else:
                arr[:, v:] = poison
    devt = [[_t(a, dev) for a in arrs] for arrs in dev_host]
    q = synth.act(rng, b * len_q * h, d).reshape(b, len_q, h, d)
    scale = 1.0 / np.sqrt(d)
    lens_np = np.array(lens, np.int32)
    mask = np.concatenate([np.tile((np.arange(L) < v).astype(np.int8), len_q) for L, v in zip(lens, valid)])
    exact = oracle.mqa_rag_buffer_quant(oracle.h2u(q), lens_np, *host, mask, hkv, scale, bshd)
    got = ops.multi_query_attention_rag_buffer_quant(_t(q, dev), _t(lens_np, dev), *[ops.make_ptr_table(x) for x in devt], None,
                                                     scale, max(lens), hkv, valid_lens=_t(np.array(valid, np.int32), dev), bshd=bshd)
    g = _np(got).astype(np.float64)
    assert np.isfinite(g).all()
    assert np.abs(g - exact).max() < 1e-3 * max(1.0, np.abs(exact).max()), np.abs(g - exact).max()


def test_decode_attention_quant_long_split(oracle, dev):
    from zhilight_amd import ops
    rng = np.random.default_rng(45)
    h, hkv, d = 8, 2, 128
    lens = [32768, 4096]
    lens_np = np.array(lens, np.int32)
    host, devt = _empty_cache(lens, hkv, d, True, dev, rng)
    q = synth.act(rng, 2 * h, d).reshape(2, 1, h, d)
    mask = np.ones(sum(lens), np.int8)
    exact = oracle.mqa_rag_buffer_quant(oracle.h2u(q), lens_np, *host, mask, hkv, 0.088, True)
    got = _np(ops.multi_query_attention_rag_buffer_quant(_t(q, dev), _t(lens_np, dev), *[ops.make_ptr_table(x) for x in devt],
                                                         _t(mask, dev), 0.088, max(lens), hkv)).astype(np.float64)
    assert np.abs(got - exact).max() < 1e-3 * max(1.0, np.abs(exact).max())


def test_quantised_cache_tracks_fp16_cache(oracle, dev):
    """end to end: quantise real K/V rows into the cache, attend; the result stays within the quantisation
    noise of the fp16-cache attention (8-bit codes: about 1e-2 of max|out| for Gaussian rows)"""
    from zhilight_amd import ops
    rng = np.random.default_rng(46)
    h, hkv, d, L = 32, 8, 128, 700
    k = synth.act(rng, L * hkv, d).reshape(L, hkv, d)
    v = synth.act(rng, L * hkv, d).reshape(L, hkv, d)
    q = synth.act(rng, h, d).reshape(1, 1, h, d)
    lens = _t(np.array([L], np.int32), dev)
    _, devt = _empty_cache([L], hkv, d, True, dev)
    ops.quant_copy_to_rag_buffer(_t(np.arange(L, dtype=np.int32), dev), lens, _t(k, dev), _t(v, dev),
                                 *[ops.make_ptr_table(x) for x in devt], len_q=L)
    valid = _t(np.array([L], np.int32), dev)
    got = _np(ops.multi_query_attention_rag_buffer_quant(_t(q, dev), lens, *[ops.make_ptr_table(x) for x in devt], None, 0.088,
                                                         L, hkv, valid_lens=valid)).astype(np.float64)
    dk, dv = [_t(k, dev)], [_t(v, dev)]
    ref = _np(ops.multi_query_attention_rag_buffer(_t(q, dev), lens, ops.make_ptr_table(dk), ops.make_ptr_table(dv), None, 0.088,
                                                   L, hkv, valid_lens=valid)).astype(np.float64)
    assert np.abs(got - ref).max() < 2e-2 * np.abs(ref).max()


def test_matrix_core_attention_random_shapes(oracle, dev):
    """24 seeded random draws (tasks, kv heads, GQA ratio, rows per kv head <= 16, ragged buffer / visible lengths, layout,
    dtype, fp16 cache or INT8 cache) through the matrix-core decode attention kernels vs the fp64 oracles"""
    from zhilight_amd import ops
    from test_gpu_ops import _make_kv
    rng = np.random.default_rng(777)
    d = 128
    for case in range(24):
        hkv = int(rng.choice([1, 2, 4, 8]))
        n_rep = int(rng.choice([1, 2, 4, 8]))
        len_q = int(rng.choice([1, 2])) if n_rep <= 8 else 1
        h = hkv * n_rep
        b = int(rng.integers(1, 6))
        lens = [int(v) * 64 for v in rng.integers(1, 12, b)]
        valid = [int(rng.integers(1, L + 1)) for L in lens]
        bshd = bool(rng.integers(0, 2))
        quant = bool(rng.integers(0, 2))
        dtype = 0 if quant else int(rng.integers(0, 2))
        lens_np, valid_np = np.array(lens, np.int32), np.array(valid, np.int32)
        mask = np.concatenate([np.tile((np.arange(L) < v).astype(np.int8), len_q) for L, v in zip(lens, valid)])
        q = _to_bits(rng.standard_normal((b, len_q, h, d)), dtype, oracle)
        scale = 1.0 / np.sqrt(d)
        if quant:
            host, devt = _empty_cache(lens, hkv, d, bshd, dev, rng)
            exact = oracle.mqa_rag_buffer_quant(q, lens_np, *host, mask, hkv, scale, bshd)
            got = ops.multi_query_attention_rag_buffer_quant(_tt(q, dev, dtype), _t(lens_np, dev), *[ops.make_ptr_table(x) for x in devt],
                                                             None, scale, max(lens), hkv, valid_lens=_t(valid_np, dev), bshd=bshd)
        else:
            kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, bshd, dev, dtype, oracle)
            exact = oracle.mqa_rag_buffer(q, lens_np, kb, vb, mask, hkv, scale, bshd, dtype=dtype, exact=True)
            got = ops.multi_query_attention_rag_buffer(_tt(q, dev, dtype), _t(lens_np, dev), ops.make_ptr_table(dk),
                                                       ops.make_ptr_table(dv), None, scale, max(lens), hkv,
                                                       valid_lens=_t(valid_np, dev), bshd=bshd)
        g = oracle.to_f32(_bits(got), dtype).astype(np.float64)
        rel = 5e-3 if dtype else 1e-3
        assert np.isfinite(g).all(), case
        assert np.abs(g - exact).max() < rel * max(1.0, np.abs(exact).max()), (case, hkv, n_rep, len_q, b, lens, valid, bshd, quant, dtype)

"""CPU self-consistency of the oracle: format definitions, R vs E gap, integer exactness, soft-float."""
import numpy as np
import pytest

import synth


def test_soft_float_conversions(oracle):
    rng = np.random.default_rng(0)
    f = (rng.standard_normal(20000) * 10.0 ** rng.integers(-8, 5, 20000)).astype(np.float32)
    mine = np.array([oracle.lib().zlo_f32_to_f16(float(v)) for v in f], np.uint16)
    assert np.array_equal(mine, f.astype(np.float16).view(np.uint16))
    for special in (0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e9, 5.96e-8, 2.98e-8, 2.9802322387695312e-08, 6.1e-5):
        assert oracle.lib().zlo_f64_to_f16(special) == int(np.float64(special).astype(np.float16).view(np.uint16))
    import torch
    bf = oracle.f32_to_bf16(f)
    assert np.array_equal(bf.view(np.int16), torch.from_numpy(f).to(torch.bfloat16).view(torch.int16).numpy())


@pytest.mark.parametrize("k,n,g", [(1024, 256, 128), (2048, 64, 64), (1024, 32, 32)])
def test_k_major_transform_equals_format_definition(oracle, k, n, g):
    rng = np.random.default_rng(1)
    qw, qz, sc = synth.gptq_hf(rng, k, n, g)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, g)
    w16 = oracle.u2h(oracle.gptq_dequant_k_major(*km))
    naive = oracle.gptq_dequant_hf_naive(qw, qz, sc, g).astype(np.float16)
    assert np.array_equal(w16, naive)
    # exact GEMM equals x @ W^T of the format definition
    x = synth.act(rng, 3, k)
    yE = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km)
    assert np.allclose(yE, x.astype(np.float64) @ oracle.gptq_dequant_hf_naive(qw, qz, sc, g).T, rtol=1e-12, atol=1e-12)
    # the reference-faithful flavour is within its fp16-partial-dot noise of the exact one
    yR = oracle.u2h(oracle.gptq_gemm_k_major(oracle.h2u(x), *km)).astype(np.float64)
    rel = np.abs(yR - yE).max() / np.sqrt((yE ** 2).mean())
    assert rel < 6e-3, rel


def test_increase_zero_wraps_and_shuffle_is_a_permutation(oracle):
    allz = np.arange(0, 2 ** 16, dtype=np.uint32)
    allz = (allz | (allz << 16)).reshape(256, 256).astype(np.uint32)
    out = oracle.gptq_increase_zero(allz)
    for j in range(8):
        a, b = (allz >> (4 * j)) & 0xF, (out >> (4 * j)) & 0xF
        assert np.array_equal(b, (a + 1) % 16)
    rng = np.random.default_rng(2)
    q = rng.integers(0, 2 ** 32, size=(16, 8), dtype=np.uint64).astype(np.uint32)
    s = oracle.gptq_shuffle(q)
    for j in range(4):
        assert np.array_equal((s >> (4 * j)) & 0xF, (q >> (8 * j)) & 0xF)              # even weights, low half
        assert np.array_equal((s >> (4 * j + 16)) & 0xF, (q >> (8 * j + 4)) & 0xF)     # odd weights, high half


def test_awq_transforms_roundtrip(oracle):
    rng = np.random.default_rng(3)
    k, n = 64, 32
    nib = rng.integers(0, 16, size=(k, n), dtype=np.uint32)
    order = [0, 4, 1, 5, 2, 6, 3, 7]                      # AWQ nibble order inside a word (utils.cu:33)
    awq = np.zeros((k, n // 8), np.uint32)
    for pos, col in enumerate(order):
        # nibble at bit position 4*col holds logical column `pos`... AWQ: logical column j sits at position order^-1
        pass
    # un_shuffle maps nibble positions de = [0,4,1,5,2,6,3,7] -> 0..7
    for s, src in enumerate(order):
        awq |= nib[:, s::8] << np.uint32(4 * src)
    nat = oracle.awq_un_shuffle(awq)
    for s in range(8):
        assert np.array_equal((nat >> (4 * s)) & 0xF, nib[:, s::8])
    g = oracle.awq_shuffle(awq, use_exllama=False)          # (K/8, N): nibble s of word = row 8*kb + s
    for s in range(8):
        assert np.array_equal((g >> (4 * s)) & 0xF, nib[s::8, :])
    ge = oracle.awq_shuffle(awq, use_exllama=True)
    assert np.array_equal(ge, oracle.gptq_shuffle(g))


def test_int8_gemm_is_exact_and_quant_rule(oracle):
    rng = np.random.default_rng(4)
    a = rng.integers(-127, 128, (5, 256)).astype(np.int8)
    b = rng.integers(-127, 128, (7, 256)).astype(np.int8)
    assert np.array_equal(oracle.int8_gemm_nt(a, b), a.astype(np.int64) @ b.astype(np.int64).T)
    x = synth.act(rng, 4, 512, 3.0)
    q, s = oracle.quant_calc_scale(oracle.h2u(x))
    amax = np.abs(x.astype(np.float32)).max(axis=1)
    assert np.array_equal(s, amax / np.float32(127.0))
    assert np.array_equal(q, np.rint(x.astype(np.float32) * (np.float32(127.0) / amax)[:, None]).astype(np.int8))
    assert np.abs(q).max() == 127


def test_attention_oracle_flavours_agree(oracle):
    rng = np.random.default_rng(5)
    h, hkv, d = 8, 2, 128
    lens = [700, 33]
    kb = [oracle.h2u(rng.standard_normal((L, hkv, d)).astype(np.float16)) for L in lens]
    vb = [oracle.h2u(rng.standard_normal((L, hkv, d)).astype(np.float16)) for L in lens]
    q = oracle.h2u(rng.standard_normal((2, 1, h, d)).astype(np.float16))
    mask = np.concatenate([(rng.random(L) < 0.8).astype(np.int8) for L in lens])
    mask[0] = mask[700] = 1
    a = oracle.u2h(oracle.mqa_rag_buffer(q, np.array(lens, np.int32), kb, vb, mask, hkv, 0.088)).astype(np.float64)
    b = oracle.u2h(oracle.mqa_rag_buffer(q, np.array(lens, np.int32), kb, vb, mask, hkv, 0.088, num_split=4)).astype(np.float64)
    e = oracle.mqa_rag_buffer(q, np.array(lens, np.int32), kb, vb, mask, hkv, 0.088, exact=True)
    assert np.abs(a - e).max() < 2e-3 and np.abs(b - e).max() < 2e-3


def test_attention_oracle_splits_and_tasks_without_a_visible_key(oracle):
    """What the mask form of the matrix-core decode kernel is checked against (tests/test_gpu_ops.py::
    test_decode_attention_matrix_core_mask_form, test_gpu_w4.py::test_attention_mask_form_split_records_merge): the restatement of the
    reference's split-KV kernel + KERNEL_mqa_combine (attention_kernel.cu:673-923: running max from -1e20, Z from 1e-20) on masks with
    holes -- a split without any visible key contributes nothing, a task without one yields zeros -- agrees with the unsplit and the
    fp64 flavours."""
    rng = np.random.default_rng(6)
    h, hkv, d = 8, 2, 128
    lens = [700, 300, 64]
    kb = [oracle.h2u(rng.standard_normal((L, hkv, d)).astype(np.float16)) for L in lens]
    vb = [oracle.h2u(rng.standard_normal((L, hkv, d)).astype(np.float16)) for L in lens]
    q = oracle.h2u(rng.standard_normal((3, 1, h, d)).astype(np.float16))
    m0 = (rng.random(700) < 0.5).astype(np.int8)
    m0[:350] = 0                                   # with four splits of 175 keys: the first two see nothing
    m0[699] = 1
    m1 = np.zeros(300, np.int8)                    # a task without a visible key
    m2 = (rng.random(64) < 0.7).astype(np.int8)
    m2[0] = 1
    mask = np.concatenate([m0, m1, m2])
    la = np.array(lens, np.int32)
    a = oracle.u2h(oracle.mqa_rag_buffer(q, la, kb, vb, mask, hkv, 0.088)).astype(np.float64)
    b = oracle.u2h(oracle.mqa_rag_buffer(q, la, kb, vb, mask, hkv, 0.088, num_split=4)).astype(np.float64)
    e = oracle.mqa_rag_buffer(q, la, kb, vb, mask, hkv, 0.088, exact=True)
    assert np.isfinite(a).all() and np.isfinite(b).all() and np.isfinite(e).all()
    assert not a[1].any() and not b[1].any() and not e[1].any()
    assert np.abs(a - e).max() < 2e-3 and np.abs(b - e).max() < 2e-3


def test_awq_oracle_matches_the_format_definition_and_its_own_exact_product():
    """native AWQ restatement: dequantize == (q - z) * s from the integers the AutoAWQ tensors were packed from
    (one fp16 rounding), == the dense matrix the AWQ-as-exllama re-route dequantises (transposed); the split-K result with
    fp16 partials stays within fp16-partial noise of the exact product."""
    import synth
    import zl_oracle as oracle
    rng = np.random.default_rng(4)
    k, n, g = 1056, 192, 32
    qw, qz, sc, q, z = synth.awq_hf(rng, k, n, g)
    w16 = oracle.awq_dequantize(qw, qz, sc, g)
    d = (q.astype(np.int32) - np.repeat(z.astype(np.int32), g, axis=0)).astype(np.float16)
    ref = (d.astype(np.float32) * np.repeat(sc.view(np.float16), g, axis=0).astype(np.float32)).astype(np.float16)
    assert np.array_equal(w16, ref.view(np.uint16))
    tr = lambda a: np.ascontiguousarray(a.T)   # noqa: E731
    km = (tr(oracle.awq_shuffle(qw, True)), tr(oracle.gptq_q4_to_q8(oracle.awq_un_shuffle(qz))), tr(sc))
    assert np.array_equal(oracle.gptq_dequant_k_major(*km), tr(w16))
    x = synth.act(rng, 3, k)
    r = oracle.u2h(oracle.awq_gemm(oracle.h2u(x), w16, 32)).astype(np.float64)
    e = oracle.awq_gemm(oracle.h2u(x), w16, exact=True)
    assert np.abs(r - e).max() <= 2e-3 * np.sqrt((e ** 2).mean())
    # W4A8: codes bounded, scale = amax / 127, round trip within half a code
    w8, s = oracle.w4a8_weight_to_int8(tr(w16))
    wf = oracle.u2h(tr(w16)).astype(np.float32)
    assert np.abs(w8).max() == 127 and np.allclose(s, np.abs(wf).max(axis=1) / 127.0, rtol=1e-6)
    assert np.abs(w8.astype(np.float32) * s[:, None] - wf).max() <= 0.5001 * s.max()


def test_moe_oracle_against_the_definition(oracle):
    """zlo_gptq_moe_up / _down (KERNEL_gemm_moe_up / _down restated) against fp64 sums over the token's LOCAL experts: routed
    ids under expert parallelism (id % world == rank, stored at id / world), shared experts appended with weight 1, ADD_C."""
    rng = np.random.default_rng(21)
    e, n_sh, n, k, g, m, top_k, world, rank = 3, 1, 24, 256, 128, 3, 2, 2, 1

    def experts(rows, cols):
        l = [oracle.gptq_prepare_k_major(*synth.gptq_hf(rng, cols, rows, g), g) for _ in range(e + n_sh)]
        return l, tuple(np.stack([a[i] for a in l]) for i in range(3))
    gl, gs = experts(n, k)
    ul, us = experts(n, k)
    dl, ds = experts(16, k)
    ids = np.array([[1, 4], [3, 5], [0, 2]], np.int32)            # global ids, 6 routed experts over 2 ranks
    wts = rng.random((m, top_k)).astype(np.float32)
    x = synth.act(rng, m, k)
    up = oracle.u2h(oracle.gptq_moe_up(oracle.h2u(x), gs, us, ids, n_sh, e, False, True, world, rank)).astype(np.float64)
    assert up.shape == (m, top_k + n_sh, n)
    a = synth.act(rng, m * (top_k + n_sh), k).reshape(m, top_k + n_sh, k)
    base = synth.act(rng, m, 16)
    dn = oracle.u2h(oracle.gptq_moe_down(oracle.h2u(a), ds, ids, wts, n_sh, e, False, True, world, rank)).astype(np.float64)
    dn_add = oracle.u2h(oracle.gptq_moe_down(oracle.h2u(a), ds, ids, wts, n_sh, e, False, True, world, rank,
                                             add_c=oracle.h2u(base))).astype(np.float64)
    for mm in range(m):
        want_dn = np.zeros(16)
        for t in range(top_k + n_sh):
            if t < top_k:
                gid = int(ids[mm, t])
                local, ex, w = gid % world == rank, gid // world, float(wts[mm, t])
            else:
                local, ex, w = True, e + t - top_k, 1.0
            if not local:
                assert (up[mm, t] == 0).all()
                continue
            a1 = oracle.gptq_gemm_k_major_exact(oracle.h2u(x[mm:mm + 1]), *gl[ex])[0]
            a2 = oracle.gptq_gemm_k_major_exact(oracle.h2u(x[mm:mm + 1]), *ul[ex])[0]
            ref = a1 / (1 + np.exp(-a1)) * a2
            assert np.abs(up[mm, t] - ref).max() <= 3e-3 * max(1.0, np.abs(ref).max())
            want_dn += w * oracle.gptq_gemm_k_major_exact(oracle.h2u(a[mm, t:t + 1]), *dl[ex])[0]
        assert np.abs(dn[mm] - want_dn).max() <= 3e-3 * max(1.0, np.abs(want_dn).max())
        assert np.abs(dn_add[mm] - (want_dn + base[mm].astype(np.float64))).max() <= 4e-3 * max(1.0, np.abs(want_dn).max())


def test_reduce_tp_int8_oracle_properties(oracle):
    """The restated INT8-compressed reduce (model_context.cpp:244-326): within int8 noise of the exact sum, every group's codes
    reach +-127, a zero group stays zero, and the result does not depend on which rank evaluates it (the oracle composes the
    owner's view of every slice)."""
    import numpy as np
    rng = np.random.default_rng(3)
    for ws in (2, 4, 8):
        n = 32 * ws * 5
        parts = [oracle.h2u(rng.standard_normal(n).astype(np.float16)) for _ in range(ws)]
        for p in parts:
            p[:32] = 0
        out = oracle.u2h(oracle.reduce_tp_int8(parts)).astype(np.float64)
        exact = sum(oracle.u2h(p).astype(np.float64) for p in parts)
        assert np.all(out[:32] == 0)
        assert np.abs(out - exact).max() <= (ws + 1) / 127.0 * np.abs(exact).max()
        q, s = oracle.quant_group_32(parts[0])
        assert np.abs(q.reshape(-1, 32)[1:]).max(axis=1).min() == 127


def test_e4m3_cast_against_the_format_definition(oracle):
    """zlo_f32_to_e4m3 (cvt.rn.satfinite.e4m3x2.f16x2 restated) pinned by the OCP E4M3FN definition: for a dense sample of all
    non-negative finite fp16 values the code is the nearest representable value (ties to the even code), saturating at 448; the
    decode table is monotone and holds the format's landmarks."""
    import numpy as np
    codes = np.arange(0, 0x7f, dtype=np.uint8)
    grid = oracle.e4m3_to_f32(codes)
    assert grid[0] == 0 and grid[1] == 2.0 ** -9 and grid[8] == 2.0 ** -6 and grid[0x7e] == 448.0 and np.all(np.diff(grid) > 0)
    assert np.isnan(oracle.e4m3_to_f32(np.array([0x7f], np.uint8))[0])
    h = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float64)[::11]
    got = oracle.f32_to_e4m3(h.astype(np.float32))
    for v, c in zip(h, got):
        if v >= 464:
            best = 0x7e
        else:
            d = np.abs(grid - v)
            j = np.flatnonzero(d == d.min())
            best = j[0] if len(j) == 1 else (j[0] if codes[j[0]] % 2 == 0 else j[1])
        assert best == c, (v, c, best)
    neg = oracle.f32_to_e4m3(np.array([-1.0, -500.0, -0.0], np.float32))
    assert list(neg) == [0x80 | 0x38, 0x80 | 0x7e, 0x80]


# ---- f4, first part: the FP8 block format and the router restatements against independent numpy statements -----------------
def test_fp8_block_format_against_its_definition(oracle):
    rng = np.random.default_rng(2)
    m, k, n = 5, 512, 256
    x = (rng.standard_normal((m, k)) * 3).astype(np.float16)
    codes, sc = oracle.fp8_per_token_cast(x.view(np.uint16), dtype=0)
    xb = x.astype(np.float32).reshape(m, k // 128, 128)
    amax = np.maximum(np.abs(xb).max(axis=2), 1e-4).astype(np.float32)
    assert np.array_equal(sc[:, :m].T, amax / np.float32(448.0))                       # scale = amax / 448 in fp32
    deq = oracle.e4m3_to_f32(codes).reshape(m, k // 128, 128)
    assert np.abs(deq).max() <= 448 and (np.abs(deq).max(axis=2) == 448).all()        # the block maximum lands on the largest code
    assert (np.abs(deq * (amax / 448)[:, :, None] - xb) <= amax[:, :, None] * 2.0 ** -4 * 1.001).all()
    # block GEMM == fp64 evaluation of the definition with numpy
    w8 = rng.integers(0, 0x78, size=(n, k), dtype=np.uint8)
    sw = (np.abs(rng.standard_normal((n // 128, k // 128))) * 0.01 + 1e-3).astype(np.float32)
    out = oracle.fp8_block_gemm(codes, sc, w8, sw, dtype=0).view(np.float16).astype(np.float64)
    a = oracle.e4m3_to_f32(codes).reshape(m, k // 128, 128)
    w = oracle.e4m3_to_f32(w8).reshape(n, k // 128, 128)
    blk = np.einsum("mbk,nbk->mnb", a, w)
    ref = (blk * sc[:, :m].T.astype(np.float64)[:, None, :] * np.repeat(sw, 128, axis=0).astype(np.float64)[None, :, :]).sum(axis=2)
    assert np.array_equal(out, ref.astype(np.float16).astype(np.float64))
    # dequant: float(code) * scale rounded once
    d = oracle.fp8_block_dequant(w8, sw, dtype=0).view(np.float16)
    assert np.array_equal(d, (oracle.e4m3_to_f32(w8).astype(np.float32) * np.repeat(np.repeat(sw, 128, axis=0), 128, axis=1)).astype(np.float16))


def test_router_restatements_against_numpy(oracle):
    rng = np.random.default_rng(4)
    tokens, e, k = 23, 128, 8
    logits = (rng.standard_normal((tokens, e)) * 1.5).astype(np.float16)
    v, idx, wl, el = oracle.moe_top_k_softmax(logits.view(np.uint16), k, k, True, 1.0, "softmax", 0, 4)
    p = np.exp(logits.astype(np.float64) - logits.astype(np.float64).max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    want = np.argsort(-p, axis=1, kind="stable")[:, :k]
    assert np.array_equal(idx, want)
    pw = np.take_along_axis(p, want, axis=1)
    assert np.allclose(v, pw / pw.sum(axis=1, keepdims=True), rtol=2e-6)
    assert el.sum() == tokens * k and np.array_equal(np.bincount(want.ravel() % 4, minlength=4), wl)
    # the "sigmoid" of top_k_softmax is 1 / (1 + exp(+x)) in the reference (ff_kernel.cu:158-160): the SMALLEST logits win
    v2, idx2, _, _ = oracle.moe_top_k_softmax(logits.view(np.uint16), k, k, False, 1.0, "sigmoid", 0, 0)
    assert np.array_equal(idx2, np.argsort(logits.astype(np.float64), axis=1, kind="stable")[:, :k])
    # group-limited routing (DeepSeek-V3 shape): groups by their best biased score, weights un-biased and renormalised
    e, g, tg, k = 256, 8, 4, 8
    logits = (rng.standard_normal((tokens, e)) * 1.2).astype(np.float16)
    bias = (rng.standard_normal(e) * 0.1).astype(np.float32)
    v, idx, _, el = oracle.moe_group_topk(logits.view(np.uint16), bias, k, g, tg, k, True, 2.5, "sigmoid", 0, 0)
    s = 1.0 / (1.0 + np.exp(-logits.astype(np.float64)))
    sb = s + bias
    for t in range(tokens):
        gs = sb[t].reshape(g, e // g).max(axis=1)
        keep = np.argsort(-gs, kind="stable")[:tg]
        masked = np.full(e, -np.inf)
        for gg in keep:
            masked[gg * (e // g):(gg + 1) * (e // g)] = sb[t, gg * (e // g):(gg + 1) * (e // g)]
        want = np.argsort(-masked, kind="stable")[:k]
        assert np.array_equal(idx[t], want), t
        w = s[t, want]
        assert np.allclose(v[t], w / w.sum() * 2.5, rtol=2e-6)


def test_sampling_restatements_self_consistent(oracle):
    """the numpy restatements of the batch generator's logit post-processing (oracle/zl_oracle.py: what tests/test_gpu_zz_sampling.py holds
    csrc/sampling_ops.hip to): probabilities sum to one, the bias is additive, temperature 0 is the plain form, top-k is descending with ties
    to the lower index, the penalties follow beam_util.cu:199-222"""
    rng = np.random.default_rng(12)
    x = rng.standard_normal((3, 200)) * 3.0
    bias = np.array([0.0, 1.5, -2.0])
    a = oracle.log_softmax_bias_ref(x, bias, 0.7)
    assert np.allclose(np.exp(a - bias[:, None]).sum(axis=1), 1.0, rtol=1e-12)
    assert np.allclose(oracle.log_softmax_bias_ref(x, bias, 0.0), oracle.log_softmax_bias_ref(x, bias, 1.0), rtol=0, atol=1e-12)
    assert np.allclose(oracle.softmax_rows_ref(x, 0.7), np.exp(oracle.log_softmax_bias_ref(x, np.zeros(3), 0.7)), rtol=1e-12)
    y = x.copy()
    y[:, 50] = y[:, 120]                                     # a tie: the lower index first
    v, i = oracle.topk_rows_ref(y, 200)
    assert (np.diff(v, axis=1) <= 0).all() and i.dtype == np.int32
    for r in range(3):
        pos = list(i[r])
        assert pos.index(50) + 1 == pos.index(120) and sorted(pos) == list(range(200))
    rnd = lambda t: np.asarray(t, np.float64).astype(np.float16).astype(np.float64)
    lg = rnd(rng.standard_normal((2, 16)))
    lg[0, 3], lg[1, 5], lg[1, 7] = 2.0, -2.0, 1.0
    out = oracle.repetition_penalty_ref(lg, rnd([1.25, 1.25, 1.25]), rnd([0.0, 0.0, 0.5]), [3, 5, 7], [0, 1, 1], rnd)
    assert out[0, 3] == rnd(2.0 / 1.25) and out[1, 5] == rnd(-2.0 * 1.25) and out[1, 7] == 0.5       # divide positives, multiply negatives, presence subtracts
    untouched = np.ones_like(lg, bool)
    untouched[0, 3] = untouched[1, 5] = untouched[1, 7] = False
    assert np.array_equal(out[untouched], lg[untouched])

"""GPU parity THROUGH the C++ boundary: Python -> zl_internals (pybind) -> nn:: / int8_op:: wrappers with the
reference's signatures (zhilight_amd/hostcpp/nn_amd.cpp) on the bmengine-on-HIP shim -> C ABI -> HIP kernels, checked
against the CPU oracle.  (The other test_gpu_* files drive the same C ABI through ctypes.)  Mirrors what the reference
tests through `zhilight.internals_` (tests/py_export_internal/)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cx(dev):
    from zhilight_amd import _lib
    _lib.lib()
    from zhilight_amd import zl_internals
    return zl_internals.Context(0)


def test_tensor_surface_and_pool(cx):
    a = np.arange(7 * 5, dtype=np.int32).reshape(7, 5)
    got = cx.tensor_roundtrip(a, 2, 6)
    assert np.array_equal(got, a[2:6].reshape(-1))
    before = cx.used_memory()
    for _ in range(3):                                   # blocks are recycled: the pool does not grow
        cx.tensor_roundtrip(a, 0, 7)
    assert cx.used_memory() == before and cx.peak_memory() >= before


@pytest.mark.parametrize("prepack", [True, False])
@pytest.mark.parametrize("m,k,n", [(1, 4096, 6144), (8, 2048, 272), (33, 1024, 64)])
def test_gptq_gemm_k_major_through_cpp(cx, oracle, m, k, n, prepack):
    """nn::gptq::gptq_gemm_k_major (gptq.h:82-95) with the operands pre-packed by amd_pack_k_major (the load-time path)
    and with the raw k-major operands of an unmodified caller; + bias."""
    rng = np.random.default_rng(m + k)
    qw, qz, sc = synth.gptq_hf(rng, k, n, 128)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, 128)
    x = synth.act(rng, m, k)
    bias = (rng.standard_normal(n) * 0.1).astype(np.float16)
    got = cx.gptq_gemm_k_major(x, km[0], km[1], km[2].view(np.float16), bias=bias, prepack=prepack).astype(np.float64)
    exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km, bias=oracle.h2u(bias))
    if m > 32:    # the M-tiled kernel multiplies with W16 = rn16(rn16(q - z) s), the reference's M > 40 arithmetic
        exact = oracle.gemm_nt(oracle.h2u(x), oracle.gptq_dequant_k_major(*km), oracle.h2u(bias), exact=True)
    rms = np.sqrt((exact ** 2).mean())
    assert (np.abs(got - exact) <= 2.0 ** -10 * np.abs(exact) + 3e-5 * rms).all()


def test_gptq_small_group_takes_the_bit_exact_kernel(cx, oracle):
    """group 32 does not fit the matrix-core tile: the wrapper routes to the warp-reduce arithmetic kernel, whose result
    is bit-identical to the R oracle (KERNEL_gemm_warp_reduce, q_gemm_k_major.cu:127-237)."""
    rng = np.random.default_rng(5)
    qw, qz, sc = synth.gptq_hf(rng, 1024, 40, 32)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, 32)
    x = synth.act(rng, 3, 1024)
    got = cx.gptq_gemm_k_major(x, km[0], km[1], km[2].view(np.float16), prepack=False)
    ref = oracle.gptq_gemm_k_major(oracle.h2u(x), *km)
    assert np.array_equal(got.view(np.uint16), ref)


def test_gptq_sym_and_dequant_and_gate(cx, oracle):
    rng = np.random.default_rng(6)
    k, n = 1024, 128
    qw, qz, sc = synth.gptq_hf(rng, k, n, 128, sym=True)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, 128)
    x = synth.act(rng, 2, k)
    got = cx.gptq_gemm_k_major(x, km[0], np.zeros_like(km[1]), km[2].view(np.float16), sym=True, prepack=False).astype(np.float64)
    exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km, sym=True)
    assert np.abs(got - exact).max() <= 2.0 ** -10 * np.abs(exact).max()
    # dequant_k_major: bit-exact W16 = rn16(rn16(q - z) s), raw and packed operands
    w16 = oracle.gptq_dequant_k_major(*km)
    for prepack in (False, True):
        assert np.array_equal(cx.gptq_dequant_k_major(km[0], km[1], km[2].view(np.float16), prepack).view(np.uint16), w16)
    # gemm_fuse_gate_in = silu(x W1^T) * (x W2^T)
    qw2, qz2, sc2 = synth.gptq_hf(rng, k, n, 128)
    km2 = oracle.gptq_prepare_k_major(qw2, qz2, sc2, 128)
    qw1, qz1, sc1 = synth.gptq_hf(rng, k, n, 128)
    km1 = oracle.gptq_prepare_k_major(qw1, qz1, sc1, 128)
    got = cx.gemm_fuse_gate_in(x, km1[0], km1[1], km1[2].view(np.float16), km2[0], km2[1], km2[2].view(np.float16)).astype(np.float64)
    g = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km1).astype(np.float16)
    u = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km2).astype(np.float16)
    ref = oracle.u2h(oracle.silu_mul(oracle.h2u(g), oracle.h2u(u))).astype(np.float64)
    assert np.abs(got - ref).max() <= 2.0 ** -8 * np.abs(ref).max()


def test_awq_and_w4a8_through_cpp(cx, oracle):
    """nn::awq::awq_gemm / awq_dequantize (awq.h:10-25) and the W4A8 int8 branch of gptq_gemm_k_major with precomputed_w8"""
    rng = np.random.default_rng(8)
    k, n, g = 1024, 1280, 128
    qw, qz, sc, _, _ = synth.awq_hf(rng, k, n, g)
    w16 = oracle.awq_dequantize(qw, qz, sc, g)
    assert np.array_equal(cx.awq_dequantize(qw, qz, sc.view(np.float16)).view(np.uint16), w16)
    x = synth.act(rng, 5, k)
    got = cx.awq_gemm(x, qw, qz, sc.view(np.float16), 32).astype(np.float64)
    exact = oracle.awq_gemm(oracle.h2u(x), w16, exact=True)
    assert np.abs(got - exact).max() <= 1e-3 * np.abs(exact).max()
    gw, gz, gs = synth.gptq_hf(rng, k, n, g)
    km = oracle.gptq_prepare_k_major(gw, gz, gs, g)
    xb = synth.act(rng, 48, k, 2.0)
    y, w8, ws = cx.gptq_w4a8(xb, km[0], km[1], km[2].view(np.float16))
    rw8, rs = oracle.w4a8_weight_to_int8(oracle.gptq_dequant_k_major(*km))
    assert np.array_equal(w8, rw8) and np.array_equal(ws, rs)
    rq, rsx = oracle.quant_calc_scale(oracle.h2u(xb))
    assert np.array_equal(y.view(np.uint16), oracle.quant_scale_back_f32(oracle.int8_gemm_nt(rq, rw8), rsx, rs))


@pytest.mark.parametrize("act_order", [False, True])
def test_gptq_legacy_route_reconstruct_and_alt_gemm(cx, oracle, act_order):
    """SURVEY 8a row a6, the part without exllama (a row-parallel act-order shard under GPTQ_KERNEL_ALGO=0): nn::gptq::reconstruct_gptq
    bit-exact against its restatement (q_gemm.cu:641-676), and nn::gptq::gptq_gemm(use_exllama = false) -- checkpoint-order words,
    raw g_idx -- against the exact product (the reference's alt kernel adds fp16 partials atomically: no bit pattern to match)."""
    rng = np.random.default_rng(61 + act_order)
    k, n, g = 1024, 264, 128
    qw, qz, sc, g_idx, w16 = synth.gptq_act_order_hf(rng, k, n, g)
    if not act_order:
        g_idx = (np.arange(k) // g).astype(np.int32)
    zp1 = oracle.gptq_increase_zero(qz)
    got = cx.gptq_reconstruct(qw.view(np.int32), zp1.view(np.int32), sc.view(np.float16), g_idx)
    assert np.array_equal(got.view(np.uint16), oracle.gptq_reconstruct(qw, zp1, sc, g_idx))
    if act_order:                                        # ... which is the matrix the k-major dequantiser produces for the same checkpoint
        assert np.array_equal(got.view(np.uint16), np.ascontiguousarray(w16.T).view(np.uint16))
    for m in (1, 6, 40):
        x = synth.act(rng, m, k)
        y = cx.gptq_gemm_legacy(x, qw.view(np.int32), zp1.view(np.int32), sc.view(np.float16), g_idx, False, g).astype(np.float64)
        want = oracle.gptq_gemm_legacy_exact(oracle.h2u(x), qw, zp1, sc, g_idx)
        rms = np.sqrt((want ** 2).mean())
        # the dense GEMM multiplies with the fp16-rounded weights (reconstruct_gptq + cuBLAS in the reference above 8 rows)
        assert np.abs(y - want).max() <= 2.0 ** -10 * np.abs(want).max() + 2e-3 * rms, (m, np.abs(y - want).max() / rms)


def test_gptq_load_transforms_bit_exact(cx, oracle):
    rng = np.random.default_rng(7)
    qw, qz, sc = synth.gptq_hf(rng, 1024, 256, 128)
    q, z8 = cx.gptq_load_transforms(qw, qz)
    assert np.array_equal(q, oracle.gptq_shuffle(qw))
    assert np.array_equal(z8.reshape(-1), oracle.gptq_q4_to_q8(oracle.gptq_increase_zero(qz)).reshape(-1))


@pytest.mark.parametrize("b,len_q,lens", [(1, 1, [1088]), (3, 1, [64, 200, 130]), (2, 4, [96, 70])])
def test_multi_query_attention_rag_buffer_through_cpp(cx, oracle, b, len_q, lens):
    rng = np.random.default_rng(b + len_q)
    h, hkv, d = 32, 8, 128
    q = synth.act(rng, b * len_q * h, d).reshape(b, len_q, h, d)
    kb = [synth.act(rng, L * hkv, d).reshape(L, hkv, d) for L in lens]
    vb = [synth.act(rng, L * hkv, d).reshape(L, hkv, d) for L in lens]
    mask = np.concatenate([(rng.random((len_q, L)) < 0.8).astype(np.int8).reshape(-1) for L in lens])
    scale = 1.0 / np.sqrt(d)
    got = cx.multi_query_attention_rag_buffer(q, np.asarray(lens, np.int32), kb, vb, mask, scale, max(lens), h // hkv).astype(np.float64)
    ref = oracle.mqa_rag_buffer(oracle.h2u(q), np.asarray(lens, np.int32), [oracle.h2u(a) for a in kb], [oracle.h2u(a) for a in vb], mask,
                                hkv, scale, True, exact=True)
    assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max()


def test_attention_qkv_rag_buffer_is_the_one_head_per_kv_head_form(cx, oracle):
    """nn::attention_qkv_rag_buffer (attention_kernel.h:39-50): every query head owns its kv head -- the same launcher with m_query = 1,
    checked against the oracle and, bit for bit, against that call"""
    rng = np.random.default_rng(31)
    b, len_q, h, d, lens = 2, 2, 16, 128, [96, 70]
    q = synth.act(rng, b * len_q * h, d).reshape(b, len_q, h, d)
    kb = [synth.act(rng, L * h, d).reshape(L, h, d) for L in lens]
    vb = [synth.act(rng, L * h, d).reshape(L, h, d) for L in lens]
    mask = np.concatenate([(rng.random((len_q, L)) < 0.8).astype(np.int8).reshape(-1) for L in lens])
    scale = 1.0 / np.sqrt(d)
    got = cx.attention_qkv_rag_buffer(q, np.asarray(lens, np.int32), kb, vb, mask, scale, max(lens))
    same = cx.multi_query_attention_rag_buffer(q, np.asarray(lens, np.int32), kb, vb, mask, scale, max(lens), 1)
    assert np.array_equal(got.view(np.uint16), same.view(np.uint16))
    ref = oracle.mqa_rag_buffer(oracle.h2u(q), np.asarray(lens, np.int32), [oracle.h2u(a) for a in kb], [oracle.h2u(a) for a in vb], mask,
                                h, scale, True, exact=True)
    assert np.abs(got.astype(np.float64) - ref).max() <= 1e-3 * np.abs(ref).max()


def test_rope_scatter_norm_elementwise_through_cpp(cx, oracle):
    rng = np.random.default_rng(11)
    h, hkv, d, s = 8, 2, 128, 5
    pos = np.array([0, 3, 17, 1024, 8191], np.int32)
    cs, sn = oracle.rope_cos_sin(pos, d, 5e5, True)
    x = synth.act(rng, s, (h + 2 * hkv) * d)
    q, k, v = cx.rope_qk_cache(cs, sn, x, h, hkv, d, True)
    rq, rk, rv = oracle.rope_qk_cache(cs, sn, oracle.h2u(x), h, hkv, d, True)
    for got, ref in ((q, rq), (k, rk), (v, rv)):
        assert synth.ulp_diff_f16(got.view(np.uint16), ref).max() <= 1
    # scatter two tasks' rows
    lens = np.array([16, 24], np.int32)
    place = np.array([[3], [23]], np.int32)
    ks, vs = synth.act(rng, 2 * hkv, d).reshape(2, 1, hkv, d), synth.act(rng, 2 * hkv, d).reshape(2, 1, hkv, d)
    kb = [np.zeros((L, hkv, d), np.float16) for L in lens]
    vb = [np.zeros((L, hkv, d), np.float16) for L in lens]
    ko, vo = cx.copy_to_rag_buffer2(place, lens, ks, vs, kb, vb)
    for t in range(2):
        assert np.array_equal(ko[t][place[t, 0]], ks[t, 0]) and np.array_equal(vo[t][place[t, 0]], vs[t, 0])
        assert np.count_nonzero(ko[t]) == np.count_nonzero(ks[t, 0])
    # RMSNorm, fused add + norm, residual add, gated activation
    xx, yy = synth.act(rng, 4, 4096, 2.0), synth.act(rng, 4, 4096)
    w = (1 + 0.1 * rng.standard_normal(4096)).astype(np.float16)
    assert synth.ulp_diff_f16(cx.layernorm(xx, w, 1e-5, 1.0).view(np.uint16), oracle.rmsnorm(oracle.h2u(xx), oracle.h2u(w), 1e-5)).max() <= 1
    out, ssum = cx.layernorm_fuse_add(xx, yy, w, 1e-5)
    ro, rsum = oracle.rmsnorm(oracle.h2u(xx), oracle.h2u(w), 1e-5, x2=oracle.h2u(yy))
    assert synth.ulp_diff_f16(out.view(np.uint16), ro).max() <= 1
    assert np.array_equal(ssum.view(np.uint16), rsum)
    assert np.array_equal(cx.element_add_scale(xx, yy, 0.5, True).view(np.uint16), oracle.element_add_scale(oracle.h2u(xx), oracle.h2u(yy), 0.5, True))
    assert synth.ulp_diff_f16(cx.gate_mul(xx, yy, "silu").view(np.uint16), oracle.silu_mul(oracle.h2u(xx), oracle.h2u(yy))).max() <= 1


def test_gate_fuse_is_gate_mul_on_the_two_halves(cx):
    """nn::gate_fuse (ff_kernel.h:10, ff_kernel.cu:33-78): act(in) * gated on the halves of a fused (rows, 2 * dim_ff) projection --
    the bits of gate_mul_inplace on the separate halves, for both activations and a ragged row count"""
    rng = np.random.default_rng(5)
    for rows, ff, act in ((4, 4096, "silu"), (3, 1536, "gelu"), (1, 14336, "silu")):
        xx, yy = synth.act(rng, rows, ff, 2.0), synth.act(rng, rows, ff)
        fused = np.ascontiguousarray(np.concatenate([xx, yy], axis=1))
        assert np.array_equal(cx.gate_fuse(fused, act).view(np.uint16), cx.gate_mul(xx, yy, act).view(np.uint16))


def test_functions_copy_last_dim_and_concat_broadcast(cx):
    """bmengine::functions::copy_last_dim (index_select.h:32-39) and concat_broadcast_b (tensor_ops.h:13-14), the two helpers the
    reference's MLA layer needs beyond the linear layers' set: pure data movement, compared with numpy bit for bit"""
    rng = np.random.default_rng(9)
    a = rng.standard_normal((3, 5, 192)).astype(np.float16)
    for width, start, pad in ((64, 0, False), (64, 128, False), (128, 64, False), (96, 160, True), (64, 192, True)):
        want = np.zeros((3, 5, width), np.float16)
        take = max(0, min(width, 192 - start))
        want[..., :take] = a[..., start:start + take]
        assert np.array_equal(cx.copy_last_dim(a, width, start, pad).view(np.uint16), want.view(np.uint16)), (width, start, pad)
    with pytest.raises(Exception):
        cx.copy_last_dim(a, 96, 160, False)                                  # past the input's width without padding_zero
    b = rng.standard_normal((3, 64)).astype(np.float16)
    want = np.concatenate([a, np.broadcast_to(b[:, None, :], (3, 5, 64))], axis=-1)
    assert np.array_equal(cx.concat_broadcast_b(a, b).view(np.uint16), want.view(np.uint16))


def test_int8_route_through_cpp_is_bit_exact(cx, oracle):
    """Int8Linear::forward composed from the int8_op:: wrappers: quantised rows, int32 product and the scaled-back
    outputs are bit-identical to the oracle (north_star: bit-exact for INT8 GEMM)."""
    rng = np.random.default_rng(12)
    m, k, n = 5, 1024, 384
    x = synth.act(rng, m, k, 3.0)
    wq = rng.integers(-127, 128, size=(n, k), dtype=np.int8)
    ws = (np.abs(rng.standard_normal(n)) * 0.01 + 1e-3).astype(np.float16)
    q, sx = cx.quant_calc_scale(x)
    rq, rsx = oracle.quant_calc_scale(oracle.h2u(x))
    assert np.array_equal(q, rq) and np.array_equal(sx, rsx)
    y, acc = cx.int8_linear(x, wq, ws)
    racc = oracle.int8_gemm_nt(rq, wq)
    assert np.array_equal(acc, racc)
    assert np.array_equal(y.view(np.uint16), oracle.quant_scale_back(racc, rsx, oracle.h2u(ws)))
    w = (1 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    out, q2, s2 = cx.layernorm_quant(x, w, 1e-5)
    ro, rq2, rs2 = oracle.rmsnorm_quant(oracle.h2u(x), oracle.h2u(w), 1e-5)
    assert np.array_equal(q2, rq2) and np.allclose(s2, rs2, rtol=3e-7, atol=0)   # the scale carries the block-sum association
    assert synth.ulp_diff_f16(out.view(np.uint16), ro).max() <= 1
    b_acc = rng.integers(-50000, 50000, size=(m, n), dtype=np.int32)
    got = cx.quant_back_act_mul(racc, rsx, ws, b_acc, rsx, ws, "silu")
    assert synth.ulp_diff_f16(got.view(np.uint16), oracle.quant_back_act_mul(racc, rsx, oracle.h2u(ws), b_acc, rsx, oracle.h2u(ws), "silu")).max() <= 1


def test_errors_become_bmengine_exceptions(cx):
    with pytest.raises(RuntimeError, match="size K mismatch"):
        cx.raises_on_bad_shape()


@pytest.mark.parametrize("prepack", [True, False])
def test_moe_feed_forward_through_cpp_is_bit_exact(cx, oracle, prepack):
    """nn::gptq::gemm_moe_up / gemm_moe_down with the reference's operands ((EXP, N, K/8) stacks, int8 zeros, float routing
    weights): packed once (amd_pack_moe, the load path) or per call (an unmodified caller); both bit-equal to the oracle's
    restatement of the CUDA kernels."""
    rng = np.random.default_rng(31)
    e, n_ff, k, dim, g, m, top_k = 4, 64, 256, 48, 128, 3, 2

    def stack(rows, cols):
        l = [oracle.gptq_prepare_k_major(*synth.gptq_hf(rng, cols, rows, g), g) for _ in range(e)]
        return tuple(np.stack([a[i] for a in l]) for i in range(3))
    gs, us, ds = stack(n_ff, k), stack(n_ff, k), stack(dim, 128)
    ids = np.stack([rng.choice(e, top_k, replace=False) for _ in range(m)]).astype(np.int32)
    wts = rng.random((m, top_k)).astype(np.float32)
    x = synth.act(rng, m, k)
    # the down projection reads the first 64 of its 128 input columns from the up output (zero-padded to one group)
    ref_up = oracle.gptq_moe_up(oracle.h2u(x), gs, us, ids)
    a_in = np.zeros((m, top_k, 128), np.float16)
    a_in[:, :, :n_ff] = oracle.u2h(ref_up)
    ref_dn = oracle.gptq_moe_down(oracle.h2u(a_in), ds, ids, wts)
    got_up, got_dn = cx.gemm_moe_steps(x, gs[0].view(np.int32), gs[1], oracle.u2h(gs[2]), us[0].view(np.int32), us[1], oracle.u2h(us[2]),
                                       a_in, ds[0].view(np.int32), ds[1], oracle.u2h(ds[2]), ids, wts, 0, prepack)
    assert np.array_equal(got_up.view(np.uint16), ref_up)
    assert np.array_equal(got_dn.view(np.uint16), ref_dn)


def test_moe_router_dispatch_combine_through_cpp(cx, oracle):
    """nn::top_k_softmax / group_topk_softmax / plus_for_sort / calc_reverse_idx / fill_m_indices_padded_indices / sum_experts
    (src/nn/feedforward/ff_kernel.h:16-96) with the reference's signatures, against the oracle"""
    rng = np.random.default_rng(12)
    tokens, experts, k = 19, 64, 4
    logits = (rng.standard_normal((tokens, experts)) * 1.3).astype(np.float16)
    v, idx = cx.moe_route(logits, None, 1, 1, k, k, True, 1.0, "softmax")
    wv, widx, _, _ = oracle.moe_top_k_softmax(logits.view(np.uint16), k, k, True, 1.0, "softmax", 0, 0)
    assert np.array_equal(idx, widx) and np.abs(v.view(np.int32).astype(np.int64) - wv.view(np.int32)).max() <= 8
    bias = (rng.standard_normal(experts) * 0.1).astype(np.float32)
    v2, idx2 = cx.moe_route(logits, bias, 4, 2, k, k + 1, True, 2.5, "sigmoid")
    wv2, widx2, _, _ = oracle.moe_group_topk(logits.view(np.uint16), bias, k, 4, 2, k + 1, True, 2.5, "sigmoid", 0, 0)
    assert np.array_equal(idx2, widx2) and np.abs(v2.view(np.int32).astype(np.int64) - wv2.view(np.int32)).max() <= 8
    # dispatch of the first routing result
    loads = np.bincount(idx.ravel(), minlength=experts).astype(np.int32)
    order = np.argsort(idx.ravel(), kind="stable").astype(np.int32)
    keys, rev, mi, pad, total = cx.moe_dispatch(idx, order, loads.tolist(), experts, 64)
    assert np.array_equal(keys, oracle.moe_plus_for_sort(idx, experts, 1))
    wrev, off = oracle.moe_calc_reverse_idx(idx, order, loads, experts)
    wmi, wpad, wtotal = oracle.moe_fill_m_indices(loads, 64, experts)
    assert np.array_equal(rev, wrev) and np.array_equal(mi, wmi) and np.array_equal(pad, wpad) and total == wtotal
    # combine per-expert outputs
    dim = 256
    y = rng.standard_normal((tokens * k, dim)).astype(np.float16)
    parts = [y[off[e]:off[e] + loads[e]] if loads[e] else None for e in range(experts)]
    got = cx.moe_combine(parts, idx.ravel(), rev, v)
    want = oracle.moe_sum_experts_arr([None if p is None else p.view(np.uint16) for p in parts], idx.ravel(), rev, v, dim)
    assert np.array_equal(got.view(np.uint16), want)

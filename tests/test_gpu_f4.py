"""f4, first part -- GPU parity through the C ABI:
  * per-token 1x128 FP8 activation quantisation (fp8_util.cu:229-323): codes and scales BIT-EXACT against the oracle
  * 128x128-block weight dequantisation (:325-385): bit-exact
  * the block-scaled FP8 GEMM deep_gemm_fp8_block_h20_group stands for (closed binary in the reference): against the format's
    definition in fp64, output rounding of T + fp32 accumulation noise; plain and grouped (m_indices) forms
  * the MoE router: top_k_softmax and the group-limited top-k of DeepSeek-V3 -- expert ids EXACT, weights within a few fp32 ulp
    (device expf vs glibc), load counters exact
  * the reference's own Fp8Block layer (linear.cpp compiled unmodified, -DENABLE_DS_DEEP_GEMM) on top of them."""
import os
import sys

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.view(dtype) if dtype is not None else t


def _bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _acts(rng, m, n, dtype):
    x = (rng.standard_normal((m, n)) * np.exp(rng.standard_normal((m, 1)) * 2.0)).astype(np.float32)
    x[:, :128] *= 1e-6                                   # a block below the 1e-4 clamp
    t = torch.from_numpy(x).to(dtype)
    return t


@pytest.mark.parametrize("dtype,code", [(torch.float16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("col_major", [True, False])
def test_per_token_cast_bit_exact(oracle, dev, dtype, code, col_major):
    from zhilight_amd import ops
    rng = np.random.default_rng(3 + code)
    for m, n in ((1, 7168), (5, 2048), (70, 512)):
        x = _acts(rng, m, n, dtype)
        pad = torch.zeros((m, n + 64), dtype=dtype)
        pad[:, :n] = x                                    # strided rows
        codes, scales = ops.fp8_per_token_cast(pad.to(dev)[:, :n], scale_col_major=col_major)
        want_c, want_s = oracle.fp8_per_token_cast(_bits(x), col_major=col_major, dtype=code)
        assert np.array_equal(codes.cpu().numpy(), want_c)
        got_s = scales.cpu().numpy()
        if col_major:
            assert np.array_equal(got_s[:, :m].view(np.uint32), want_s[:, :m].view(np.uint32))
        else:
            assert np.array_equal(got_s[:m].view(np.uint32), want_s[:m].view(np.uint32))
        # the format's promise: dequantised codes within half an E4M3 step (2^-4 relative) of x, the block maximum exact
        deq = oracle.e4m3_to_f32(want_c).reshape(m, n // 128, 128) * (want_s[:, :m].T if col_major else want_s[:m])[:, :, None]
        xf = x.float().numpy().reshape(m, n // 128, 128)
        amax = np.maximum(np.abs(xf).max(axis=2, keepdims=True), 1e-4)
        assert (np.abs(deq - xf) <= amax * 2.0 ** -4 * 1.001).all()


@pytest.mark.parametrize("dtype,code", [(torch.float16, 0), (torch.bfloat16, 1)])
def test_block_dequant_bit_exact(oracle, dev, dtype, code):
    from zhilight_amd import ops
    rng = np.random.default_rng(9)
    rows, cols = 300, 640                                 # ragged last row block
    w8 = rng.integers(0, 256, size=(rows, cols), dtype=np.uint8)
    w8[(w8 & 0x7f) == 0x7f] = 0x7e                        # no NaN codes in a checkpoint
    sc = (np.abs(rng.standard_normal((3, 5))) * 0.01 + 1e-3).astype(np.float32)
    got = ops.fp8_block_dequant(_t(w8, dev), _t(sc, dev), dtype)
    assert np.array_equal(_bits(got), oracle.fp8_block_dequant(w8, sc, dtype=code))


def _block_weight(rng, n, k, groups=None):
    shape = (n, k) if groups is None else (groups, n, k)
    w8 = rng.integers(0, 256, size=shape, dtype=np.uint8)
    w8[(w8 & 0x7f) == 0x7f] = 0x7e
    w8[(w8 & 0x7f) > 0x70] -= 0x20                        # keep |w| moderate: the products stay far from fp16 overflow
    sshape = ((n + 127) // 128, k // 128) if groups is None else (groups, (n + 127) // 128, k // 128)
    sw = (np.abs(rng.standard_normal(sshape)) * 2e-3 + 1e-3).astype(np.float32)
    return w8, sw


def _gemm_bar(got_bits, want_bits, oracle, code):
    f = (lambda b: oracle.u2h(b).astype(np.float64)) if code == 0 else (lambda b: (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64))
    got, want = f(got_bits), f(want_bits)
    rms = np.sqrt((want ** 2).mean(axis=1, keepdims=True))     # per row: the rows carry very different magnitudes
    ulp = 2.0 ** (-10 if code == 0 else -7)               # one output rounding of T on either side ...
    # ... + the fp8 MFMA's own floor: inside a lane's group of 8 products, whatever lies 2^12..2^15 below the largest one is
    # dropped before the fp32 accumulation (profiles/r03_fp8_mfma_precision_probe.txt): <= 3e-4 of the output rms on random
    # data (measured 2.2e-4 of a row's rms), where an fp16 MFMA kernel is held to 2e-5
    assert (np.abs(got - want) <= ulp * np.abs(want) + 5e-4 * rms).all(), float((np.abs(got - want) / rms).max())


@pytest.mark.parametrize("dtype,code", [(torch.bfloat16, 1), (torch.float16, 0)])
@pytest.mark.parametrize("m,n,k", [(1, 512, 1024), (5, 200, 512), (70, 384, 640)])
def test_block_gemm_against_the_format_definition(oracle, dev, dtype, code, m, n, k):
    from zhilight_amd import ops
    rng = np.random.default_rng(m + n)
    x = _acts(rng, m, k, dtype)
    a8, sa = oracle.fp8_per_token_cast(_bits(x), dtype=code)
    w8, sw = _block_weight(rng, n, k)
    got = ops.fp8_block_gemm(_t(a8, dev), _t(sa, dev), _t(w8, dev), _t(sw, dev), dtype=dtype)
    _gemm_bar(_bits(got), oracle.fp8_block_gemm(a8, sa, w8, sw, dtype=code), oracle, code)
    # the composed linear (Fp8Block::forward) = the same launch behind the device-side cast
    got2 = ops.fp8_block_linear(x.to(dev), _t(w8, dev), _t(sw, dev))
    assert torch.equal(got2, got)
    # and it is a sane quantisation of x . W^T: rms error of a few percent of the exact product of the dequantised operands
    wd = oracle.e4m3_to_f32(w8) * np.repeat(np.repeat(sw, 128, axis=0)[:n], 128, axis=1)
    ref = x.float().numpy().astype(np.float64) @ wd.T
    g = got.float().cpu().numpy().astype(np.float64)
    assert np.sqrt(((g - ref) ** 2).mean()) <= 0.05 * np.sqrt((ref ** 2).mean())


def test_grouped_block_gemm(oracle, dev):
    """the m-grouped contiguous layout of FP8Block::grouped_gemm: rows sorted by expert, 64-row aligned, -1 = padding row"""
    from zhilight_amd import ops
    rng = np.random.default_rng(77)
    groups, n, k = 3, 256, 512
    m_idx = np.concatenate([np.full(64, 2), np.full(40, 0), np.full(24, -1), np.full(64, 1), np.full(10, 0), np.full(54, -1)]).astype(np.int32)
    m = m_idx.size
    x = _acts(rng, m, k, torch.bfloat16)
    a8, sa = oracle.fp8_per_token_cast(_bits(x), dtype=1)
    w8, sw = _block_weight(rng, n, k, groups)
    sentinel = torch.full((m, n), 7.0, dtype=torch.bfloat16, device=dev)
    got = ops.fp8_block_gemm(_t(a8, dev), _t(sa, dev), _t(w8, dev), _t(sw, dev), m_indices=_t(m_idx, dev), out=sentinel.clone())
    want = oracle.fp8_block_gemm(a8, sa, w8, sw, m_indices=m_idx, dtype=1)
    live = m_idx >= 0
    _gemm_bar(_bits(got)[live], want[live], oracle, 1)
    assert torch.equal(got[~torch.from_numpy(live).to(dev)], sentinel[~torch.from_numpy(live).to(dev)])     # padding rows untouched


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


@pytest.mark.parametrize("dtype,code", [(torch.float16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("scoring,renorm,ext", [("softmax", True, 8), ("softmax", False, 10), ("sigmoid", True, 8), ("linear", False, 8)])
def test_top_k_softmax_router(oracle, dev, dtype, code, scoring, renorm, ext):
    from zhilight_amd import ops
    rng = np.random.default_rng(11)
    tokens, experts, k, world = 37, 128, 8, 4            # Qwen3-MoE: 128 experts, top 8
    logits = torch.from_numpy(rng.standard_normal((tokens, experts)).astype(np.float32) * 1.5).to(dtype)
    wl = torch.zeros(world, dtype=torch.int32, device=dev)
    el = torch.zeros(experts, dtype=torch.int32, device=dev)
    v, idx = ops.moe_top_k_softmax(logits.to(dev), k, ext, renorm, 1.25, scoring, wl, el, world)
    wv, widx, wwl, wel = oracle.moe_top_k_softmax(_bits(logits), k, ext, renorm, 1.25, scoring, code, world)
    assert np.array_equal(idx.cpu().numpy()[:, :k], widx[:, :k])
    assert _ulps(v.cpu().numpy(), wv).max() <= 8
    assert np.array_equal(wl.cpu().numpy(), wwl) and np.array_equal(el.cpu().numpy(), wel)
    assert (v.cpu().numpy()[:, k:] == 1.0).all()


@pytest.mark.parametrize("dtype,code", [(torch.float16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("experts,groups,topk_group,k,scoring,bias", [(256, 8, 4, 8, "sigmoid", True), (160, 8, 3, 6, "softmax", False),
                                                                      (64, 4, 2, 4, "sigmoid", False)])
def test_group_limited_router(oracle, dev, dtype, code, experts, groups, topk_group, k, scoring, bias):
    from zhilight_amd import ops
    rng = np.random.default_rng(experts + k)
    tokens, world, ext = 29, 8, k + 1                     # + one shared-expert slot (top_k_may_share)
    logits = torch.from_numpy(rng.standard_normal((tokens, experts)).astype(np.float32) * 1.2).to(dtype)
    b = (rng.standard_normal(experts) * 0.1).astype(np.float32) if bias else None
    wl = torch.zeros(world, dtype=torch.int32, device=dev)
    el = torch.zeros(experts, dtype=torch.int32, device=dev)
    v, idx = ops.moe_group_topk(logits.to(dev), None if b is None else _t(b, dev), groups, topk_group, k, ext, True, 2.5, scoring, wl, el, world)
    wv, widx, wwl, wel = oracle.moe_group_topk(_bits(logits), b, k, groups, topk_group, ext, True, 2.5, scoring, code, world)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert _ulps(v.cpu().numpy(), wv).max() <= 8
    assert np.array_equal(wl.cpu().numpy(), wwl) and np.array_equal(el.cpu().numpy(), wel)
    # the routing contract itself: ids inside the selected groups only, weights sum to the scaling factor
    gi = idx.cpu().numpy()[:, :k] // (experts // groups)
    assert all(len(set(r)) <= topk_group for r in gi)
    assert np.allclose(v.cpu().numpy()[:, :k].sum(axis=1), 2.5, rtol=1e-5)


def test_reference_fp8block_layer(oracle, dev):
    """the reference's Fp8Block (src/nn/linear/linear.cpp:1697-1950, compiled unmodified with ENABLE_DS_DEEP_GEMM): load_parameter of
    the fp8 weight + weight_scale_inv, quant_input -> per_token_cast_to_fp8, deep_gemm_fp8_block_h20_group = this boundary's GEMM"""
    from zhilight_amd import _lib, build, ops
    _lib.lib()
    path = build.refcompile_target()
    if not os.path.exists(path):
        pytest.skip("zl_reflinear was not built (no reference tree at build time)")
    sys.path.insert(0, os.path.dirname(path))
    try:
        import zl_reflinear
    finally:
        sys.path.pop(0)
    rng = np.random.default_rng(5)
    k, n, m = 1024, 384, 6
    w8, sw = _block_weight(rng, n, k)
    lin = zl_reflinear.RefLinear(k, n, 10, bf16=True)      # QuantType::FP8_Block
    lin.load({"l.weight": w8.view(np.int8), "l.weight_scale_inv": sw}, "l")
    x = _acts(rng, m, k, torch.bfloat16)
    got = lin.forward(_bits(x))                           # uint16 = bf16 bits
    want = ops.fp8_block_linear(x.to(dev), _t(w8, dev), _t(sw, dev))
    assert np.array_equal(got, _bits(want))
    a8, sa = oracle.fp8_per_token_cast(_bits(x), dtype=1)
    _gemm_bar(got, oracle.fp8_block_gemm(a8, sa, w8, sw, dtype=1), oracle, 1)
    # get_dequant_weight -> dequant_fp8_block_weight
    assert np.array_equal(lin.dequant_weight(), oracle.fp8_block_dequant(w8, sw, dtype=1))


@pytest.mark.parametrize("dtype,code", [(torch.float16, 0), (torch.bfloat16, 1)])
def test_moe_dispatch_and_combine_bit_exact(oracle, dev, dtype, code):
    """the index bookkeeping between router and grouped GEMMs and the weighted combine (ff_kernel.cu:518-1082), each against its
    restatement, plus the property that ties them together: gathering by the computed positions and combining per expert
    equals combining the concatenated layout"""
    from zhilight_amd import ops
    rng = np.random.default_rng(21)
    tokens, experts, k, world, dim = 45, 16, 4, 4, 384
    ids = np.stack([rng.choice(experts, size=k, replace=False) for _ in range(tokens)]).astype(np.int32)
    w = rng.random((tokens, k)).astype(np.float32)
    loads = np.bincount(ids.ravel(), minlength=experts).astype(np.int32)
    rank_loads = np.array([loads[r::world].sum() for r in range(world)], np.int32)
    all_loads = np.concatenate([loads, rank_loads])
    # plus_for_sort + the sort the reference runs on it (functions::sort: here a stable argsort of the same keys)
    keys = ops.moe_plus_for_sort(_t(ids, dev), experts, world)
    assert np.array_equal(keys.cpu().numpy(), oracle.moe_plus_for_sort(ids, experts, world))
    for by_rank in (False, True):
        order = np.argsort((keys.cpu().numpy() if by_rank else ids).ravel(), kind="stable").astype(np.int32)
        rev = ops.moe_calc_reverse_idx(_t(ids, dev), _t(order, dev), all_loads, experts, world, by_rank)
        want, _ = oracle.moe_calc_reverse_idx(ids, order, all_loads, experts, world, by_rank)
        assert np.array_equal(rev.cpu().numpy(), want)
        # a position inside an expert's run: 0 .. load - 1, every value once per expert
        r = rev.cpu().numpy().reshape(tokens, k)
        for e in range(experts):
            assert sorted(r[ids == e].tolist()) == list(range(loads[e]))
    # m-grouped layout of the local experts (all of them, and rank 1's under expert parallelism)
    for ep, rk in ((False, 0), (True, 1)):
        mi, pad, total = ops.moe_fill_m_indices_padded_indices(all_loads, 64, experts, dev, ep, rk, world)
        wmi, wpad, wtotal = oracle.moe_fill_m_indices(all_loads, 64, experts, rk if ep else 0, world if ep else 1)
        assert total == wtotal and np.array_equal(mi.cpu().numpy(), wmi) and np.array_equal(pad.cpu().numpy(), wpad)
    # combine: concatenated form
    order = np.argsort(ids.ravel(), kind="stable").astype(np.int32)
    rev, off = oracle.moe_calc_reverse_idx(ids, order, all_loads, experts)
    y = torch.from_numpy(rng.standard_normal((tokens * k, dim)).astype(np.float32)).to(dtype)     # expert outputs, sorted-by-expert rows
    pos = (off[ids.ravel()] + rev).astype(np.int32)                                              # row of (token, slot) in y
    got = ops.moe_sum_experts(y.to(dev), _t(pos, dev), _t(w, dev))
    assert np.array_equal(_bits(got), oracle.moe_sum_experts(_bits(y), pos, w, dtype=code))
    # per-expert form: the same rows split by expert, positions = rev; experts without tokens pass None
    parts = [y[off[e]:off[e] + loads[e]].contiguous() if loads[e] else None for e in range(experts)]
    got2 = ops.moe_sum_experts_arr([None if p is None else p.to(dev) for p in parts], _t(ids.ravel(), dev), _t(rev, dev), _t(w, dev))
    assert torch.equal(got2, got)
    want2 = oracle.moe_sum_experts_arr([None if p is None else _bits(p) for p in parts], ids.ravel(), rev, w, dim, dtype=code)
    assert np.array_equal(_bits(got2), want2)
    # expert parallelism: the ranks' partial sums of one token cover every expert exactly once
    one = ids[:1]
    rows = [torch.from_numpy(rng.standard_normal((1, dim)).astype(np.float32)).to(dtype) for _ in range(experts)]
    full = oracle.moe_sum_experts_arr([_bits(r) for r in rows], one.ravel(), None, w[:1], dim, dtype=code)
    acc = np.zeros((1, dim), np.float64)
    for rk in range(world):
        part = ops.moe_sum_experts_arr([r.to(dev) for r in rows], _t(one.ravel(), dev), None, _t(w[:1], dev), True, world, rk)
        assert np.array_equal(_bits(part), oracle.moe_sum_experts_arr([_bits(r) for r in rows], one.ravel(), None, w[:1], dim, True, world, rk, code))
        acc += part.float().cpu().numpy()
    f = oracle.u2h(full).astype(np.float64) if code == 0 else (full.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert np.abs(acc - f).max() <= 2.0 ** (-6 if code else -9) * max(1.0, np.abs(f).max())


def test_moe_shared_expert_load_balancing(oracle, dev):
    from zhilight_amd import ops
    rng = np.random.default_rng(8)
    tokens, k, ext, world, local = 23, 6, 8, 4, 16
    ids = np.zeros((tokens, ext), np.int32)
    ids[:, :k] = rng.integers(0, local * world, size=(tokens, k))
    wl = np.bincount(ids[:, :k].ravel() % world, minlength=world).astype(np.int32)
    el = np.zeros((local + ext - k) * world, np.int32)
    t_ids, t_wl, t_el = _t(ids, dev), _t(wl, dev), _t(el, dev)
    ops.moe_route_shared_lb(t_ids, torch.ones((tokens, ext), dtype=torch.float32, device=dev), t_wl, t_el, k, local)
    w_ids, w_wl, w_el = oracle.moe_route_shared_lb(ids, wl, el, k, local)
    assert np.array_equal(t_ids.cpu().numpy(), w_ids) and np.array_equal(t_wl.cpu().numpy(), w_wl) and np.array_equal(t_el.cpu().numpy(), w_el)
    assert t_wl.cpu().numpy().max() <= (tokens * ext + world - 1) // world + 0      # nobody above the even share


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("dtype,code", [(torch.float16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("h", [16, 128, 20])
def test_mla_decode_attention_over_the_latent_cache(oracle, dev, dtype, code, h, algo):
    """DeepSeek MLA, decode rows: every head reads the same 576-value latent rows (key = all of it, value = the first 512).  Against
    the fp64 statement (E) at T's output rounding, and against the reference's open route (R: scores and probabilities rounded to
    T) within that route's own noise."""
    from zhilight_amd import ops
    rng = np.random.default_rng(h)
    lens = [1, 64, 65, 700] if h == 16 else ([130, 1] if h == 128 else [17, 767])      # (h = 20: a ragged last group of heads)
    b, max_len = len(lens), 768
    q = torch.from_numpy((rng.standard_normal((b, h, 576)) * 0.4).astype(np.float32)).to(dtype)
    bufs = [torch.from_numpy((rng.standard_normal((max_len, 576)) * 0.6).astype(np.float32)).to(dtype) for _ in range(b)]
    for t in bufs:
        t[-1] = float("nan")                                            # never visible: must not leak
    dbufs = [t.to(dev) for t in bufs]
    addrs = torch.tensor([t.data_ptr() for t in dbufs], dtype=torch.int64, device=dev)
    buf_lens = torch.tensor([max_len - 1] * b, dtype=torch.int32, device=dev)
    valid = torch.tensor(lens, dtype=torch.int32, device=dev)
    scale = 0.1147
    got = ops.mla_decode_attention(q.to(dev), buf_lens, addrs, scale, max_len, valid, algo=algo)   # 0: matrix cores, 1: the VALU kernel
    args = (_bits(q), [max_len - 1] * b, lens, [_bits(t) for t in bufs])
    f = (lambda u: oracle.u2h(u).astype(np.float64)) if code == 0 else (lambda u: (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64))
    E = f(oracle.mla_decode_attn(*args, scale=scale, dtype=code, flavour="E"))
    R = f(oracle.mla_decode_attn(*args, scale=scale, dtype=code, flavour="R"))
    g = f(_bits(got))
    assert np.isfinite(g).all()
    ulp = 2.0 ** (-10 if code == 0 else -7)
    assert (np.abs(g - E) <= ulp * np.abs(E) + 2e-5 * np.abs(E).max()).all(), float(np.abs(g - E).max() / np.abs(E).max())
    assert np.abs(g - R).max() <= max(np.abs(R - E).max() * 1.5, 2 * ulp * np.abs(E).max())


@pytest.mark.parametrize("dtype,code", [(torch.float16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("b,max_len", [(3, 768), (9, 320), (32, 1088), (16, 2112)])
def test_mla_decode_wide_kernel(oracle, dev, dtype, code, b, max_len):
    """k_mla_decode_wide (all 128 heads of a task in one workgroup, latent rows through LDS by LDS-DMA; zl_mla_decode_attn_ex algo 2,
    and what algo 0 picks from 16 tasks with 2048-key buffers on): against the fp64 statement at T's output rounding, against the 16-heads-per-workgroup
    kernel within two output roundings (different split plans), ragged lengths incl. 1 key, a split boundary, 16 k + 1 keys, the
    whole buffer, and an empty task."""
    from zhilight_amd import ops
    rng = np.random.default_rng(1000 * b + code)
    h = 128
    pool = [1, 16, 17, 63, 64, 65, 129, max_len // 2 + 1, max_len - 1, max_len - 2, 0]
    lens = [pool[i % len(pool)] if i < len(pool) else int(rng.integers(1, max_len)) for i in range(b)]
    q = torch.from_numpy((rng.standard_normal((b, h, 576)) * 0.4).astype(np.float32)).to(dtype)
    bufs = [torch.from_numpy((rng.standard_normal((max_len, 576)) * 0.6).astype(np.float32)).to(dtype) for _ in range(b)]
    for t in bufs:
        t[-1] = float("nan")                                            # never visible: must not leak
    dbufs = [t.to(dev) for t in bufs]
    addrs = torch.tensor([t.data_ptr() for t in dbufs], dtype=torch.int64, device=dev)
    buf_lens = torch.tensor([max_len - 1] * b, dtype=torch.int32, device=dev)
    valid = torch.tensor(lens, dtype=torch.int32, device=dev)
    scale = 0.1147
    wide = ops.mla_decode_attention(q.to(dev), buf_lens, addrs, scale, max_len, valid, algo=2)
    auto = ops.mla_decode_attention(q.to(dev), buf_lens, addrs, scale, max_len, valid, algo=0)
    picks_wide = b >= 16 and max_len >= 2048
    if picks_wide:
        assert torch.equal(wide.view(torch.int16), auto.view(torch.int16))           # algo 0 IS the wide kernel there
    live = [i for i, n in enumerate(lens) if n > 0]
    args = (_bits(q[live]), [max_len - 1] * len(live), [lens[i] for i in live], [_bits(bufs[i]) for i in live])
    f = (lambda u: oracle.u2h(u).astype(np.float64)) if code == 0 else (lambda u: (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64))
    E = f(oracle.mla_decode_attn(*args, scale=scale, dtype=code, flavour="E"))
    g = f(_bits(wide))[live]
    assert np.isfinite(g).all()
    ulp = 2.0 ** (-10 if code == 0 else -7)
    assert (np.abs(g - E) <= ulp * np.abs(E) + 2e-5 * np.abs(E).max()).all(), float(np.abs(g - E).max() / np.abs(E).max())
    if not picks_wide:
        a = f(_bits(auto))[live]
        assert np.abs(g - a).max() <= 2 * ulp * np.abs(E).max()
    dead = [i for i, n in enumerate(lens) if n == 0]
    assert all(float(wide[i].float().abs().max()) == 0.0 for i in dead)               # an empty task: zeros, as the other kernels


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_flash_mla_binding_over_a_paged_cache(dev, dtype):
    """ds::mha_fwd_kvcache_mla (ds_flash_mla_api.h:16-30), the reference's FlashMLA FFI: the paged call returns the bits of the
    contiguous-buffer call on the same rows (the same split plan, the same arithmetic), two query rows fold into the head axis
    (ds_flash_mla_api.cpp:121-126), and softmax_lse is log sum exp of the scaled scores."""
    from zhilight_amd import ops
    rng = np.random.default_rng(11)
    lens, len_q, h, page = [1, 64, 65, 300], 2, 16, 64
    b, max_blocks = len(lens), 6
    max_len = page * max_blocks
    q = torch.from_numpy((rng.standard_normal((b, len_q, h, 576)) * 0.4).astype(np.float32)).to(dtype).to(dev)
    bufs = [torch.from_numpy((rng.standard_normal((max_len, 576)) * 0.6).astype(np.float32)).to(dtype).to(dev) for _ in range(b)]
    num_blocks = b * max_blocks + 3
    order = rng.permutation(num_blocks)[: b * max_blocks].astype(np.int32).reshape(b, max_blocks)
    kcache = torch.full((num_blocks, page, 1, 576), float("nan"), dtype=dtype, device=dev)
    for i in range(b):
        for j in range(max_blocks):
            kcache[int(order[i, j]), :, 0, :] = bufs[i][j * page:(j + 1) * page]
    table = torch.from_numpy(order).to(dev)
    seqlens = torch.tensor(lens, dtype=torch.int32, device=dev)
    scale = 0.1147
    out, lse = ops.mha_fwd_kvcache_mla(q, kcache, 512, seqlens, table, scale)
    assert out.shape == (b, len_q, h, 512) and lse.shape == (b, 1, len_q * h)
    addrs = torch.tensor([t.data_ptr() for t in bufs], dtype=torch.int64, device=dev)
    want = ops.mla_decode_attention(q.view(b, len_q * h, 576), seqlens, addrs, scale, max_len)
    assert torch.equal(out.view(b, len_q * h, 512).view(torch.int16), want.view(torch.int16))
    for i, n in enumerate(lens):
        s = (q[i].view(len_q * h, 576).double() @ bufs[i][:n].double().T) * scale
        assert torch.allclose(lse[i, 0].double(), torch.logsumexp(s, dim=-1), rtol=0, atol=2e-5)
    with pytest.raises(ops.ZLError):
        ops.mha_fwd_kvcache_mla(q, kcache, 512, seqlens, table, scale, is_causal=True)       # causal multi-row queries: not on this path


def test_flash_mla_binding_through_the_cpp_names(dev):
    """The same two functions under the reference's C++ names (hostcpp/nn_amd.h ds::get_mla_metadata / ds::mha_fwd_kvcache_mla)."""
    from zhilight_amd import ops, zl_internals
    rng = np.random.default_rng(12)
    lens, h, page, max_blocks = [70, 129], 16, 64, 3
    b = len(lens)
    q = (rng.standard_normal((b, 1, h, 576)) * 0.4).astype(np.float16)
    kcache = (rng.standard_normal((b * max_blocks, page, 1, 576)) * 0.6).astype(np.float16)
    table = rng.permutation(b * max_blocks).astype(np.int32).reshape(b, max_blocks)
    seqlens = np.asarray(lens, np.int32)
    ctx = zl_internals.Context(0)
    got, lse = ctx.mha_fwd_kvcache_mla(q, kcache, seqlens, table, 0.1147)
    want, wlse = ops.mha_fwd_kvcache_mla(torch.from_numpy(q).to(dev), torch.from_numpy(kcache).to(dev), 512, torch.from_numpy(seqlens).to(dev),
                                         torch.from_numpy(table).to(dev), 0.1147)
    assert np.array_equal(got.view(np.uint16), want.cpu().numpy().view(np.uint16))
    assert np.array_equal(lse, wlse.cpu().numpy())


@pytest.mark.parametrize("tokens", [3, 37])
@pytest.mark.parametrize("router,dtype", [("top_k", torch.bfloat16), ("group", torch.bfloat16), ("top_k", torch.float16)])
def test_moe_feed_forward_flow_equals_the_per_token_sum(dev, tokens, router, dtype):
    """FeedForward::forward_gpu_dispatch (feedforward.cpp:1075-1150) strung together from the launchers (zhilight_amd/moe.py): route ->
    m-grouped 64-aligned layout -> grouped FP8 block GEMMs -> act(in) * gated -> grouped GEMM -> weighted combine.  Every row of the
    grouped GEMMs is computed from its own fragments, so the flow must return the BITS of the same sum written token by token and
    slot by slot with Fp8Block::forward on single rows (3 tokens: most experts get nothing; 37: ragged runs, padding rows)."""
    from zhilight_amd.moe import Fp8BlockMoE
    g = torch.Generator(device="cpu").manual_seed(100 + tokens)
    e, k, dim, ff = 64, 4, 256, 384

    def codes(*shape):                                                        # finite e4m3 codes of either sign (0x7f / 0xff are NaN)
        return (torch.randint(0, 0x78, shape, generator=g, dtype=torch.int32) | (torch.randint(0, 2, shape, generator=g, dtype=torch.int32) << 7)) \
            .to(torch.uint8).to(dev)

    def scales(*shape):
        return (torch.rand(shape, generator=g) * 0.008 + 0.002).to(dev)

    moe = Fp8BlockMoE((torch.randn(e, dim, generator=g) * 0.5).to(dtype).to(dev), codes(e, ff, dim), scales(e, ff // 128, dim // 128),
                      codes(e, ff, dim), scales(e, ff // 128, dim // 128), codes(e, dim, ff), scales(e, dim // 128, ff // 128), top_k=k,
                      scoring_func="sigmoid" if router == "group" else "softmax", n_group=4 if router == "group" else 1,
                      topk_group=2 if router == "group" else 1, routed_scaling_factor=2.5 if router == "group" else 1.0,
                      e_score_correction_bias=(torch.randn(e, generator=g) * 0.1).to(dev) if router == "group" else None,
                      shared=(codes(2 * ff, dim), scales(2 * ff // 128, dim // 128), codes(2 * ff, dim), scales(2 * ff // 128, dim // 128),
                              codes(dim, 2 * ff), scales(dim // 128, 2 * ff // 128)) if router == "group" else None)   # DeepSeek: + a shared expert
    x = torch.randn(tokens, dim, generator=g).to(dtype).to(dev)
    ids, w, loads = moe.route(x)
    assert int(loads[:e].sum()) == tokens * k and bool((ids >= 0).all()) and bool((ids < e).all())
    got = moe.forward(x)
    want = moe.forward_per_token(x)
    assert got.shape == (tokens, dim) and torch.isfinite(got.float()).all() and float(got.float().abs().max()) > 0
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    if router == "group":                                                      # the shared expert is really in the sum
        routed, moe.shared = moe.shared, None
        assert not torch.equal(moe.forward(x), got)
        moe.shared = routed


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("tokens", [5, 41])
def test_moe_feed_forward_expert_parallel(dev, tokens, world):
    """MOE_EXP_PARALLEL of the reference (feedforward.cpp:251-305, :599-629, :1079-1150) in zhilight_amd/moe.py: rank r holds the experts
    e % world == r, routes every token, sorts the (token, slot) pairs by (rank, expert), takes its slice, and returns the PARTIAL sum of
    its experts (+ its shard of the tensor-parallel shared expert) that the layer's all-reduce completes.  All ranks run here one after
    the other on one device: a token whose experts all live on one rank comes out of that rank with the BITS of the one-rank flow minus
    the shared part and as exact zeros from the others; every row's rank partials add up to the one-rank result within the roundings
    of the partials."""
    from zhilight_amd.moe import Fp8BlockMoE
    g = torch.Generator(device="cpu").manual_seed(300 + tokens + world)
    e, k, dim, ff, dtype = 32, 2, 256, 256, torch.bfloat16

    def codes(*shape):
        return (torch.randint(0, 0x78, shape, generator=g, dtype=torch.int32) | (torch.randint(0, 2, shape, generator=g, dtype=torch.int32) << 7)) \
            .to(torch.uint8).to(dev)

    def scales(*shape):
        return (torch.rand(shape, generator=g) * 0.008 + 0.002).to(dev)

    router = (torch.randn(e, dim, generator=g) * 0.5).to(dtype).to(dev)
    w_in, s_in, w_g, s_g = codes(e, ff, dim), scales(e, ff // 128, dim // 128), codes(e, ff, dim), scales(e, ff // 128, dim // 128)
    w_out, s_out = codes(e, dim, ff), scales(e, dim // 128, ff // 128)
    ffs = 128 * world                                                          # the shared expert: one 128-row block of dim_ff per rank
    sh = (codes(ffs, dim), scales(ffs // 128, dim // 128), codes(ffs, dim), scales(ffs // 128, dim // 128), codes(dim, ffs), scales(dim // 128, ffs // 128))
    x = torch.randn(tokens, dim, generator=g).to(dtype).to(dev)
    one = Fp8BlockMoE(router, w_in, s_in, w_g, s_g, w_out, s_out, top_k=k)
    want_routed = one.forward(x).float()
    one.shared = sh
    want = one.forward(x).float()
    ids = one.route(x)[0].cpu().numpy()
    parts_routed, parts = [], []
    for r in range(world):
        sl = slice(r, None, world)
        moe = Fp8BlockMoE(router, w_in[sl].contiguous(), s_in[sl].contiguous(), w_g[sl].contiguous(), s_g[sl].contiguous(), w_out[sl].contiguous(),
                          s_out[sl].contiguous(), top_k=k, world_size=world, rank=r)
        parts_routed.append(moe.forward(x))
        blk = slice(128 * r, 128 * (r + 1))
        moe.shared = (sh[0][blk].contiguous(), sh[1][r:r + 1].contiguous(), sh[2][blk].contiguous(), sh[3][r:r + 1].contiguous(),
                      sh[4][:, blk].contiguous(), sh[5][:, r:r + 1].contiguous())
        parts.append(moe.forward(x))
    one.shared = None
    bits_one = one.forward(x).view(torch.int16)
    seen_single = 0
    for t in range(tokens):
        owners = {int(i) % world for i in ids[t]}
        if len(owners) == 1:                                                   # all of the token's experts on one rank
            seen_single += 1
            own = owners.pop()
            for r in range(world):
                row = parts_routed[r][t]
                if r == own:
                    assert torch.equal(row.view(torch.int16), bits_one[t]), (t, r)
                else:
                    assert not row.float().abs().any(), (t, r)
    assert seen_single > 0 or tokens < 8
    # bf16 roundings (half an ulp = 2^-8 relative at worst): per rank the routed partial, the shared partial and their sum; on the
    # one-rank side the routed sum, the shared output and the total.  (A wrong expert, weight or row would be off by O(1) relative.)
    mag = sum(pr_.float().abs() + (p_.float() - pr_.float()).abs() for pr_, p_ in zip(parts_routed, parts))
    for pr, w_, m_ in ((parts_routed, want_routed, sum(p_.float().abs() for p_ in parts_routed)), (parts, want, mag)):
        total = sum(p_.float() for p_ in pr)
        bar = 2.0 ** -6 * m_ + 2.0 ** -7 * (want_routed.abs() + (want - want_routed).abs()) + 1e-6
        bad = (total - w_).abs() > bar
        assert not bool(bad.any()), (int(bad.sum()), float(((total - w_).abs() / (m_ + 1e-9)).max()))
        assert float(w_.abs().max()) > 0


@pytest.mark.parametrize("world", [1, 2])
def test_moe_feed_forward_load_balanced_shared_expert(dev, world):
    """MOE_DYN_SHARED of the reference (feedforward.cpp:268-276 pseudo expert ids, :459-462 route_shared_lb): every rank keeps a copy of the
    shared expert behind its routed ones, the router writes top_k + 1 slots (the extra one with weight 1) and route_shared_lb gives each
    token's shared slot to the first rank with spare capacity.  The rank partials must add up to the one-rank flow with the STATIC shared
    expert, every token's shared slot must name a pseudo expert, and the ranks' loads must be level (the point of the exercise)."""
    from zhilight_amd.moe import Fp8BlockMoE
    g = torch.Generator(device="cpu").manual_seed(500 + world)
    e, k, dim, ff, tokens, dtype = 16, 2, 256, 256, 45, torch.bfloat16

    def codes(*shape):
        return (torch.randint(0, 0x78, shape, generator=g, dtype=torch.int32) | (torch.randint(0, 2, shape, generator=g, dtype=torch.int32) << 7)) \
            .to(torch.uint8).to(dev)

    def scales(*shape):
        return (torch.rand(shape, generator=g) * 0.008 + 0.002).to(dev)

    router = (torch.randn(e, dim, generator=g) * 0.5).to(dtype).to(dev)
    w_in, s_in, w_g, s_g = codes(e, ff, dim), scales(e, ff // 128, dim // 128), codes(e, ff, dim), scales(e, ff // 128, dim // 128)
    w_out, s_out = codes(e, dim, ff), scales(e, dim // 128, ff // 128)
    sh = (codes(ff, dim), scales(ff // 128, dim // 128), codes(ff, dim), scales(ff // 128, dim // 128), codes(dim, ff), scales(dim // 128, ff // 128))
    x = torch.randn(tokens, dim, generator=g).to(dtype).to(dev)
    one = Fp8BlockMoE(router, w_in, s_in, w_g, s_g, w_out, s_out, top_k=k, shared=sh)
    want = one.forward(x).float()
    one.shared = None
    want_routed = one.forward(x).float()
    parts = []
    for r in range(world):
        sl = slice(r, None, world)
        stack = lambda routed, extra: torch.cat([routed[sl], extra[None]], dim=0).contiguous()
        moe = Fp8BlockMoE(router, stack(w_in, sh[0]), stack(s_in, sh[1]), stack(w_g, sh[2]), stack(s_g, sh[3]), stack(w_out, sh[4]), stack(s_out, sh[5]),
                          top_k=k, world_size=world, rank=r, dyn_shared=1)
        if r == 0:
            ids, w, loads = moe.route(x)
            ids, w, loads = ids.cpu().numpy(), w.cpu().numpy(), loads.cpu().numpy()
            assert ids.shape == (tokens, k + 1) and (w[:, k] == 1.0).all()
            assert (ids[:, :k] < e).all() and (ids[:, k] >= e).all() and (ids[:, k] < e + world).all()
            rank_loads = loads[e + world:]
            assert rank_loads.sum() == tokens * (k + 1) and loads[:e + world].sum() == tokens * (k + 1)
            routed_only = np.array([(ids[:, :k] % world == rr).sum() for rr in range(world)])
            assert (rank_loads >= routed_only).all()
            assert rank_loads.max() - rank_loads.min() <= max(1, routed_only.max() - routed_only.min())      # the shared slots level the ranks
        parts.append(moe.forward(x))
    total = sum(p_.float() for p_ in parts)
    mag = sum(p_.float().abs() for p_ in parts) + want_routed.abs() + (want - want_routed).abs()
    bad = (total - want).abs() > 2.0 ** -6 * mag + 1e-6
    assert not bool(bad.any()), (int(bad.sum()), float(((total - want).abs() / (mag + 1e-9)).max()))
    assert float((want - want_routed).abs().max()) > 0


@pytest.mark.parametrize("scoring", ["softmax", "sigmoid", "linear"])
def test_top_k_router_ties(oracle, dev, scoring):
    """logits from a handful of levels: most of the top-k boundary is a TIE.  The reference's insertion sort never lets a later
    equal value displace an earlier one (ff_kernel.cu:98-124) -- equal scores rank by expert index -- which is the integer order
    of the (score, ~index) keys the wave64 router selects by."""
    from zhilight_amd import ops
    rng = np.random.default_rng(5)
    tokens, experts, k = 64, 128, 8
    levels = np.array([-2.0, -0.5, 0.0, 0.25, 1.0, 1.0, 3.0], np.float32)
    logits = torch.from_numpy(levels[rng.integers(0, len(levels), (tokens, experts))]).to(torch.float16)
    logits[0, :] = 0.75                                               # every expert equal: the first k indices, in order
    v, idx = ops.moe_top_k_softmax(logits.to(dev), k, k, True, 1.0, scoring, None, None, 1)
    wv, widx, _, _ = oracle.moe_top_k_softmax(_bits(logits), k, k, True, 1.0, scoring, 0, 0)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert np.array_equal(idx.cpu().numpy()[0], np.arange(k))
    assert _ulps(v.cpu().numpy(), wv).max() <= 8


@pytest.mark.parametrize("experts,groups,topk_group,k,bias", [(256, 8, 4, 8, True), (256, 8, 4, 8, False), (64, 4, 2, 4, False)])
def test_group_limited_router_ties(oracle, dev, experts, groups, topk_group, k, bias):
    """the same for the group-limited router: tied GROUP scores (kept by smaller group index) and tied expert scores inside and
    across the kept groups (smaller expert index first, the bitonic compare-exchange's rule, ff_kernel.cu:273-291)"""
    from zhilight_amd import ops
    rng = np.random.default_rng(experts + k)
    tokens = 48
    levels = np.array([-1.0, 0.0, 0.5, 0.5, 2.0], np.float32)
    logits = torch.from_numpy(levels[rng.integers(0, len(levels), (tokens, experts))]).to(torch.float16)
    logits[0, :] = 0.5                                                # all groups tie, all experts tie
    b = (np.round(rng.standard_normal(experts) * 2) * 0.125).astype(np.float32) if bias else None     # a coarse bias keeps ties alive
    v, idx = ops.moe_group_topk(logits.to(dev), None if b is None else _t(b, dev), groups, topk_group, k, k, True, 2.5, "sigmoid", None, None, 1)
    wv, widx, _, _ = oracle.moe_group_topk(_bits(logits), b, k, groups, topk_group, k, True, 2.5, "sigmoid", 0, 0)
    assert np.array_equal(idx.cpu().numpy(), widx)
    if not bias:
        assert np.array_equal(idx.cpu().numpy()[0], np.arange(k))     # groups 0 .. topk_group - 1 kept, their lowest indices win
    assert _ulps(v.cpu().numpy(), wv).max() <= 8


def test_per_token_cast_more_rows_than_a_grid_dimension(oracle, dev):
    """a DeepSeek-V3 prefill of 8K+ tokens casts tokens x top_k (+ padding) > 65535 grouped rows; the rows ride on grid.x
    now (the reference launches grid (m, n / 128) as well, fp8_util.cu:277-322)"""
    from zhilight_amd import ops
    m, n = 65536 + 77, 256
    x = (torch.randn(m, n) * 3).to(torch.bfloat16)
    codes, scales = ops.fp8_per_token_cast(x.to(dev))
    want_c, want_s = oracle.fp8_per_token_cast(_bits(x), col_major=True, dtype=1)
    assert np.array_equal(codes.cpu().numpy(), want_c)
    assert np.array_equal(scales.cpu().numpy()[:, :m], want_s[:, :m])


@pytest.mark.parametrize("n,max_key", [(1, 7), (255, 16), (256, 512), (4097, 300), (32768, 2304), (100000, 0)])
def test_dispatch_index_helpers(dev, n, max_key):
    """functions::arange / sort_pair_1d / divide / scatter_update_dim0 of the MoE dispatch route (feedforward.cpp:599-629, 1040-1075) as
    kernels: the sort is STABLE (equal keys keep their input order -- the dispatch relies on it), against numpy's stable argsort."""
    from zhilight_amd import ops
    rng = np.random.default_rng(n)
    hi = max_key if max_key else 2 ** 31 - 1
    keys = rng.integers(0, hi + 1 if max_key else hi, n).astype(np.int32)
    if n > 10:
        keys[:5] = hi if max_key else hi - 1                                     # the largest key, and ties at the front
    vals = ops.arange_i32(n, dev)
    assert np.array_equal(vals.cpu().numpy(), np.arange(n, dtype=np.int32))
    ko, vo = ops.sort_pairs_i32(_t(keys, dev), vals, max_key=max_key)
    order = np.argsort(keys, kind="stable").astype(np.int32)
    assert np.array_equal(ko.cpu().numpy(), keys[order]) and np.array_equal(vo.cpu().numpy(), order)
    assert np.array_equal(ops.divide_i32(vo, 6).cpu().numpy(), order // 6)
    if max_key == 0 and n > 10:
        # no key bound: the full signed order of cub's int32 radix sort -- negative keys (an unfilled -1 expert id) first, stable
        sk = keys.copy()
        sk[::7] = -1
        sk[3] = -(2 ** 31)
        ko2, vo2 = ops.sort_pairs_i32(_t(sk, dev), vals, max_key=0)
        order2 = np.argsort(sk, kind="stable").astype(np.int32)
        assert np.array_equal(ko2.cpu().numpy(), sk[order2]) and np.array_equal(vo2.cpu().numpy(), order2)
        ko3, vo3 = ops.sort_pairs_i32(_t(sk, dev), vals, max_key=-1)              # a negative bound means the same
        assert torch.equal(ko3, ko2) and torch.equal(vo3, vo2)
    assert np.array_equal(ops.arange_i32(5, dev, start=3, step=4).cpu().numpy(), np.array([3, 7, 11, 15, 19], np.int32))
    # scatter: dst rows named by dst_index <- src rows named by src_index (repeats allowed on the source side)
    m = min(n, 300)
    for dtype, width in ((torch.uint8, 512), (torch.float32, 7 * 2), (torch.float16, 129)):
        src = torch.from_numpy(rng.integers(0, 250, (97, width)).astype(np.float32)).to(dtype).to(dev)
        dst = torch.zeros((m + 11, width), dtype=dtype, device=dev)
        di = rng.permutation(m + 11)[:m].astype(np.int32)
        si = rng.integers(0, 97, m).astype(np.int32)
        ops.scatter_update_dim0(dst, _t(di, dev), src, _t(si, dev))
        want = np.zeros((m + 11, width), np.float32)
        want[di] = src.float().cpu().numpy()[si]
        assert np.array_equal(dst.float().cpu().numpy(), want)
        dst2 = torch.zeros((m + 11, width), dtype=dtype, device=dev)
        mm = min(m, 50)
        ops.scatter_update_dim0(dst2, _t(di[:mm], dev), src[:mm].contiguous())           # no source index: row i
        want2 = np.zeros((m + 11, width), np.float32)
        want2[di[:mm]] = src.float().cpu().numpy()[:mm]
        assert np.array_equal(dst2.float().cpu().numpy(), want2)


@pytest.mark.parametrize("dtype,code", [(torch.bfloat16, 1), (torch.float16, 0)])
@pytest.mark.parametrize("m,n,k", [(17, 256, 1024), (32, 4096, 7168), (25, 200, 2048), (32, 512, 4096), (20, 144, 8192), (32, 128, 1536 + 512)])
def test_block_gemm_17_32_rows_with_lds_dma_activations(oracle, dev, dtype, code, m, n, k):
    """k_fp8_block_gemm_dma (round 6: 17..32 rows, the wave's activation blocks moved global -> LDS by LDS-DMA, XOR-swizzled): a row of
    the product depends on its own row of the activations only and a wave walks its blocks in the same order, so the launch returns the
    BITS of the 16-row launches (k_fp8_block_gemm<1, ..>) on the two halves of the rows -- 1, 2, 4, 7, 8 blocks per wave, a ragged K
    (the last round of blocks partly past the end), ragged N and M; and it sits inside the format's bar."""
    from zhilight_amd import ops
    rng = np.random.default_rng(m * 31 + n)
    x = _acts(rng, m, k, dtype)
    a8, sa = oracle.fp8_per_token_cast(_bits(x), dtype=code)
    w8, sw = _block_weight(rng, n, k)
    a8_t, w8_t, sw_t = _t(a8, dev), _t(w8, dev), _t(sw, dev)
    got = ops.fp8_block_gemm(a8_t, _t(sa, dev), w8_t, sw_t, dtype=dtype)
    for lo, hi in ((0, 16), (16, m)):
        a8h, sah = oracle.fp8_per_token_cast(_bits(x[lo:hi]), dtype=code)
        assert np.array_equal(a8h, a8[lo:hi])
        part = ops.fp8_block_gemm(_t(a8h, dev), _t(sah, dev), w8_t, sw_t, dtype=dtype)
        assert torch.equal(got[lo:hi], part), (lo, hi)
    if n * k <= 512 * 4096:
        _gemm_bar(_bits(got), oracle.fp8_block_gemm(a8, sa, w8, sw, dtype=code), oracle, code)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k", [(1, 512, 1024), (5, 200, 512), (16, 4096, 7168), (17, 256, 1024), (32, 2048, 7168), (25, 200, 2048), (32, 144, 1536)])
def test_block_gemm_on_the_packed_weight_layout(oracle, dev, dtype, m, n, k):
    """ZLF8M (zl_fp8_block_pack + zl_fp8_block_gemm_group_packed): the same kernels reading the weight codes in 1 KiB contiguous
    fragment loads -- the bits of the launch on the row-major codes; 1..16 rows, 17..32 rows (LDS-DMA activations, and the block
    counts per wave that kernel is not built for), ragged N."""
    from zhilight_amd import ops
    rng = np.random.default_rng(m * 13 + n)
    code = 1 if dtype == torch.bfloat16 else 0
    x = _acts(rng, m, k, dtype)
    a8, sa = oracle.fp8_per_token_cast(_bits(x), dtype=code)
    w8, sw = _block_weight(rng, n, k)
    w8_t, sw_t = _t(w8, dev), _t(sw, dev)
    wp = ops.Fp8BlockMWeight(w8_t)
    want = ops.fp8_block_gemm(_t(a8, dev), _t(sa, dev), w8_t, sw_t, dtype=dtype)
    assert torch.equal(ops.fp8_block_gemm(_t(a8, dev), _t(sa, dev), wp, sw_t, dtype=dtype), want)


def test_grouped_block_gemm_on_the_packed_weight_layout(oracle, dev):
    """the m-grouped form (an expert per 64-row run, padding rows untouched) on packed (G, N, K) codes: the bits of the row-major launch"""
    from zhilight_amd import ops
    rng = np.random.default_rng(77)
    G, n, k, m = 4, 256, 1024, 256
    x = _acts(rng, m, k, torch.bfloat16)
    a8, sa = oracle.fp8_per_token_cast(_bits(x), dtype=1)
    ws = [_block_weight(rng, n, k) for _ in range(G)]
    w8 = _t(np.stack([w[0] for w in ws]), dev)
    sw = _t(np.stack([w[1] for w in ws]), dev)
    idx = np.repeat(np.array([2, -1, 0, 3], np.int32), 64)
    want = ops.fp8_block_gemm(_t(a8, dev), _t(sa, dev), w8, sw, m_indices=_t(idx, dev))
    got = ops.fp8_block_gemm(_t(a8, dev), _t(sa, dev), ops.Fp8BlockMWeight(w8), sw, m_indices=_t(idx, dev))
    assert torch.equal(got, want)

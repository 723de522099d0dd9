"""Decode batches of 5..32 rows on digit planes (zl_w4a16_planes + zl_w4a16_gemm_planes / _qkv_rope_scatter_planes; the I8
instantiations of k_w4a16_phase): the activation matrix is converted ONCE, every workgroup consumes the planes on the integer
matrix cores.  Oracle: the exact fp64 product of the fp16 activations with (q - z) s (gptq_gemm_k_major_exact; the reference's
branch for these row counts is q_gemm_k_major.cu:580-686); bars = fp16 output rounding, as for the 1..4-row kernel."""
import zlib

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


# off the product path since round 5 (include/zhilight_amd.h, #ifdef ZL_EXPERIMENTAL): built only by ZL_BUILD_EXPERIMENTAL=1 python -m zhilight_amd.build
@pytest.fixture(autouse=True)
def _experimental_build_only():
    from zhilight_amd import ops
    if not ops.experimental_build():
        pytest.skip("not a ZL_BUILD_EXPERIMENTAL=1 build of libzhilight_amd.so")


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.view(dtype)
    return t.to(dev)


def _np(t):
    return t.detach().cpu().numpy()


def _weight(oracle, dev, rng, k, n, g=128):
    from zhilight_amd import ops
    qw, qz, sc = synth.gptq_hf(rng, k, n, g)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, g)
    return km, ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g)


def _close(got, exact, extra=0.0):
    rms = np.sqrt((exact ** 2).mean())
    d = np.abs(got - exact)
    assert (d <= 2.0 ** -10 * np.abs(exact) + (2e-5 + extra) * rms).all(), float((d / rms).max())


@pytest.mark.parametrize("m", [5, 8, 13, 16, 17, 24, 32])
@pytest.mark.parametrize("k,n", [(1024, 272), (4096, 512), (14336, 256), (1152, 40)])
def test_planes_gemm_rows_and_shapes(oracle, dev, m, k, n):
    from zhilight_amd import ops
    rng = np.random.default_rng(zlib.crc32(f"{m}-{k}-{n}".encode()))
    km, w = _weight(oracle, dev, rng, k, n)
    x = synth.act(rng, m, k)
    assert ops.w4_planes_ok(m, k)
    planes = ops.w4_planes(_t(x, dev))
    got = _np(ops.w4_linear_planes(planes, m, w)).astype(np.float64)
    exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km)
    _close(got, exact)
    # and against the fp16-dequant phase kernel the route replaces: both within output rounding of the exact product
    want = _np(ops.w4a16_gemm_mfma(_t(x, dev), w)).astype(np.float64)
    assert np.abs(got - want).max() <= 2.0 ** -9 * np.abs(exact).max()


@pytest.mark.parametrize("rounds", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("m", [8, 32])
def test_planes_gemm_tiles_per_workgroup(oracle, dev, m, rounds, monkeypatch):
    """every R instantiation (tiles per workgroup), both row-block counts; N not a multiple of 16 R"""
    from zhilight_amd import ops
    rng = np.random.default_rng(300 + rounds + m)
    k, n = 2048 + 128, 16 * (3 * rounds + 1) + 8
    km, w = _weight(oracle, dev, rng, k, n)
    x = synth.act(rng, m, k)
    monkeypatch.setenv("ZL_W4_PHASE_ROUNDS", str(rounds))
    got = _np(ops.w4_linear_planes(ops.w4_planes(_t(x, dev)), m, w)).astype(np.float64)
    _close(got, oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km))


@pytest.mark.parametrize("m", [6, 16, 19, 32])
@pytest.mark.parametrize("ksplit", [2, 4])
def test_planes_gemm_k_split_bias_residual(oracle, dev, m, ksplit, monkeypatch):
    """the down projection's shape class (few tiles, long K): K split over workgroups, bias + residual epilogue; repeated, so a
    counter left non-zero by one launch would show in the next"""
    from zhilight_amd import ops
    rng = np.random.default_rng(410 + m + ksplit)
    k, n = 14336, 1024
    km, w = _weight(oracle, dev, rng, k, n)
    monkeypatch.setenv("ZL_W4_PHASE_KSPLIT", str(ksplit))
    bias = (rng.standard_normal(n) * 0.1).astype(np.float16)
    for it in range(3):
        x = synth.act(rng, m, k)
        res = rng.standard_normal((m, n)).astype(np.float16)
        got = _np(ops.w4_linear_planes(ops.w4_planes(_t(x, dev)), m, w, bias=_t(bias, dev), residual=_t(res, dev),
                                       epilogue=ops.EPI_RESIDUAL)).astype(np.float64)
        exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km)
        lin = (exact + bias.astype(np.float64)).astype(np.float16).astype(np.float64)
        ref = (res.astype(np.float64) + lin).astype(np.float16).astype(np.float64)
        assert np.abs(got - ref).max() <= 2.0 ** -9 * max(1.0, np.abs(ref).max()), it


@pytest.mark.parametrize("m", [8, 24])
def test_planes_fused_norm_and_silu_mul(oracle, dev, m):
    """the gate|up launch of a decode layer: RMSNorm inside the plane conversion, silu(gate) * up in the epilogue"""
    from zhilight_amd import ops
    rng = np.random.default_rng(62 + m)
    k, nff, g = 2048, 200, 128
    qw1, qz1, sc1 = synth.gptq_hf(rng, k, nff, g)
    qw2, qz2, sc2 = synth.gptq_hf(rng, k, nff, g)
    km1, km2 = oracle.gptq_prepare_k_major(qw1, qz1, sc1, g), oracle.gptq_prepare_k_major(qw2, qz2, sc2, g)
    cat = [np.concatenate([a, b], axis=0) for a, b in zip(km1, km2)]
    w = ops.W4MWeight.from_k_major(_t(cat[0].view(np.int32), dev), _t(cat[1], dev), _t(cat[2], dev, torch.float16), g,
                                   row_interleave=True)
    x = synth.act(rng, m, k, 2.0)
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    xn = oracle.u2h(oracle.rmsnorm(oracle.h2u(x), oracle.h2u(nw), 1e-5))
    ge = oracle.gptq_gemm_k_major_exact(oracle.h2u(xn), *km1).astype(np.float16)
    ue = oracle.gptq_gemm_k_major_exact(oracle.h2u(xn), *km2).astype(np.float16)
    ref = oracle.u2h(oracle.silu_mul(oracle.h2u(ge), oracle.h2u(ue))).astype(np.float64)
    planes = ops.w4_planes(_t(x, dev), norm_weight=_t(nw, dev), norm_eps=1e-5)
    got = _np(ops.w4_linear_planes(planes, m, w, epilogue=ops.EPI_SILU_MUL)).astype(np.float64)
    assert got.shape == (m, nff)
    # the normalised row may differ from the oracle's by an fp16 ulp in a few places (summation order of the squares), a gate / up
    # value on a rounding tie by one more
    assert np.abs(got - ref).max() <= 2.0 ** -8 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("m", [6, 16, 17, 32])
@pytest.mark.parametrize("bshd", [True, False])
def test_planes_qkv_rotary_scatter(oracle, dev, m, bshd):
    """zl_w4a16_qkv_rope_scatter_planes against zl_w4a16_gemm_planes + zl_rope_scatter_decode on the same planes, bit for bit
    (the epilogue is the fp16 route's), and the projection itself against the exact product"""
    from zhilight_amd import ops
    rng = np.random.default_rng(80 + m)
    h, hkv, d, k, g = 8, 2, 128, 1024 + 256, 128
    n = (h + 2 * hkv) * d
    km, w = _weight(oracle, dev, rng, k, n)
    x = _t(synth.act(rng, m, k, 2.0), dev)
    bias = _t((rng.standard_normal(n) * 0.1).astype(np.float16), dev)
    nw = _t((1.0 + 0.1 * rng.standard_normal(k)).astype(np.float16), dev)
    lens = [int(v) for v in rng.integers(2, 6, m) * 32]
    pos = np.array([int(rng.integers(0, L)) for L in lens], np.int32)
    pos[0] = lens[0] - 1
    placement = pos.copy()
    placement[1] = -1
    cs, sn = oracle.rope_cos_sin(pos, d, 5e5, True, (8.0, 1.0, 4.0, 8192.0))
    shape = (lambda L: (L, hkv, d)) if bshd else (lambda L: (hkv, L, d))
    mk = lambda: [torch.full(shape(L), 3.0, dtype=torch.float16, device=dev) for L in lens]
    k1, v1, k2, v2 = mk(), mk(), mk(), mk()
    lens_t, place_t = _t(np.array(lens, np.int32), dev), _t(placement, dev)
    planes = ops.w4_planes(x, norm_weight=nw, norm_eps=1e-5)
    qkv = ops.w4_linear_planes(planes, m, w, bias=bias)
    q_ref = ops.rope_scatter_decode(_t(cs, dev), _t(sn, dev), qkv, place_t, lens_t, ops.make_ptr_table(k1), ops.make_ptr_table(v1),
                                    h, hkv, d, True, bshd)
    q_got = ops.w4_qkv_rope_scatter_planes(planes, m, w, _t(cs, dev), _t(sn, dev), place_t, lens_t, ops.make_ptr_table(k2),
                                           ops.make_ptr_table(v2), h, hkv, d, bias=bias, bshd=bshd)
    assert torch.equal(q_got, q_ref)
    for a, b_ in zip(k1 + v1, k2 + v2):
        assert torch.equal(a, b_)
    assert not torch.equal(k1[0], torch.full_like(k1[0], 3.0))
    want = ops.w4a16_gemm_mfma(ops.rmsnorm(x, nw, 1e-5), w, bias=bias)
    assert (qkv.float() - want.float()).abs().max().item() <= 2.0 ** -8 * want.float().abs().max().item()


@pytest.mark.parametrize("m,k", [(8, 4096), (20, 14336)])
@pytest.mark.parametrize("case", ["outlier100", "outlier1000", "subnormal_group", "max_half", "mixed", "nonfinite"])
def test_planes_activation_edge_cases(oracle, dev, m, k, case):
    """the block-floating image on inputs an N(0, 1) draw never produces (tests/test_gpu_w4.py::test_i8p_activation_edge_cases'
    cases and bar), and inf / NaN: every output that reads a group with a non-finite value is NaN, the other rows untouched"""
    from zhilight_amd import ops
    rng = np.random.default_rng(zlib.crc32(f"planes-{m}-{k}-{case}".encode()))
    n = 256
    km, w = _weight(oracle, dev, rng, k, n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    if case in ("outlier100", "outlier1000", "mixed"):
        cols = rng.choice(k, size=k // 64, replace=False)
        x[:, cols] *= 100.0 if case == "outlier100" else 1000.0
    if case in ("subnormal_group", "mixed"):
        g0 = 128 * int(rng.integers(0, k // 128))
        x[:, g0:g0 + 128] = rng.uniform(-6e-5, 6e-5, (m, 128))
        x[0, g0 + 5] = 5.96e-8
    if case in ("max_half", "mixed"):
        cols = rng.choice(k, size=8, replace=False)
        x[:, cols] = 65504.0 * np.sign(rng.standard_normal((m, 8)))
    xh = np.clip(x, -65504, 65504).astype(np.float16)
    if case == "nonfinite":
        xh[1, 300] = np.inf
        xh[m - 1, k - 1] = np.nan
    got = _np(ops.w4_linear_planes(ops.w4_planes(_t(xh, dev)), m, w)).astype(np.float64)
    if case == "nonfinite":
        assert np.isnan(got[1]).all() and np.isnan(got[m - 1]).all()
        keep = [r for r in range(m) if r not in (1, m - 1)]
        xh, got = xh[keep], got[keep]
    exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(xh), *km)
    rms = np.sqrt((exact ** 2).mean())
    fin = np.abs(exact) < 65504.0 * (1 - 2.0 ** -11)
    d = np.abs(got - exact)
    assert (d[fin] <= 2.0 ** -10 * np.abs(exact[fin]) + 2e-5 * rms).all(), float((d[fin] / rms).max())
    assert np.isinf(got[~fin]).all()


def test_planes_refusals(dev):
    from zhilight_amd import ops
    from zhilight_amd._lib import ZLError
    with pytest.raises(ZLError):
        ops.w4_planes(torch.zeros(33, 1024, dtype=torch.float16, device=dev))
    with pytest.raises(ZLError):
        ops.w4_planes(torch.zeros(8, 1024 + 64, dtype=torch.float16, device=dev))
    with pytest.raises(ZLError):
        ops.w4_planes(torch.zeros(8, 32768, dtype=torch.float16, device=dev))
    assert not ops.w4_planes_ok(4, 4096) and not ops.w4_planes_ok(33, 4096) and ops.w4_planes_ok(5, 128)

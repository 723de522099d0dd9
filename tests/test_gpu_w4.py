"""GPU parity: W4A16 path through the C ABI vs the CPU oracle.

Bars: integer layout work bit-exact; W4A16 GEMM bit-exact vs the reference-faithful (R) oracle
(the kernel replays the reference's per-lane chains and 32-lane trees), and within
1e-3 of the output rms vs the exact (E) fp64 oracle (north_star tolerance: logits within 1e-3 rel)."""
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


@pytest.mark.parametrize("k,n,g", [(1024, 256, 128), (4096, 1024, 128), (2048, 264, 64), (1024, 64, 32)])
def test_layout_transforms_bit_exact(oracle, dev, k, n, g):
    from zhilight_amd import ops
    rng = np.random.default_rng(0)
    qw, qz, sc = synth.gptq_hf(rng, k, n, g)
    d_qw = ops.gptq_shuffle(_t(qw.view(np.int32), dev))
    assert np.array_equal(_np(d_qw).view(np.uint32), oracle.gptq_shuffle(qw))
    d_qz = ops.increase_zero(_t(qz.view(np.int32), dev))
    o_qz = oracle.gptq_increase_zero(qz)
    assert np.array_equal(_np(d_qz).view(np.uint32), o_qz)
    d_q8 = ops.q4_to_q8(d_qz)
    assert np.array_equal(_np(d_q8), oracle.gptq_q4_to_q8(o_qz))
    for arr, dt in ((qw.view(np.int32), np.int32), (sc.view(np.int16), np.int16), (_np(d_q8), np.uint8)):
        assert np.array_equal(_np(ops.transpose_2d(_t(arr, dev))), np.ascontiguousarray(arr.T))
    # full wildcard nibble pattern incl. 0xF wrap
    allz = np.arange(0, 2 ** 16, dtype=np.uint32)
    allz = (allz | (allz << 16)).astype(np.uint32).reshape(256, 256)
    assert np.array_equal(_np(ops.increase_zero(_t(allz.view(np.int32), dev))).view(np.uint32),
                          oracle.gptq_increase_zero(allz))


def test_awq_transforms_bit_exact(oracle, dev):
    from zhilight_amd import ops
    rng = np.random.default_rng(1)
    k, n = 256, 128
    q = rng.integers(0, 2 ** 32, size=(k, n // 8), dtype=np.uint64).astype(np.uint32)
    for ex in (True, False):
        out = ops.shuffle_awq(_t(q.view(np.int32), dev), ex)
        assert np.array_equal(_np(out).view(np.uint32), oracle.awq_shuffle(q, ex))
    z = rng.integers(0, 2 ** 32, size=(k // 128, n // 8), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(_np(ops.awq_un_shuffle(_t(z.view(np.int32), dev))).view(np.uint32), oracle.awq_un_shuffle(z))


def _make(oracle, dev, rng, k, n, g, sym=False, interleave=False):
    """k-major oracle tensors (first n rows of an n-rounded-to-8 HF matrix) + the packed device weight.
    When n % 8 == 0 the device weight goes through the full HF load path (shuffle, +1, q4->q8, transposes)."""
    from zhilight_amd import ops
    n8 = (n + 7) // 8 * 8
    qw, qz, sc = synth.gptq_hf(rng, k, n8, g, sym)
    km = tuple(np.ascontiguousarray(a[:n]) for a in oracle.gptq_prepare_k_major(qw, qz, sc, g))
    if n8 == n:
        w = ops.W4Weight.from_hf_gptq(_t(qw.view(np.int32), dev), _t(qz.view(np.int32), dev), _t(sc, dev, torch.float16),
                                      g, sym=sym, row_interleave=interleave)
    else:
        w = ops.W4Weight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g,
                                      sym=sym, row_interleave=interleave)
    return km, w


@pytest.mark.parametrize("k,n,g", [(1024, 256, 128), (4096, 512, 128), (2048, 130, 64), (3072, 64, 32),
                                   (1536, 96, 128), (1024, 32, 1024)])
def test_pack_and_dequant_bit_exact(oracle, dev, k, n, g):
    rng = np.random.default_rng(2)
    km, w = _make(oracle, dev, rng, k, n, g)
    ref = oracle.gptq_dequant_k_major(*km)
    got = _np(w.dequant()).view(np.uint16)
    assert np.array_equal(got, ref)
    # and the k-major dequant equals the format definition (q - (z+1)) * s rounded to fp16
    n8 = (n + 7) // 8 * 8
    qw, qz, sc = synth.gptq_hf(np.random.default_rng(2), k, n8, g)
    naive = oracle.gptq_dequant_hf_naive(qw, qz, sc, g).astype(np.float16)[:n]
    assert np.array_equal(oracle.u2h(ref), naive)


def _check_gemm(oracle, dev, k, n, g, m, seed, sym=False, bias=False, add_c=False):
    from zhilight_amd import ops
    rng = np.random.default_rng(seed)
    km, w = _make(oracle, dev, rng, k, n, g, sym)
    x = synth.act(rng, m, k)
    b = (rng.standard_normal(n) * 0.1).astype(np.float16) if bias else None
    c0 = synth.act(rng, m, n) if add_c else None
    ref = oracle.gptq_gemm_k_major(oracle.h2u(x), *km, bias=None if b is None else oracle.h2u(b), sym=sym,
                                   add_c=None if c0 is None else oracle.h2u(c0))
    out = None if c0 is None else _t(c0, dev)
    y = ops.w4a16_gemm(_t(x, dev), w, bias=None if b is None else _t(b, dev), out=out,
                       epilogue=ops.EPI_ADD_C if add_c else 0)
    got = _np(y).view(np.uint16)
    nbad = int((got != ref).sum())
    assert nbad == 0, f"{nbad}/{ref.size} outputs differ from the R oracle (max ulp {synth.ulp_diff_f16(got, ref).max()})"
    if not add_c:
        exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km, bias=None if b is None else oracle.h2u(b), sym=sym)
        err = np.abs(oracle.u2h(got).astype(np.float64) - exact).max() / np.sqrt((exact ** 2).mean())
        assert err < 4e-3, err  # R itself is ~1e-3 rms-relative noisy (fp16 partial dots); see DESIGN.md
    return got


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 7, 8, 17])
def test_gemm_small_bit_exact_vs_R(oracle, dev, m):
    _check_gemm(oracle, dev, 2048, 264, 128, m, seed=10 + m)


@pytest.mark.parametrize("k,n,g", [(1024, 2, 128), (1024, 3, 128), (1536, 70, 128), (2048, 66, 64), (1024, 34, 32),
                                   (2304, 96, 128), (1024, 40, 1024), (5120, 8, 128)])
def test_gemm_ragged_shapes(oracle, dev, k, n, g):
    _check_gemm(oracle, dev, k, n, g, 1, seed=3)
    _check_gemm(oracle, dev, k, n, g, 3, seed=4)


def test_gemm_sym_bias_addc(oracle, dev):
    _check_gemm(oracle, dev, 2048, 128, 128, 2, seed=5, sym=True)
    _check_gemm(oracle, dev, 2048, 128, 128, 2, seed=6, bias=True)
    _check_gemm(oracle, dev, 2048, 128, 128, 3, seed=7, bias=True, add_c=True)


@pytest.mark.parametrize("k,n", [(4096, 6144), (4096, 4096), (14336, 4096), (4096, 28672)])
def test_gemm_llama3_shapes_m1(oracle, dev, k, n):
    """BASELINE config 2 shapes (Llama-3-8B GPTQ-Int4, batch 1): bit-exact vs the R oracle."""
    _check_gemm(oracle, dev, k, n, 128, 1, seed=0)


def test_gemm_residual_and_norm_prologue(oracle, dev):
    from zhilight_amd import ops
    rng = np.random.default_rng(8)
    k, n, g, m = 2048, 256, 128, 2
    km, w = _make(oracle, dev, rng, k, n, g)
    x = synth.act(rng, m, k, 3.0)
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    res = synth.act(rng, m, n)
    # oracle: rmsnorm -> gemm -> element_add_scale (scale 1)
    xn = oracle.rmsnorm(oracle.h2u(x), oracle.h2u(nw), 1e-5)
    lin = oracle.gptq_gemm_k_major(xn, *km)
    ref = oracle.element_add_scale(oracle.h2u(res), lin, 1.0, True)
    y = ops.w4a16_gemm(_t(x, dev), w, residual=_t(res, dev), norm_weight=_t(nw, dev), norm_eps=1e-5,
                       epilogue=ops.EPI_RESIDUAL)
    got = _np(y).view(np.uint16)
    ulp = synth.ulp_diff_f16(got, ref)
    # the block-wide sum of squares is associated differently from the reference's 1024-thread
    # tree -> the normalised input may differ by 1 fp16 ulp on a few elements
    assert ulp.max() <= 2 and (ulp > 0).mean() < 0.05, (ulp.max(), (ulp > 0).mean())


@pytest.mark.parametrize("f32", [False, True])
def test_gemm_fused_gate_up_silu(oracle, dev, f32):
    from zhilight_amd import ops
    rng = np.random.default_rng(9)
    k, nff, g, m = 2048, 192, 128, 2
    qw1, qz1, sc1 = synth.gptq_hf(rng, k, nff, g)
    qw2, qz2, sc2 = synth.gptq_hf(rng, k, nff, g)
    km1, km2 = oracle.gptq_prepare_k_major(qw1, qz1, sc1, g), oracle.gptq_prepare_k_major(qw2, qz2, sc2, g)
    cat = [np.concatenate([a, b], axis=0) for a, b in zip(km1, km2)]  # [gate; up] rows
    w = ops.W4Weight.from_k_major(_t(cat[0].view(np.int32), dev), _t(cat[1], dev), _t(cat[2], dev, torch.float16), g,
                                  row_interleave=True)
    x = synth.act(rng, m, k)
    if f32:
        ref = oracle.gptq_gemm_fuse_gate_in(oracle.h2u(x), km1, km2)
    else:
        ref = oracle.silu_mul(oracle.gptq_gemm_k_major(oracle.h2u(x), *km1), oracle.gptq_gemm_k_major(oracle.h2u(x), *km2))
    y = ops.w4a16_gemm(_t(x, dev), w, epilogue=ops.EPI_SILU_MUL_F32 if f32 else ops.EPI_SILU_MUL)
    got = _np(y).view(np.uint16)
    ulp = synth.ulp_diff_f16(got, ref)
    assert ulp.max() <= 1 and (ulp > 0).mean() < 0.01, (ulp.max(), (ulp > 0).mean())  # expf: device vs glibc


def test_gemm_error_behaviour(dev):
    from zhilight_amd import ops
    from zhilight_amd._lib import ZLError
    rng = np.random.default_rng(0)
    qw, qz, sc = synth.gptq_hf(rng, 1024, 64, 128)
    w = ops.W4Weight.from_hf_gptq(_t(qw.view(np.int32), dev), _t(qz.view(np.int32), dev), _t(sc, dev, torch.float16), 128)
    with pytest.raises(ZLError, match="size K mismatch"):
        ops.w4a16_gemm(torch.zeros(1, 2048, dtype=torch.float16, device=dev), w)
    with pytest.raises(ZLError, match="A must be half"):
        ops.w4a16_gemm(torch.zeros(1, 1024, dtype=torch.bfloat16, device=dev), w)


# ---------------------------------------------------------------------------------------------------
# MFMA flavour (fp32 accumulation): checked against the exact fp64 oracle and against the restatement of
# the reference's M > 40 branch (dequant_k_major -> fp16 weights, fp32-accumulating GEMM)
# ---------------------------------------------------------------------------------------------------
def _check_mfma(oracle, dev, k, n, m, seed, bias=False, norm=False, residual=False, force_tiled=False):
    from zhilight_amd import ops
    rng = np.random.default_rng(seed)
    g = 128
    n8 = (n + 7) // 8 * 8
    qw, qz, sc = synth.gptq_hf(rng, k, n8, g)
    km = tuple(np.ascontiguousarray(a[:n]) for a in oracle.gptq_prepare_k_major(qw, qz, sc, g))
    w = ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g)
    x = synth.act(rng, m, k, 2.0 if norm else 1.0)
    b = (rng.standard_normal(n) * 0.1).astype(np.float16) if bias else None
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float16) if norm else None
    res = synth.act(rng, m, n) if residual else None
    xin = oracle.rmsnorm(oracle.h2u(x), oracle.h2u(nw), 1e-5) if norm else oracle.h2u(x)
    exact = oracle.gptq_gemm_k_major_exact(xin, *km, bias=None if b is None else oracle.h2u(b))
    if force_tiled:
        y = ops.w4a16_gemm_tiled(_t(x, dev), w, bias=None if b is None else _t(b, dev), residual=None if res is None else _t(res, dev),
                                 epilogue=ops.EPI_RESIDUAL if residual else 0)
    else:
        y = ops.w4a16_gemm_mfma(_t(x, dev), w, bias=None if b is None else _t(b, dev), residual=None if res is None else _t(res, dev),
                                norm_weight=None if nw is None else _t(nw, dev), norm_eps=1e-5,
                                epilogue=ops.EPI_RESIDUAL if residual else 0)
    got = _np(y).astype(np.float64)
    # m > 16 without a fused norm runs the M-tiled kernel: it multiplies with W16 = rn16(rn16(q - z) * s), the
    # matrix the reference's M > 40 branch dequantises (dequant_k_major), so THAT product is its exact value
    # (up to 32 rows stay on the streaming arithmetic: the phase-pipelined kernel, w4_phase.hip -- K beyond 8192 with
    # 17..32 rows through its K-split variant)
    tiled = force_tiled or (m > 32 and not norm)
    w16 = oracle.gptq_dequant_k_major(*km)
    ref40 = oracle.gemm_nt(xin, w16, None if b is None else oracle.h2u(b), exact=True)
    lin = np.zeros_like(exact)
    if residual:
        # the linear output is rounded to fp16 BEFORE the residual add: a 1-ulp flip of it (fp32 noise on a
        # rounding tie) survives the add, so the bound carries |linear| as well as |sum|
        lin = np.abs(exact)
        exact = res.astype(np.float64) + exact.astype(np.float16).astype(np.float64)
        ref40 = res.astype(np.float64) + ref40.astype(np.float16).astype(np.float64)
    rms = np.sqrt((exact ** 2).mean())
    # fp16 output rounding (2^-11 relative) + fp32 accumulation noise; with the fused norm the normalised
    # input may differ by 1 fp16 ulp on a few elements (block-sum association)
    tight = 2.0 ** -10 * (np.abs(exact) + lin) + (3e-4 if norm else 2e-5) * rms
    # + the W16 weight rounding between the two flavours (rn16 of (q - z) * s: with one or two groups per row the
    # per-group rounding patterns do not average out over K)
    loose = 2.0 ** -10 * (np.abs(exact) + lin) + 1.5e-3 * max(1.0, (1024.0 / k) ** 0.5) * rms
    d_exact, d_40 = np.abs(got - exact), np.abs(got - ref40)
    if tiled:
        assert (d_40 <= 2.0 ** -10 * (np.abs(ref40) + lin) + 2e-5 * rms).all(), float((d_40 / rms).max())
        assert (d_exact <= loose).all(), float((d_exact / rms).max())
    else:
        assert (d_exact <= tight).all(), float((d_exact / rms).max())
    if not (norm or residual):
        assert (d_40 <= 2.0 ** -10 * np.abs(ref40) + 1.5e-3 * rms).all()
        # and the warp-reduce kernel's own noise level vs both
        r = oracle.u2h(oracle.gptq_gemm_k_major(oracle.h2u(x), *km, bias=None if b is None else oracle.h2u(b))).astype(np.float64)
        assert np.abs(got - r).max() <= 1e-2 * rms                                     # the warp-reduce kernel's fp16 noise (max over up to 6e5 outputs)


@pytest.mark.parametrize("m", [1, 2, 5, 8, 16, 17, 33])
def test_mfma_gemm_small(oracle, dev, m):
    _check_mfma(oracle, dev, 2048, 264, m, seed=20 + m)


@pytest.mark.parametrize("k,n", [(1024, 16), (1152, 40), (4096, 6144), (14336, 512), (4096, 4096)])
def test_mfma_gemm_shapes(oracle, dev, k, n):
    _check_mfma(oracle, dev, k, n, 1, seed=30)
    _check_mfma(oracle, dev, k, n, 9, seed=31, bias=True)


@pytest.mark.parametrize("m,k,n", [(17, 1024, 264), (33, 2048, 40), (64, 4096, 512), (100, 1152 + 128, 1000), (257, 2048, 384),
                                   (300, 1024, 2048), (515, 2304, 200)])
def test_tiled_gemm_shapes(oracle, dev, m, k, n):
    """The M-tiled kernel (w4_gemm_tiled.hip, both M-tile heights) incl. ragged M / N tails, bias and residual
    epilogues, split-K for the short grids; and the public entry for the same shapes."""
    _check_mfma(oracle, dev, k, n, m, seed=50 + m, force_tiled=True)
    _check_mfma(oracle, dev, k, n, m, seed=51 + m, bias=True, force_tiled=True)
    _check_mfma(oracle, dev, k, n, m, seed=52 + m, residual=True, force_tiled=True)
    _check_mfma(oracle, dev, k, n, m, seed=53 + m, bias=True)
    _check_mfma(oracle, dev, k, n, m, seed=54 + m, residual=True)


@pytest.mark.parametrize("m,k,n,splitk,force", [(128, 1024, 256, 0, 0), (257, 2048, 384, 0, 0), (1000, 1152 + 128, 1000, 0, 0),
                                                (130, 4096, 512, 3, 0), (70, 1024, 520, 0, 1), (384, 128, 256, 0, 0),
                                                (256, 256, 4096, 0, 0), (300, 1024, 512, 0, 3), (515, 2304, 776, 0, 3),
                                                (257, 4096, 256, 2, 3), (1024, 384, 1024, 0, 3), (300, 1024, 200, 0, 4), (515, 1152, 776, 0, 4),
                                                (1024, 512, 6144, 0, 0)])
def test_wide_gemm_shapes(oracle, dev, m, k, n, splitk, force, monkeypatch):
    """The prompt-chunk tiles (k_w4a16_gemm_wide<8 / 16>: 128 or 256 x 256 outputs per workgroup, swizzled LDS image of the
    activation chunk, MFMA / dequant interleave in issue order): ragged M / N, one- and two-chunk K (shorter than the
    pipeline), odd chunk counts (the zero-scale tail chunk), split-K through the caller's scratch, bias / residual epilogues;
    force = ZL_W4_TILED_WIDE (1: also below 128 rows, 3: the 256-row tile, 4: 192-column tiles -- what the cost model picks
    for N = 6144, the last case); the same bars as the other M-tiled kernel."""
    if splitk:
        monkeypatch.setenv("ZL_W4_TILED_SPLITK", str(splitk))
    if force:
        monkeypatch.setenv("ZL_W4_TILED_WIDE", str(force))
    _check_mfma(oracle, dev, k, n, m, seed=70 + m, force_tiled=True)
    _check_mfma(oracle, dev, k, n, m, seed=71 + m, bias=True, force_tiled=True)
    _check_mfma(oracle, dev, k, n, m, seed=72 + m, residual=True, force_tiled=True)


@pytest.mark.parametrize("m,n,k", [(1024, 28672, 256), (768, 28672, 128), (2048, 14336, 128)])
def test_wide_gemm_two_tile_heights(oracle, dev, m, n, k, monkeypatch):
    """The two-height plan of k_w4a16_gemm_wide (256-row tiles over all column strips + 192-row tiles over the rows the tall ones
    leave on some strips, two launches that each fill the chip once -- gate|up of a 1 024-token prompt chunk): every output is
    still one workgroup's sum over K in chunk order, so the result equals the one-height tiling (ZL_W4_TILED_WIDE=6) BIT FOR BIT,
    plain / bias / residual / silu.mul epilogues; and rows on both sides of every tile seam against the oracle's product."""
    from zhilight_amd import ops
    _u16 = lambda t: _np(t).view(np.uint16)
    rng = np.random.default_rng(900 + m + k)
    g = 128
    qw, qz, sc = synth.gptq_hf(rng, k, n, g)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, g)
    w = ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g)
    x = synth.act(rng, m, k)
    b = (rng.standard_normal(n) * 0.1).astype(np.float16)
    res = synth.act(rng, m, n)
    outs = {}
    for mode in ("0", "6"):
        monkeypatch.setenv("ZL_W4_TILED_WIDE", mode)
        outs[mode] = [_u16(ops.w4a16_gemm_tiled(_t(x, dev), w)),
                      _u16(ops.w4a16_gemm_tiled(_t(x, dev), w, bias=_t(b, dev))),
                      _u16(ops.w4a16_gemm_tiled(_t(x, dev), w, residual=_t(res, dev), epilogue=ops.EPI_RESIDUAL)),
                      _u16(ops.w4a16_gemm_tiled(_t(x, dev), w, epilogue=ops.EPI_SILU_MUL))]
    for a, c in zip(outs["0"], outs["6"]):
        assert np.array_equal(a, c)
    # seams: 256-row tiles start at 0, 256, ...; the 192-row tiles at 256 kbig + 192 i
    rows = sorted({r for s0 in range(0, m, 64) for r in (s0, s0 + 63)} | {m - 1})
    w16 = oracle.gptq_dequant_k_major(*km)
    ref = oracle.gemm_nt(oracle.h2u(x[rows]), w16, None, exact=True)
    got = outs["0"][0].view(np.float16)[rows].astype(np.float64)
    rms = np.sqrt((ref ** 2).mean())
    assert (np.abs(got - ref) <= 2.0 ** -10 * np.abs(ref) + 2e-5 * rms).all()


@pytest.mark.parametrize("rounds", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("m", [5, 16, 17, 32])
def test_phase_gemm_rounds(oracle, dev, m, rounds, monkeypatch):
    """The phase-pipelined streaming kernel (w4_phase.hip), every tiles-per-workgroup instantiation x both row-block
    counts: ragged N (tile overrun of the last workgroup), K with a partial last phase, K shorter than one ring,
    bias / residual epilogues."""
    monkeypatch.setenv("ZL_W4_PHASE_ROUNDS", str(rounds))
    monkeypatch.setenv("ZL_W4_SLAB", "-1")                # 9..32 rows default to w4_slab.hip since round 6: this test is the phase kernel's
    _check_mfma(oracle, dev, 1152, 16 * (3 * rounds - 1) + 8, m, seed=70 + m + rounds)                 # 2 phases, partial
    _check_mfma(oracle, dev, 4096, 16 * 2 * rounds, m, seed=71 + m + rounds, bias=True)                # 4 phases
    _check_mfma(oracle, dev, 9 * 1024 + 256, 16 * rounds + 16, m, seed=72 + m + rounds, residual=True)  # 10 phases, beyond any BODY


@pytest.mark.parametrize("m", [13, 16, 17, 25, 32])
def test_phase_gemm_k_split(oracle, dev, m, monkeypatch):
    """17..32 rows with K > 8192: K split over adjacent workgroups, partials merged by the last arriver in split order
    (deterministic); ragged N, K with a partial last phase and an odd phase count, bias / residual; repeated launches
    leave the arrival counters clean"""
    from zhilight_amd import ops
    monkeypatch.setenv("ZL_W4_SLAB", "-1")                # the phase kernel's own K split (w4_slab.hip is the default since round 6)
    _check_mfma(oracle, dev, 9 * 1024 + 256, 16 * 9 + 8, m, seed=110 + m)
    _check_mfma(oracle, dev, 14336, 256, m, seed=111 + m, bias=True)
    _check_mfma(oracle, dev, 14336, 512, m, seed=112 + m, residual=True)
    rng = np.random.default_rng(113)
    w = ops.W4MWeight.random(1024, 14336, 128, dev)
    x = _t(synth.act(rng, m, 14336), dev)
    y0 = ops.w4a16_gemm_mfma(x, w)
    for _ in range(5):
        assert torch.equal(ops.w4a16_gemm_mfma(x, w), y0)


@pytest.mark.parametrize("k,n,norm,bias", [(8192, 2560, True, True), (2048, 8192, False, False), (8192, 1856, True, False), (7424, 1024, False, False)])
def test_qwen2_72b_tp4_rank_shapes(oracle, dev, k, n, norm, bias):
    """per-rank projections of BASELINE configs[3] (Qwen2-72B GPTQ-Int4, TP = 4: dim 8192, 16 + 2 x 2 heads of 128, dim_ff
    29696 / 4 = 7424 = 58 groups; qkv carries a bias; gate|up and o cut to 1/8 of their rows to keep the CPU oracle
    quick): decode rows 1 and 8"""
    _check_mfma(oracle, dev, k, n, 1, seed=300 + n, norm=norm, bias=bias)
    _check_mfma(oracle, dev, k, n, 8, seed=301 + n, bias=bias)
    _check_mfma(oracle, dev, k, n, 20, seed=302 + n, residual=not bias)


@pytest.mark.parametrize("m", [2, 5, 7, 8])
def test_phase_gemm_fused_norm_rows(oracle, dev, m, monkeypatch):
    """the register-resident fused RMSNorm of the phase kernel for 1..8 rows (one instantiation of the row reductions
    per row count), K with a partial last phase and K = 4096, several tiles-per-workgroup settings"""
    for rounds in (1, 2, 7):
        monkeypatch.setenv("ZL_W4_PHASE_ROUNDS", str(rounds))
        _check_mfma(oracle, dev, 1152, 16 * rounds + 8, m, seed=130 + m + rounds, norm=True)
    monkeypatch.delenv("ZL_W4_PHASE_ROUNDS")
    _check_mfma(oracle, dev, 4096, 264, m, seed=135 + m, norm=True, bias=True)


def test_streaming_gemm_random_shapes(oracle, dev, monkeypatch):
    """40 seeded random (M, N, K, epilogue, tiles-per-workgroup) draws over the streaming kernels' whole dispatch range
    (1..32 rows; fused norm where the launcher offers it; K from one partial phase to 20 phases; ragged N)"""
    rng = np.random.default_rng(2024)
    for case in range(40):
        m = int(rng.integers(1, 33))
        k = 128 * int(rng.integers(1, 161))
        n = 8 * int(rng.integers(2, 80))
        norm = bool(rng.integers(0, 2)) and m <= 8 and k <= 4096
        kind = int(rng.integers(0, 3))
        if rng.integers(0, 2):
            monkeypatch.setenv("ZL_W4_PHASE_ROUNDS", str(int(rng.integers(1, 9))))
        else:
            monkeypatch.delenv("ZL_W4_PHASE_ROUNDS", raising=False)
        _check_mfma(oracle, dev, k, n, m, seed=1000 + case, bias=kind == 1, residual=kind == 2 and not norm, norm=norm)


@pytest.mark.parametrize("geom", [None, (4, 1), (8, 2), (4, 4), (1, 8, 4), (2, 8, 4), (7, 8, 4), (8, 8, 4)])
@pytest.mark.parametrize("m", [3, 5, 8, 9, 16, 17, 32])
def test_slab_gemm_geometries(oracle, dev, m, geom, monkeypatch):
    """w4_slab.hip (5..32 rows, round 6): the planner's own geometry and forced (waves per workgroup, groups per wave) pairs -- with the
    tiles per workgroup forced as well in the 3-tuples: the two-groups-resident / DMA-two-groups-ahead schedule with 1, 2, 7, 8 weight
    items per group --, one
    and two row blocks -- ragged N (a partial last tile group and a partial tile), K with a partial last split and K shorter than
    one workgroup's slice, bias / residual epilogues, every K-split count from 1 to 18; same bars as the phase kernel."""
    if geom is not None:
        if len(geom) == 3:
            monkeypatch.setenv("ZL_W4_SLAB_R", str(geom[0]))
        monkeypatch.setenv("ZL_W4_SLAB_NW", str(geom[-2]))
        monkeypatch.setenv("ZL_W4_SLAB_GPW", str(geom[-1]))
    _check_mfma(oracle, dev, 1152, 16 * 11 + 8, m, seed=500 + m)
    _check_mfma(oracle, dev, 4096, 512, m, seed=501 + m, bias=True)
    _check_mfma(oracle, dev, 9 * 1024 + 256, 144, m, seed=502 + m, residual=True)
    _check_mfma(oracle, dev, 128, 40, m, seed=503 + m)
    _check_mfma(oracle, dev, 14336, 256, m, seed=504 + m, residual=True, bias=True)


def test_slab_gemm_llama_shapes_repeatable(oracle, dev):
    """the four projections of a Llama-3-8B layer at 32 rows through the slab kernel: two runs return the same bits (every sum
    has a fixed order), interleaved launches of different shapes share the scratch, the arrival counters end at zero."""
    from zhilight_amd import ops
    rng = np.random.default_rng(77)
    ws = {(n, k): ops.W4MWeight.random(n, k, 128, dev) for n, k in [(6144, 4096), (4096, 4096), (4096, 14336)]}
    xs = {k: _t(synth.act(rng, 32, k), dev) for k in (4096, 14336)}
    first = {key: ops.w4a16_gemm_mfma(xs[key[1]], w) for key, w in ws.items()}
    for _ in range(4):
        for key, w in ws.items():
            assert torch.equal(ops.w4a16_gemm_mfma(xs[key[1]], w), first[key])
    scratch = ops.w4_scratch(dev, 32, 6144)
    torch.cuda.synchronize()
    assert int(scratch[:65536].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("m", [9, 20, 32])
def test_slab_against_phase_kernel(dev, m, monkeypatch):
    """same operands through w4_slab.hip and through w4_phase.hip: the per-(column, group) arithmetic is identical, only the
    order of the fp32 additions over the K / 128 groups differs -- outputs agree to a few fp32 roundings of the sum, i.e. the
    fp16 results differ on rounding ties only (<= 1 ulp, rarely)."""
    from zhilight_amd import ops
    rng = np.random.default_rng(78 + m)
    w = ops.W4MWeight.random(1024, 4096, 128, dev)
    x = _t(synth.act(rng, m, 4096), dev)
    a = ops.w4a16_gemm_mfma(x, w).float()
    monkeypatch.setenv("ZL_W4_SLAB", "-1")
    b = ops.w4a16_gemm_mfma(x, w).float()
    diff = (a - b).abs()
    assert float((diff > 0).float().mean()) < 0.02
    assert bool((diff <= 2.0 ** -10 * b.abs() + 1e-6).all())


@pytest.mark.parametrize("m", [8, 9, 24, 32])
def test_phase_gemm_silu_mul(oracle, dev, m):
    from zhilight_amd import ops
    rng = np.random.default_rng(62)
    k, nff, g = 2048, 200, 128
    qw1, qz1, sc1 = synth.gptq_hf(rng, k, nff, g)
    qw2, qz2, sc2 = synth.gptq_hf(rng, k, nff, g)
    km1, km2 = oracle.gptq_prepare_k_major(qw1, qz1, sc1, g), oracle.gptq_prepare_k_major(qw2, qz2, sc2, g)
    cat = [np.concatenate([a, b], axis=0) for a, b in zip(km1, km2)]
    w = ops.W4MWeight.from_k_major(_t(cat[0].view(np.int32), dev), _t(cat[1], dev), _t(cat[2], dev, torch.float16), g,
                                   row_interleave=True)
    x = synth.act(rng, m, k)
    ge = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km1).astype(np.float16)
    ue = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km2).astype(np.float16)
    ref = oracle.u2h(oracle.silu_mul(oracle.h2u(ge), oracle.h2u(ue))).astype(np.float64)
    got = _np(ops.w4a16_gemm_mfma(_t(x, dev), w, epilogue=ops.EPI_SILU_MUL)).astype(np.float64)
    assert got.shape == (m, nff)
    # a gate / up value on a rounding tie may flip by one fp16 ulp (fp32 accumulation noise): 2^-9 of max|out|
    assert np.abs(got - ref).max() <= 2.0 ** -9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("m", [70, 300])
def test_tiled_gemm_silu_mul(oracle, dev, m):
    from zhilight_amd import ops
    rng = np.random.default_rng(61)
    k, nff, g = 1024, 200, 128
    qw1, qz1, sc1 = synth.gptq_hf(rng, k, nff, g)
    qw2, qz2, sc2 = synth.gptq_hf(rng, k, nff, g)
    km1, km2 = oracle.gptq_prepare_k_major(qw1, qz1, sc1, g), oracle.gptq_prepare_k_major(qw2, qz2, sc2, g)
    cat = [np.concatenate([a, b], axis=0) for a, b in zip(km1, km2)]
    w = ops.W4MWeight.from_k_major(_t(cat[0].view(np.int32), dev), _t(cat[1], dev), _t(cat[2], dev, torch.float16), g,
                                   row_interleave=True)
    x = synth.act(rng, m, k)
    ge = oracle.gemm_nt(oracle.h2u(x), oracle.gptq_dequant_k_major(*km1), exact=True).astype(np.float16)
    ue = oracle.gemm_nt(oracle.h2u(x), oracle.gptq_dequant_k_major(*km2), exact=True).astype(np.float16)
    ref = oracle.u2h(oracle.silu_mul(oracle.h2u(ge), oracle.h2u(ue))).astype(np.float64)
    got = _np(ops.w4a16_gemm_mfma(_t(x, dev), w, epilogue=ops.EPI_SILU_MUL)).astype(np.float64)
    assert got.shape == (m, nff)
    assert np.abs(got - ref).max() <= 2.0 ** -9 * max(1.0, np.abs(ref).max())


def test_mfma_gemm_fused_norm_residual_silu(oracle, dev):
    from zhilight_amd import ops
    _check_mfma(oracle, dev, 2048, 256, 3, seed=40, norm=True)
    _check_mfma(oracle, dev, 2048, 256, 3, seed=41, residual=True)
    rng = np.random.default_rng(42)
    k, nff, g, m = 2048, 192, 128, 4
    qw1, qz1, sc1 = synth.gptq_hf(rng, k, nff, g)
    qw2, qz2, sc2 = synth.gptq_hf(rng, k, nff, g)
    km1, km2 = oracle.gptq_prepare_k_major(qw1, qz1, sc1, g), oracle.gptq_prepare_k_major(qw2, qz2, sc2, g)
    cat = [np.concatenate([a, b], axis=0) for a, b in zip(km1, km2)]
    w = ops.W4MWeight.from_k_major(_t(cat[0].view(np.int32), dev), _t(cat[1], dev), _t(cat[2], dev, torch.float16), g,
                                   row_interleave=True)
    x = synth.act(rng, m, k)
    ge = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km1).astype(np.float16)
    ue = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km2).astype(np.float16)
    ref = oracle.u2h(oracle.silu_mul(oracle.h2u(ge), oracle.h2u(ue))).astype(np.float64)
    got = _np(ops.w4a16_gemm_mfma(_t(x, dev), w, epilogue=ops.EPI_SILU_MUL)).astype(np.float64)
    assert np.abs(got - ref).max() <= 2.0 ** -9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("algo", ["mfma", "exact"])
def test_awq_checkpoint_linear(oracle, dev, algo, monkeypatch):
    """An AWQ checkpoint through the AWQ-as-exllama load route (Int4GPTQ is_awq, linear.cpp:1139-1143):
    the linear equals x . ((q - z) * s)^T from the integers the checkpoint was packed from."""
    from zhilight_amd import ops
    from zhilight_amd.llama import Int4GPTQ, QuantConfig
    monkeypatch.setenv("ZL_W4_ALGO", algo)
    rng = np.random.default_rng(77)
    k, n, g, m = 1024, 384, 128, 3
    qw, qz, sc, q, z = synth.awq_hf(rng, k, n, g)
    quant = QuantConfig.from_hf(dict(quant_method="awq", bits=4, group_size=g, zero_point=True, version="gemm"))
    lin = Int4GPTQ("l", k, n, quant)
    lin.load_state_dict({"l.qweight": torch.from_numpy(qw.view(np.int32)), "l.qzeros": torch.from_numpy(qz.view(np.int32)),
                         "l.scales": torch.from_numpy(sc.view(np.float16))}, "l", dev)
    # the load transform equals the oracle's restatement of the reference's
    tr = lambda a: np.ascontiguousarray(a.T)   # functions::Transpose of a 2-d tensor  # noqa: E731
    km_ref = (tr(oracle.awq_shuffle(qw, True)), tr(oracle.gptq_q4_to_q8(oracle.awq_un_shuffle(qz))), tr(sc))
    assert np.array_equal(_np(lin.km[0]).view(np.uint32), km_ref[0])
    assert np.array_equal(_np(lin.km[1]), km_ref[1])
    assert np.array_equal(_np(lin.km[2]).view(np.uint16), km_ref[2])
    lin.pack()
    x = synth.act(rng, m, k)
    got = _np(lin.forward(_t(x, dev))).astype(np.float64)
    w = (q.astype(np.float64) - np.repeat(z, g, axis=0)) * np.repeat(sc.view(np.float16).astype(np.float64), g, axis=0)
    ref = x.astype(np.float64) @ w                       # (m, n)
    rms = np.sqrt((ref ** 2).mean())
    assert np.abs(got - ref).max() <= 2.0 ** -10 * np.abs(ref).max() + 4e-3 * rms
    if algo == "exact":   # and bit for bit the reference-faithful kernel arithmetic on the converted operands
        r = oracle.gptq_gemm_k_major(oracle.h2u(x), *km_ref)
        assert np.array_equal(_np(lin.forward(_t(x, dev))).view(np.uint16), r)


@pytest.mark.parametrize("algo", ["mfma", "exact"])
def test_act_order_linear(oracle, dev, algo, monkeypatch):
    """desc_act checkpoints (a6): rows regrouped at load, activation columns gathered in forward -- equals
    x . W^T for W[k, n] = (q - z[g_idx[k]]) * s[g_idx[k]]."""
    from zhilight_amd.llama import Int4GPTQ, QuantConfig
    monkeypatch.setenv("ZL_W4_ALGO", algo)
    rng = np.random.default_rng(91)
    k, n, g = 1024, 256, 128
    qw, qz, sc, g_idx, w16 = synth.gptq_act_order_hf(rng, k, n, g)
    quant = QuantConfig.from_hf(dict(quant_method="gptq", bits=4, group_size=g, desc_act=True))
    lin = Int4GPTQ("l", k, n, quant)
    lin.load_state_dict({"l.qweight": torch.from_numpy(qw.view(np.int32)), "l.qzeros": torch.from_numpy(qz.view(np.int32)),
                         "l.scales": torch.from_numpy(sc.view(np.float16)), "l.g_idx": torch.from_numpy(g_idx)}, "l", dev)
    assert lin.perm is not None
    lin.pack()
    for m in (1, 5, 70):
        x = synth.act(rng, m, k)
        got = _np(lin.forward(_t(x, dev))).astype(np.float64)
        ref = x.astype(np.float64) @ w16.astype(np.float64).T
        rms = np.sqrt((ref ** 2).mean())
        assert np.abs(got - ref).max() <= 2.0 ** -10 * np.abs(ref).max() + 6e-3 * rms, (m, np.abs(got - ref).max() / rms)
    # a g_idx that is not a regrouping into groups of g is rejected
    bad = g_idx.copy()
    bad[0] = bad[1]
    with pytest.raises(Exception):
        Int4GPTQ("l", k, n, quant).load_state_dict(
            {"l.qweight": torch.from_numpy(qw.view(np.int32)), "l.qzeros": torch.from_numpy(qz.view(np.int32)),
             "l.scales": torch.from_numpy(sc.view(np.float16)), "l.g_idx": torch.from_numpy(bad)}, "l", dev)


@pytest.mark.parametrize("m,norm", [(1, True), (3, True), (4, False), (6, True), (8, True), (8, False), (9, False), (16, False), (17, False), (32, False)])
@pytest.mark.parametrize("bshd", [True, False])
def test_fused_qkv_rotary_scatter_equals_two_call_sequence(oracle, dev, m, norm, bshd):
    """zl_w4a16_qkv_rope_scatter == zl_w4a16_gemm_mfma + zl_rope_scatter_decode, bit for bit: rotated q, and the K / V
    buffers (incl. a task whose placement is -1 and the last slot of a buffer); bias on."""
    from zhilight_amd import ops
    _fused_qkv_rope_case(oracle, dev, m, norm, bshd, 8, 2, 128, 1024 + 256)


@pytest.mark.parametrize("m", [12, 32])
@pytest.mark.parametrize("h,hkv,d,k", [(6, 3, 64, 4096), (5, 1, 32, 1152), (4, 4, 128, 8192)])
def test_fused_qkv_rotary_scatter_slab_head_sizes(oracle, dev, m, h, hkv, d, k):
    """the slab kernel's rotation epilogue (9..32 rows): head sizes 32 / 64 / 128 (a workgroup's eight tiles hold 4 / 2 / 1 heads),
    a head count that leaves the last tile group partial, K = 8192 (K split 4)"""
    _fused_qkv_rope_case(oracle, dev, m, False, True, h, hkv, d, k)


def _fused_qkv_rope_case(oracle, dev, m, norm, bshd, h, hkv, d, k):
    from zhilight_amd import ops
    rng = np.random.default_rng(80 + m)
    g = 128
    n = (h + 2 * hkv) * d
    qw, qz, sc = synth.gptq_hf(rng, k, n, g)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, g)
    w = ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g)
    x = _t(synth.act(rng, m, k, 2.0 if norm else 1.0), dev)
    bias = _t((rng.standard_normal(n) * 0.1).astype(np.float16), dev)
    nw = _t((1.0 + 0.1 * rng.standard_normal(k)).astype(np.float16), dev) if norm else None
    lens = [int(v) for v in rng.integers(2, 6, m) * 32]
    pos = np.array([int(rng.integers(0, L)) for L in lens], np.int32)
    pos[0] = lens[0] - 1
    placement = pos.copy()
    if m > 1:
        placement[1] = -1
    cs, sn = oracle.rope_cos_sin(pos, d, 5e5, True, (8.0, 1.0, 4.0, 8192.0))
    shape = (lambda L: (L, hkv, d)) if bshd else (lambda L: (hkv, L, d))
    mk = lambda: [torch.full(shape(L), 3.0, dtype=torch.float16, device=dev) for L in lens]
    k1, v1, k2, v2 = mk(), mk(), mk(), mk()
    lens_t, place_t = _t(np.array(lens, np.int32), dev), _t(placement, dev)
    # two calls
    qkv = ops.w4a16_gemm_mfma(x, w, bias=bias, norm_weight=nw, norm_eps=1e-5)
    q_ref = ops.rope_scatter_decode(_t(cs, dev), _t(sn, dev), qkv, place_t, lens_t, ops.make_ptr_table(k1), ops.make_ptr_table(v1),
                                    h, hkv, d, True, bshd)
    # fused
    assert ops.w4_qkv_rope_scatter_ok(m, k, d, norm)
    q_got = ops.w4_qkv_rope_scatter(x, w, _t(cs, dev), _t(sn, dev), place_t, lens_t, ops.make_ptr_table(k2), ops.make_ptr_table(v2),
                                    h, hkv, d, bias=bias, norm_weight=nw, norm_eps=1e-5, bshd=bshd)
    assert torch.equal(q_got, q_ref)
    for a, b_ in zip(k1 + v1, k2 + v2):
        assert torch.equal(a, b_)
    assert not torch.equal(k1[0], torch.full_like(k1[0], 3.0))


@pytest.mark.parametrize("b,h,hkv,n,lens,valid", [
    (1, 32, 8, 4096, [1088], [1025]),                       # the batch-1 decode step: 9 splits, the last holds one key
    (1, 32, 8, 4096, [1088], [1]),                          # a single key
    (1, 16, 4, 2048 + 16, [2048], [2048]),                  # K = 2048: half the threads hold no activation; 16 splits
    (3, 32, 8, 4096, [640, 128, 1088], [517, 128, 1000]),   # ragged tasks, different split counts per row
    (4, 32, 4, 8192, [256, 192, 64, 256], [129, 192, 33, 256]),   # two row tiles per workgroup
])
@pytest.mark.parametrize("algo", ["i8p_half", "phase_f32"])
def test_attention_split_merge_in_attn_out_projection(oracle, dev, b, h, hkv, n, lens, valid, algo, monkeypatch):
    """The split merge in the attn_out projection's prologue against zl_decode_attn + zl_w4a16_gemm_mfma, plain and residual
    epilogues.  phase_f32 (ZL_W4_SMALL_ALGO=1: zl_decode_attn_splits + zl_w4a16_gemm_attn_merge, fp32 partials, the round-2
    kernels): bit for bit.  i8p_half (the default: zl_decode_attn_splits_h + zl_w4a16_gemm_attn_merge_h): the partial rows
    pass through fp16 once more, so the merged activations may differ from the merge launch's by one fp16 ulp of the largest
    partial -- the projection outputs then agree within 2^-9 of the output scale (random signs over K = 4096 terms), and the
    merged rows themselves are checked against the merge launch through an identity-like probe below."""
    from zhilight_amd import ops
    if algo == "phase_f32":
        monkeypatch.setenv("ZL_W4_SMALL_ALGO", "1")
    rng = np.random.default_rng(600 + b + n)
    d, g, k = 128, 128, h * 128
    n8 = (n + 7) // 8 * 8
    qw, qz, sc = synth.gptq_hf(rng, k, n8, g)
    km = tuple(np.ascontiguousarray(a[:n]) for a in oracle.gptq_prepare_k_major(qw, qz, sc, g))
    w = ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g)
    max_len = max(lens)
    dk = [torch.randn(L, hkv, d, device=dev).half() for L in lens]
    dv = [torch.randn(L, hkv, d, device=dev).half() for L in lens]
    q = torch.randn(b, 1, h, d, device=dev).half()
    bl, vl = _t(np.array(lens, np.int32), dev), _t(np.array(valid, np.int32), dev)
    ka, va = ops.make_ptr_table(dk), ops.make_ptr_table(dv)
    scale = 1.0 / np.sqrt(d)
    import os
    os.environ["ZL_ATTN_MERGE_MAX_B"] = "4"
    try:
        plan = ops.attn_merge_plan(b, h, hkv, d, max_len, w)
    finally:
        del os.environ["ZL_ATTN_MERGE_MAX_B"]
    assert plan is not None
    ws = ops.decode_attn_workspace(b, 1, h, d, max_len, dev)
    att = ops.multi_query_attention_rag_buffer(q, bl, ka, va, None, scale, max_len, hkv, valid_lens=vl, workspace=ws)
    res = torch.randn(b, n, device=dev).half()
    want = ops.w4a16_gemm_mfma(att.view(b, k), w)
    want_res = ops.w4a16_gemm_mfma(att.view(b, k), w, residual=res, epilogue=ops.EPI_RESIDUAL)
    ws.fill_(float("nan"))                                   # nothing stale may be read
    ops.decode_attention_splits(q, bl, ka, va, vl, scale, max_len, hkv, ws)
    got = ops.w4_attn_out_merge(ws, bl, vl, plan, b, w)
    got_res = ops.w4_attn_out_merge(ws, bl, vl, plan, b, w, residual=res, epilogue=ops.EPI_RESIDUAL)
    assert torch.isfinite(got.float()).all()
    if algo == "phase_f32":
        assert torch.equal(got, want) and torch.equal(got_res, want_res)
    else:
        tol = 2.0 ** -9 * want.float().abs().max().item()
        assert (got.float() - want.float()).abs().max().item() <= tol
        assert (got_res.float() - want_res.float()).abs().max().item() <= tol + 2.0 ** -10 * want_res.float().abs().max().item()
        # the fraction of outputs that moved at all stays small-ish and unbiased: mean signed difference ~ 0
        diff = got.float() - want.float()
        assert abs(diff.mean().item()) <= 0.05 * tol


@pytest.mark.parametrize("b,h,hkv,n,lens", [
    (1, 32, 8, 4096, [1088]),
    (1, 16, 4, 2048 + 16, [2048]),
    (3, 32, 8, 4096, [640, 128, 1088]),
    (4, 32, 4, 8192, [256, 192, 33, 256]),
])
@pytest.mark.parametrize("mode", ["holes", "prefix"])
def test_attention_mask_form_split_records_merge(oracle, dev, b, h, hkv, n, lens, mode):
    """The reference's int8 visibility mask on the split-record route the host library defers (hostcpp/nn_amd.cpp, DeferredOp kind 3):
    zl_decode_attn_splits_h_mask leaves half records of every split of the buffer; their stand-alone merge (zl_decode_attn_combine_h)
    against the fp64 oracle (the records pass through fp16 once: 2^-10 of the largest row on top of the 1e-3 bar), and the merging
    projection (zl_w4a16_gemm_attn_merge_h with valid_lens = buf_lens) bit for bit against the plain projection of those merged rows --
    whichever the next call turns out to be, the layer sees the same bits.  NaN in K / V of every invisible key must not leak."""
    from zhilight_amd import ops
    rng = np.random.default_rng(700 + b + n)
    d, g, k = 128, 128, h * 128
    n8 = (n + 7) // 8 * 8
    qw, qz, sc = synth.gptq_hf(rng, k, n8, g)
    km = tuple(np.ascontiguousarray(a[:n]) for a in oracle.gptq_prepare_k_major(qw, qz, sc, g))
    w = ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g)
    max_len = max(lens)
    kh = [rng.standard_normal((L, hkv, d)).astype(np.float16) for L in lens]
    vh = [rng.standard_normal((L, hkv, d)).astype(np.float16) for L in lens]
    masks = []
    for i, L in enumerate(lens):
        if mode == "prefix":
            m = (np.arange(L) < max(1, L - 63)).astype(np.int8)
        else:
            m = (rng.random(L) < 0.6).astype(np.int8)
            m[: min(L // 2, 300)] = 0                          # whole splits without a visible key
            m[L - 1] = 1
        masks.append(m)
    mask = np.concatenate(masks)
    dk, dv = [], []
    for i, L in enumerate(lens):                               # the device copies carry NaN where nothing may be read
        pk, pv = kh[i].copy(), vh[i].copy()
        pk[masks[i] == 0] = np.nan
        pv[masks[i] == 0] = np.nan
        dk.append(_t(pk, dev))
        dv.append(_t(pv, dev))
    qh = rng.standard_normal((b, 1, h, d)).astype(np.float16)
    q = _t(qh, dev)
    bl = _t(np.array(lens, np.int32), dev)
    ka, va = ops.make_ptr_table(dk), ops.make_ptr_table(dv)
    scale = 1.0 / np.sqrt(d)
    exact = oracle.mqa_rag_buffer(qh.view(np.uint16), np.array(lens, np.int32), [a.view(np.uint16) for a in kh], [a.view(np.uint16) for a in vh],
                                  mask, hkv, scale, True, exact=True)
    import os
    os.environ["ZL_ATTN_MERGE_MAX_B"] = "4"
    try:
        plan = ops.attn_merge_plan(b, h, hkv, d, max_len, w)
    finally:
        del os.environ["ZL_ATTN_MERGE_MAX_B"]
    assert plan is not None and plan[2]
    ws = ops.decode_attn_workspace(b, 1, h, d, max_len, dev)
    ws.fill_(float("nan"))
    ops.decode_attention_splits_mask(q, bl, ka, va, _t(mask, dev), scale, max_len, hkv, ws)
    att = ops.decode_attention_combine_h(ws, bl, None, b, h, hkv, max_len)
    ga = att.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(ga).all()
    top = max(1.0, np.abs(exact).max())
    assert np.abs(ga - exact.reshape(ga.shape)).max() < (1e-3 + 2.0 ** -10) * top, np.abs(ga - exact.reshape(ga.shape)).max()
    res = torch.randn(b, n, device=dev).half()
    want = ops.w4a16_gemm_mfma(att.view(b, k), w)
    want_res = ops.w4a16_gemm_mfma(att.view(b, k), w, residual=res, epilogue=ops.EPI_RESIDUAL)
    got = ops.w4_attn_out_merge(ws, bl, bl, plan, b, w)
    got_res = ops.w4_attn_out_merge(ws, bl, bl, plan, b, w, residual=res, epilogue=ops.EPI_RESIDUAL)
    assert torch.equal(got, want) and torch.equal(got_res, want_res)


def test_attention_split_merge_plan_limits(dev):
    from zhilight_amd import ops
    from zhilight_amd._lib import ZLError
    w = ops.W4MWeight(4096, 4096, 128, None, None)
    assert ops.attn_merge_plan(1, 32, 8, 128, 1088, w) == (128, 9, True)
    assert ops.attn_merge_plan(1, 32, 8, 128, 1088, w, torch.bfloat16) == (128, 9, False)   # one layout decision for both launches
    assert ops.attn_merge_plan(2, 32, 8, 128, 1088, w) is None          # ZL_ATTN_MERGE_MAX_B defaults to 1
    assert ops.attn_merge_plan(1, 32, 8, 128, 4096, w) is None          # 32 splits
    assert ops.attn_merge_plan(1, 32, 8, 64, 1088, w) is None
    ws = torch.zeros(1 << 16, dtype=torch.float32, device=dev)
    i32 = torch.ones(8, dtype=torch.int32, device=dev)
    wq = torch.zeros(8 * 1024 * 1024 // 4, dtype=torch.int32, device=dev)
    real = ops.W4MWeight(4096, 4096, 128, wq, wq)
    with pytest.raises(ZLError):
        ops.w4_attn_out_merge(ws, i32, i32, (128, 9), 5, real)          # more than 4 rows
    with pytest.raises(ZLError):
        ops.w4_attn_out_merge(ws, i32, i32, (128, 17), 1, real)         # more than 16 splits


@pytest.mark.parametrize("k,n,g,inter", [(1024, 256, 128, False), (4096, 1000, 128, False), (2048, 528, 256, True), (1152, 40, 128, True)])
def test_w4m_pack_unpack_round_trip(oracle, dev, k, n, g, inter):
    """zl_w4m_unpack is the exact inverse of zl_w4m_pack (ragged N, group = 2 tiles, row-interleaved gate|up pairs), and a
    weight generated directly in the packed layout (W4MWeight.random: what bench.py times) unpacks to operands whose exact
    product the kernel reproduces -- the check bench.py runs before its timed region."""
    from zhilight_amd import ops
    rng = np.random.default_rng(k + n)
    qw, qz, sc = synth.gptq_hf(rng, k, (n + 7) // 8 * 8, g)
    km = tuple(np.ascontiguousarray(a[:n]) for a in oracle.gptq_prepare_k_major(qw, qz, sc, g))
    w = ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g, row_interleave=inter)
    uq, uz, us = w.to_k_major()
    assert np.array_equal(_np(uq).view(np.uint32), km[0])
    assert np.array_equal(_np(uz), km[1] & 0xF)
    assert np.array_equal(_np(us).view(np.uint16), km[2])
    wr = ops.W4MWeight.random(n // 16 * 16 or 16, k, g, dev, row_interleave=inter)
    rq, rz, rs = wr.to_k_major()
    x = synth.act(rng, 3, k)
    y = _np(ops.w4a16_gemm_mfma(_t(x, dev), wr)).astype(np.float64)
    if inter:
        y = np.concatenate([y[:, 0::2], y[:, 1::2]], axis=1)
    exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), _np(rq).view(np.uint32), _np(rz), _np(rs).view(np.uint16))
    assert np.abs(y - exact).max() <= 2.0 ** -10 * np.abs(exact).max() + 1e-6


@pytest.mark.parametrize("n,k,m", [(8192, 4096, 4), (12288, 4096, 4), (4096, 4096, 3), (4096, 14336, 2)])
def test_i8p_rows_repeated_against_fp16_kernels(dev, n, k, m):
    """The integer-plane kernel (default for 1..4 rows) against the round-2 fp16-dequant kernels on fresh random activations, 12
    times per shape.  Regression for a source-operand hazard: with the A / B registers of v_mfma_i32_16x16x64_i8 reused one
    issue slot after the MFMA, the rows of its last pass -- the FOURTH batch row -- came out wrong in one tile every other
    launch (N = 8192: two tiles per workgroup), everything else exact."""
    from zhilight_amd import ops
    import os
    torch.manual_seed(n + k + m)
    w = ops.W4MWeight.random(n, k, 128, dev)
    for it in range(12):
        x = torch.randn(m, k, dtype=torch.float16, device=dev)
        os.environ["ZL_W4_SMALL_ALGO"] = "1"
        try:
            want = ops.w4a16_gemm_mfma(x, w)
        finally:
            del os.environ["ZL_W4_SMALL_ALGO"]
        got = ops.w4a16_gemm_mfma(x, w)
        tol = 2.0 ** -10 * want.float().abs().max().item()      # both within fp16 output rounding of the exact product
        assert (got.float() - want.float()).abs().max().item() <= tol, it


def _edge_weight(oracle, dev, rng, k, n):
    from zhilight_amd import ops
    qw, qz, sc = synth.gptq_hf(rng, k, n, 128)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, 128)
    return km, ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), 128)


@pytest.mark.parametrize("m,k,n", [(1, 4096, 512), (4, 4096, 256), (2, 14336, 256)])
@pytest.mark.parametrize("case", ["outlier100", "outlier1000", "subnormal_group", "max_half", "mixed"])
def test_i8p_activation_edge_cases(oracle, dev, m, k, n, case):
    """the integer-plane kernel's block-floating image of the activations (one power-of-two scale per
    128-k group, three signed byte digits) on inputs an N(0, 1) draw never produces -- outlier channels 100x / 1000x the group's
    typical value (the small values then sit 2^-7 .. 2^-10 below the group maximum), a group of fp16 SUBNORMALS, +-65504, and all
    of them together.  Bar: the exact fp64 product of the fp16 activations with (q - z) s, to fp16 output rounding + 2e-5 rms
    -- the same bar as for benign inputs (the image is exact for every value within 2^-12 of its group's maximum and carries
    2^-22 of that maximum otherwise)."""
    from zhilight_amd import ops
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{m}-{k}-{n}-{case}".encode()))
    km, w = _edge_weight(oracle, dev, rng, k, n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    if case in ("outlier100", "outlier1000", "mixed"):
        cols = rng.choice(k, size=k // 64, replace=False)
        x[:, cols] *= 100.0 if case == "outlier100" else 1000.0
    if case in ("subnormal_group", "mixed"):
        g0 = 128 * int(rng.integers(0, k // 128))
        x[:, g0:g0 + 128] = rng.uniform(-6e-5, 6e-5, (m, 128))          # fp16 subnormals are < 6.1e-5
        x[0, g0 + 5] = 5.96e-8                                           # the smallest one
    if case in ("max_half", "mixed"):
        cols = rng.choice(k, size=8, replace=False)
        x[:, cols] = 65504.0 * np.sign(rng.standard_normal((m, 8)))
    xh = np.clip(x, -65504, 65504).astype(np.float16)
    got = _np(ops.w4a16_gemm_mfma(_t(xh, dev), w)).astype(np.float64)
    exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(xh), *km)
    assert np.isfinite(exact).all()
    rms = np.sqrt((exact ** 2).mean())
    fin = np.abs(exact) < 65504.0 * (1 - 2.0 ** -11)                     # (outputs past the fp16 range round to inf on both sides)
    d = np.abs(got - exact)
    assert (d[fin] <= 2.0 ** -10 * np.abs(exact[fin]) + 2e-5 * rms).all(), float((d[fin] / rms).max())
    assert np.isinf(got[~fin]).all()


@pytest.mark.parametrize("m,k", [(1, 4096), (3, 4096), (2, 14336)])
def test_i8p_nonfinite_activations_propagate(oracle, dev, m, k):
    """an infinity or a NaN among the activations: every output of the row is non-finite, as with the bit-exact kernel
    (zl_w4a16_gemm, the reference's arithmetic: inf x (q - z) s summed with mixed signs) -- and rows without one are untouched.
    Before round 4 the group exponent was clamped and such a row came out as finite garbage."""
    from zhilight_amd import ops
    rng = np.random.default_rng(k + m)
    n = 256
    km, w = _edge_weight(oracle, dev, rng, k, n)
    w_exact = ops.W4Weight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), 128)
    x = rng.standard_normal((m, k)).astype(np.float16)
    clean = _np(ops.w4a16_gemm_mfma(_t(x, dev), w))
    for bad in (np.inf, -np.inf, np.nan):
        xb = x.copy()
        xb[0, int(rng.integers(0, k))] = bad
        got = _np(ops.w4a16_gemm_mfma(_t(xb, dev), w))
        ref = _np(ops.w4a16_gemm(_t(xb, dev), w_exact))
        assert not np.isfinite(ref[0]).any()                             # the reference's arithmetic: nothing finite survives
        assert not np.isfinite(got[0]).any()
        assert np.array_equal(got[1:], clean[1:])                        # the other rows do not see it


# ---------------------------------------------------------------------------------------------------------------------
# round 5: the RMSNorm of 9..32 decode rows DEFERRED into the phase kernel (w4_phase.hip DN) -- no stand-alone norm launch
# ---------------------------------------------------------------------------------------------------------------------
def _dn_bound(exact, lin=0.0):
    """What separates the deferred norm from norm-then-GEMM is WHERE the activation is rounded to fp16: T(x w) rs against
    T(x rs w) -- a relative rounding error of rms ~0.6 x 2^-11 per activation either way, independent between the two -> ~4-5e-4
    rms(y) per output between them (random signs over K; measured 4.9e-4 on the 6144 x 4096 matrix), 4.5 sigma over 2e5 outputs
    = 2.2e-3 measured; the bound leaves 1.6 x that.  Plus the fp16 rounding of the output itself.  For scale: E's own distance to
    the unrounded-activation product is the same 3-4e-4 rms, the reference's fp16 partial sums (R) sit 1e-2 rms from E."""
    rms = np.sqrt((exact ** 2).mean())
    return 2.0 ** -10 * (np.abs(exact) + lin) + 3.5e-3 * rms


@pytest.mark.parametrize("m", [9, 16, 17, 32])
@pytest.mark.parametrize("k,n,epi", [(4096, 6144, "plain"), (4096, 2 * 1024, "silu"), (1024 + 256, 264, "bias"), (4096, 28672, "silu"),
                                     (2048, 4096, "residual")])
def test_deferred_norm_rows_9_32(oracle, dev, m, k, n, epi, monkeypatch):
    """zl_w4a16_gemm_mfma with a norm weight and 9..32 rows (every tile count per workgroup the Llama shapes produce, one and two
    row blocks, the gated / bias / residual epilogues) against the exact product of the oracle's normalised rows.  On request only
    (zl_w4_opts_t::defer_norm): without it the boundary refuses the shape."""
    from zhilight_amd import ops
    xr, wr = torch.zeros((m, 1024), dtype=torch.float16, device=dev), ops.W4MWeight.random(64, 1024, 128, dev)
    with pytest.raises(ops.ZLError):
        ops.w4a16_gemm_mfma(xr, wr, norm_weight=torch.ones(1024, dtype=torch.float16, device=dev))
    monkeypatch.setenv("ZL_DEFER_NORM", "1")
    rng = np.random.default_rng(900 + m + n % 97)
    g = 128
    silu = epi == "silu"
    kms = []
    for _ in range(2 if silu else 1):
        qw, qz, sc = synth.gptq_hf(rng, k, n // 2 if silu else n, g)
        kms.append(oracle.gptq_prepare_k_major(qw, qz, sc, g))
    km = tuple(np.concatenate([kk[i] for kk in kms], axis=0) for i in range(3))
    w = ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g, row_interleave=silu)
    x = synth.act(rng, m, k, 3.0)
    x[m // 2] *= np.float16(0.05)                           # rows of very different norms: rs is per row
    nw = (1.0 + 0.2 * rng.standard_normal(k)).astype(np.float16)
    b = (rng.standard_normal(n) * 0.1).astype(np.float16) if epi == "bias" else None
    res = synth.act(rng, m, n) if epi == "residual" else None
    xin = oracle.rmsnorm(oracle.h2u(x), oracle.h2u(nw), 1e-5)
    y = ops.w4a16_gemm_mfma(_t(x, dev), w, bias=None if b is None else _t(b, dev), residual=None if res is None else _t(res, dev),
                            norm_weight=_t(nw, dev), norm_eps=1e-5,
                            epilogue=ops.EPI_SILU_MUL if silu else ops.EPI_RESIDUAL if res is not None else 0)
    got = _np(y).astype(np.float64)
    if silu:
        gate = oracle.gptq_gemm_k_major_exact(xin, *kms[0]).astype(np.float16)
        up = oracle.gptq_gemm_k_major_exact(xin, *kms[1]).astype(np.float16)
        ref = oracle.u2h(oracle.silu_mul(oracle.h2u(gate), oracle.h2u(up))).astype(np.float64)
        gu = np.abs(gate.astype(np.float64)) * np.abs(up.astype(np.float64))
        rms = np.sqrt((ref ** 2).mean())
        # gate and up each carry the activation-rounding noise (1.5e-3 of THEIR rms); silu' <= 1.1
        rg, ru = np.sqrt((gate.astype(np.float64) ** 2).mean()), np.sqrt((up.astype(np.float64) ** 2).mean())
        tol = 2.0 ** -9 * np.abs(ref) + 2.0 ** -10 * gu + 3.5e-3 * (1.1 * rg * np.abs(up.astype(np.float64)) + ru * np.abs(gate.astype(np.float64))) + 3e-4 * rms
        bad = np.abs(got - ref) > tol
        assert not bad.any(), (int(bad.sum()), float((np.abs(got - ref) / rms).max()))
        return
    exact = oracle.gptq_gemm_k_major_exact(xin, *km, bias=None if b is None else oracle.h2u(b))
    lin = 0.0
    if res is not None:
        lin = np.abs(exact)
        exact = res.astype(np.float64) + exact.astype(np.float16).astype(np.float64)
    d = np.abs(got - exact)
    assert (d <= _dn_bound(exact, lin)).all(), float((d / np.sqrt((exact ** 2).mean())).max())
    # and no further from the unfused sequence (stand-alone norm launch, then the same kernel) than that
    y2 = ops.w4a16_gemm_mfma(ops.rmsnorm(_t(x, dev), _t(nw, dev), 1e-5), w, bias=None if b is None else _t(b, dev),
                             residual=None if res is None else _t(res, dev), epilogue=ops.EPI_RESIDUAL if res is not None else 0)
    d2 = np.abs(got - _np(y2).astype(np.float64))
    assert (d2 <= _dn_bound(exact, lin)).all()


@pytest.mark.parametrize("m", [9, 16, 17, 32])
@pytest.mark.parametrize("bshd", [True, False])
def test_deferred_norm_fused_qkv_rotary_scatter(oracle, dev, m, bshd, monkeypatch):
    """zl_w4a16_qkv_rope_scatter with a norm weight and 9..32 rows against the unfused sequence (norm launch, projection,
    rope + scatter launch): q and the new K / V rows within the activation-rounding bound, slots and untouched rows identical."""
    from zhilight_amd import ops
    monkeypatch.setenv("ZL_DEFER_NORM", "1")
    rng = np.random.default_rng(180 + m)
    h, hkv, d, k, g = 8, 2, 128, 4096, 128
    n = (h + 2 * hkv) * d
    qw, qz, sc = synth.gptq_hf(rng, k, n, g)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, g)
    w = ops.W4MWeight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), g)
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
    qkv = ops.w4a16_gemm_mfma(ops.rmsnorm(x, nw, 1e-5), w, bias=bias)
    q_ref = ops.rope_scatter_decode(_t(cs, dev), _t(sn, dev), qkv, place_t, lens_t, ops.make_ptr_table(k1), ops.make_ptr_table(v1),
                                    h, hkv, d, True, bshd)
    assert ops.w4_qkv_rope_scatter_ok(m, k, d, True)
    q_got = ops.w4_qkv_rope_scatter(x, w, _t(cs, dev), _t(sn, dev), place_t, lens_t, ops.make_ptr_table(k2), ops.make_ptr_table(v2),
                                    h, hkv, d, bias=bias, norm_weight=nw, norm_eps=1e-5, bshd=bshd)
    ref = q_ref.float().cpu().numpy().astype(np.float64)
    rms = np.sqrt((ref ** 2).mean())
    assert (np.abs(q_got.float().cpu().numpy() - ref) <= 2.0 ** -9 * np.abs(ref) + 4e-3 * rms).all()
    for a, b_ in zip(k1 + v1, k2 + v2):
        a64, b64 = a.float().cpu().numpy().astype(np.float64), b_.float().cpu().numpy().astype(np.float64)
        assert (np.abs(a64 - b64) <= 2.0 ** -9 * np.abs(a64) + 4e-3 * rms).all()
        assert np.array_equal(a64 == 3.0, b64 == 3.0) or np.abs(a64 - b64).max() <= 4e-3 * rms     # untouched slots stay untouched
    assert not torch.equal(k2[0], torch.full_like(k2[0], 3.0))


# ---- round 6: the rows' statistics travel with the residual stream (zl_w4_opts_t::row_ss / row_ss_out) -----------------------------
def _tile_ss(x16):
    """numpy statement of a tile's sum of squares (zl_sum16): pairwise tree over the tile's 16 columns, fp32"""
    v = x16.astype(np.float32).reshape(x16.shape[0], -1, 16)
    v = v * v                                          # exact: 11-bit significands
    while v.shape[-1] > 1:
        v = v[..., 0::2] + v[..., 1::2]
    return v[..., 0]


def _rs_from_ss(ss, k, eps):
    """... and of the normalising launch's rs: lane q sums the tile sums 4 q .. 4 q + 3 (+ 64 u, u ascending) as (p0 + p1) + (p2 + p3),
    then the same 16-lane tree; rs = 1 / sqrt(sum / K + eps) in fp32"""
    m, parts = ss.shape
    p = ss.reshape(m, parts // 64, 16, 4)
    t = np.zeros((m, 16), np.float32)
    for u in range(parts // 64):
        t = t + ((p[:, u, :, 0] + p[:, u, :, 1]) + (p[:, u, :, 2] + p[:, u, :, 3]))
    while t.shape[-1] > 1:
        t = t[..., 0::2] + t[..., 1::2]
    val = t[:, 0] / np.float32(k) + np.float32(eps)
    return (np.float32(1.0) / np.sqrt(val, dtype=np.float32)).astype(np.float32)


@pytest.mark.parametrize("m,k", [(1, 16), (7, 4096), (32, 8192), (5, 1040)])
def test_row_ss_bit_exact(dev, m, k):
    from zhilight_amd import ops
    rng = np.random.default_rng(900 + m)
    x = synth.act(rng, m, k, 3.0)
    got = _np(ops.row_ss(_t(x, dev)))
    assert np.array_equal(got, _tile_ss(x))


@pytest.mark.parametrize("m", [5, 8, 9, 16, 17, 32])
@pytest.mark.parametrize("n,k,epi", [(4096, 4096, "residual"), (4096, 14336, "residual"), (1024 + 16, 4096, "bias"), (512, 2048, "addc")])
def test_slab_leaves_row_statistics(dev, m, n, k, epi):
    """a producing launch's row_ss_out == zl_row_ss of the rows it stored, bit for bit (one K slice, K split 4 folded by the last
    arriver, a ragged tile count, ADD_C), and the route question agrees with what the launch did"""
    from zhilight_amd import ops
    rng = np.random.default_rng(910 + m)
    w = ops.W4MWeight.random(n, k, 128, dev)
    x = _t(synth.act(rng, m, k), dev)
    res = _t(synth.act(rng, m, n, 2.0), dev)
    ss = torch.full((m, n // 16), -1.0, dtype=torch.float32, device=dev)
    kw = {}
    if epi == "residual":
        kw = dict(residual=res, epilogue=ops.EPI_RESIDUAL)
    elif epi == "bias":
        kw = dict(bias=_t((rng.standard_normal(n) * 0.1).astype(np.float16), dev))
    out = res.clone() if epi == "addc" else None
    if epi == "addc":
        kw = dict(epilogue=ops.EPI_ADD_C)
    assert ops.w4_row_ss_routes(m, [w], [], dev)
    y = ops.w4a16_gemm_mfma(x, w, out=out, row_ss_out=ss, **kw)
    assert torch.equal(ss, ops.row_ss(y))
    plain = ops.w4a16_gemm_mfma(x, w, out=res.clone() if epi == "addc" else None, **kw)
    assert torch.equal(plain, y)                       # and the rows themselves are what the launch stores without it


@pytest.mark.parametrize("m", [5, 8, 9, 16, 17, 32])
@pytest.mark.parametrize("n,k,epi", [(6144, 4096, 0), (2048, 4096, "silu"), (512 + 16, 8192, "bias"), (4096, 3072, "residual"),
                                     (28672, 4096, "silu"), (16 * 8 * 256, 4096, 0)])       # 7 and 8 tiles per workgroup
def test_slab_norm_from_row_statistics(oracle, dev, m, n, k, epi, monkeypatch):
    """norm_weight + row_ss on the slab kernel's NORM instantiations: the launch returns the BITS of the same kernel fed with
    T(x rs w) formed on the host from the statistics' definition (rs in the statistics' summation order, zl_rmsnorm's two products and one
    rounding per element), and sits inside the fused-norm bar of the exact product"""
    from zhilight_amd import ops
    rng = np.random.default_rng(920 + m)
    inter = epi == "silu"
    w = ops.W4MWeight.random(n, k, 128, dev, row_interleave=inter)
    x = synth.act(rng, m, k, 2.0)
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    ss = ops.row_ss(_t(x, dev))
    rs = _rs_from_ss(_np(ss), k, 1e-5)
    xn = ((x.astype(np.float32) * rs[:, None]) * nw.astype(np.float32)[None, :]).astype(np.float16)
    # the stand-alone norm agrees with that to a rounding of rs: a few elements of a row by one ulp
    xn_ref = oracle.u2h(oracle.rmsnorm(oracle.h2u(x), oracle.h2u(nw), 1e-5))
    d = np.abs(xn.astype(np.float64) - xn_ref.astype(np.float64))
    assert (d <= np.abs(xn_ref.astype(np.float64)) * 2.0 ** -10).all() and (d > 0).mean() < 0.02
    kw = {}
    if epi == "silu":
        kw = dict(epilogue=ops.EPI_SILU_MUL)
    elif epi == "bias":
        kw = dict(bias=_t((rng.standard_normal(n) * 0.1).astype(np.float16), dev))
    elif epi == "residual":
        kw = dict(residual=_t(synth.act(rng, m, n), dev), epilogue=ops.EPI_RESIDUAL)
    assert ops.w4_row_ss_routes(m, [], [(w, False)], dev)
    got = ops.w4a16_gemm_mfma(_t(x, dev), w, norm_weight=_t(nw, dev), norm_eps=1e-5, row_ss=ss, **kw)
    monkeypatch.setenv("ZL_W4_SLAB", "2")              # the same geometry for the plain launch (gate|up-sized grids at <= 16 rows)
    want = ops.w4a16_gemm_mfma(_t(xn, dev), w, **kw)
    assert torch.equal(got, want)


@pytest.mark.parametrize("m", [5, 8, 12, 32])
@pytest.mark.parametrize("h,hkv,d,k", [(32, 8, 128, 4096), (6, 3, 64, 4096), (4, 4, 128, 8192)])
def test_slab_norm_fused_qkv_rotary_scatter_from_row_statistics(oracle, dev, m, h, hkv, d, k):
    """the decode step's first launch in that mode: norm (from the statistics) + qkv + rotation + KV scatter == the plain fused launch on
    the host-normalised rows, bit for bit (q, K and V buffers)"""
    from zhilight_amd import ops
    rng = np.random.default_rng(930 + m)
    n = (h + 2 * hkv) * d
    w = ops.W4MWeight.random(n, k, 128, dev)
    x = synth.act(rng, m, k, 2.0)
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    ss = ops.row_ss(_t(x, dev))
    rs = _rs_from_ss(_np(ss), k, 1e-5)
    xn = ((x.astype(np.float32) * rs[:, None]) * nw.astype(np.float32)[None, :]).astype(np.float16)
    lens = [int(v) for v in rng.integers(2, 6, m) * 32]
    pos = np.array([int(rng.integers(0, L)) for L in lens], np.int32)
    placement = pos.copy()
    placement[1] = -1
    cs, sn = oracle.rope_cos_sin(pos, d, 5e5, True, None)
    mk = lambda: [torch.full((L, hkv, d), 3.0, dtype=torch.float16, device=dev) for L in lens]
    k1, v1, k2, v2 = mk(), mk(), mk(), mk()
    lens_t, place_t = _t(np.array(lens, np.int32), dev), _t(placement, dev)
    assert ops.w4_row_ss_routes(m, [], [(w, True)], dev)
    q_got = ops.w4_qkv_rope_scatter(_t(x, dev), w, _t(cs, dev), _t(sn, dev), place_t, lens_t, ops.make_ptr_table(k1), ops.make_ptr_table(v1),
                                    h, hkv, d, norm_weight=_t(nw, dev), norm_eps=1e-5, row_ss=ss)
    q_ref = ops.w4_qkv_rope_scatter(_t(xn, dev), w, _t(cs, dev), _t(sn, dev), place_t, lens_t, ops.make_ptr_table(k2), ops.make_ptr_table(v2),
                                    h, hkv, d)
    assert torch.equal(q_got, q_ref)
    for a, b_ in zip(k1 + v1, k2 + v2):
        assert torch.equal(a, b_)

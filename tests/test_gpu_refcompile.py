"""The reference's OWN host code on the MI355X boundary: zhilight_amd/_ref/zl_reflinear*.so is the reference's
src/nn/linear/linear.cpp compiled UNMODIFIED (read in place from /root/reference by zhilight_amd.build.build_refcompile,
never copied) against hostcpp/refshim + bm_hip.h / bm_layer.h / bm_functions.h / nn_amd.cpp, linked with
libzhilight_amd.so.  Here nn::Linear -- the reference's class, constructor dispatch, load_state_dict and forward -- runs
on the GPU for the three flavours on the hot path and is checked against the oracle:
  GPTQ      Int4GPTQ: load_parameter -> gptq_shuffle / increase_zero / q4_to_q8 / 3x Transpose -> gptq_gemm_k_major with the
            raw k-major operands (re-tiled once through the weight-identity cache), plus the act-order variant
  AutoInt8  Int8Linear: quant_calc_scale at load and per call, the cublasLt IMMA call (refshim -> zl_int8_gemm_nt),
            quant_scale_back -- integer product exact, so the result is bit-identical to the oracle chain
  NoQuant   NormalLinear: functions::Gemm
The prebuilt module travels to the GPU box with the snapshot; nothing here reads /root/reference at run time."""
import os
import sys

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref(dev):
    from zhilight_amd import _lib, build
    _lib.lib()                      # libzhilight_amd.so resolved first (the module's rpath points at it as well)
    path = build.refcompile_target()
    if not os.path.exists(path):
        pytest.skip("zl_reflinear was not built (no reference tree at build time)")
    sys.path.insert(0, os.path.dirname(path))
    try:
        import zl_reflinear
    finally:
        sys.path.pop(0)
    return zl_reflinear


def _gptq_state(qw, qz, sc, g_idx=None):
    sd = {"l.qweight": np.ascontiguousarray(qw.view(np.int32)), "l.qzeros": np.ascontiguousarray(qz.view(np.int32)),
          "l.scales": np.ascontiguousarray(sc.view(np.float16))}
    if g_idx is not None:
        sd["l.g_idx"] = np.ascontiguousarray(g_idx.astype(np.int32))
    return sd


@pytest.mark.parametrize("k,n", [(4096, 6144), (1024, 272)])
def test_reference_int4gptq_layer(ref, oracle, k, n):
    rng = np.random.default_rng(k + n)
    g = 128
    qw, qz, sc = synth.gptq_hf(rng, k, n, g)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, g)
    ref.weight_cache_clear()
    lin = ref.RefLinear(k, n, 5, group_size=g)           # QuantType::GPTQ
    assert lin.layer_type() == "Linear"
    lin.load(_gptq_state(qw, qz, sc), "l")
    # the layer's own dequantised weight (get_dequant_weight -> dequant_k_major): bit-exact W16
    w16 = oracle.gptq_dequant_k_major(*km)
    assert np.array_equal(lin.dequant_weight().view(np.uint16), w16)
    for m in (1, 4, 8, 70):
        x = synth.act(rng, m, k)
        got = lin.forward(x).astype(np.float64)
        exact = oracle.gptq_gemm_k_major_exact(oracle.h2u(x), *km)
        if m > 32:    # the M-tiled kernel multiplies with W16, the reference's M > 40 arithmetic
            exact = oracle.gemm_nt(oracle.h2u(x), w16, None, exact=True)
        rms = np.sqrt((exact ** 2).mean())
        assert got.shape == (m, n)
        assert (np.abs(got - exact) <= 2.0 ** -10 * np.abs(exact) + 3e-5 * rms).all(), (m, float((np.abs(got - exact) / rms).max()))
    # one packed copy for the four calls: the raw operands were re-tiled once
    assert ref.weight_cache_size() == 1
    ref.weight_cache_clear()
    assert ref.weight_cache_size() == 0


def test_reference_int4gptq_layer_act_order(ref, oracle):
    """desc_act: Int4GPTQ::argsort_cpu -> gptq_shuffle(q_perm) -> int32_to_int16 / reverse_perm at load, permute_input in
    gptq_gemm_k_major."""
    rng = np.random.default_rng(17)
    k, n, g = 1024, 256, 128
    qw, qz, sc, g_idx, w16 = synth.gptq_act_order_hf(rng, k, n, g)
    lin = ref.RefLinear(k, n, 5, group_size=g, act_order=True)
    lin.load(_gptq_state(qw, qz, sc, g_idx), "l")
    for m in (1, 5, 70):
        x = synth.act(rng, m, k)
        got = lin.forward(x).astype(np.float64)
        want = x.astype(np.float64) @ w16.astype(np.float64).T
        rms = np.sqrt((want ** 2).mean())
        assert np.abs(got - want).max() <= 2.0 ** -10 * np.abs(want).max() + 6e-3 * rms, (m, np.abs(got - want).max() / rms)
    ref.weight_cache_clear()


@pytest.mark.parametrize("m", [1, 7, 32, 45])
def test_reference_int8linear_layer_bit_exact(ref, oracle, m):
    """AutoInt8: the weight is quantised per row at load (quant_calc_scale), the activations per token per call; the
    int8 x int8 product is exact and the scale-back is the reference's expression -> bit-identical to the oracle chain."""
    rng = np.random.default_rng(m)
    k, n = 1024, 384
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float16)
    x = synth.act(rng, m, k)
    lin = ref.RefLinear(k, n, 2)                          # QuantType::AutoInt8
    lin.load({"l.weight": w}, "l")
    got = lin.forward(x)
    wq, ws = oracle.quant_calc_scale(oracle.h2u(w))
    xq, xs = oracle.quant_calc_scale(oracle.h2u(x))
    acc = oracle.int8_gemm_nt(xq, wq)
    want = oracle.quant_scale_back(acc, xs, oracle.h2u(ws.astype(np.float16)))
    assert np.array_equal(got.view(np.uint16), want)


@pytest.mark.parametrize("m", [1, 3, 40])
def test_reference_normal_linear_layer(ref, oracle, m):
    rng = np.random.default_rng(100 + m)
    k, n = 2048, 512
    w = (rng.standard_normal((n, k)) * 0.03).astype(np.float16)
    x = synth.act(rng, m, k)
    lin = ref.RefLinear(k, n, 0)
    lin.load({"l.weight": w}, "l")
    got = lin.forward(x).astype(np.float64)
    exact = oracle.gemm_nt(oracle.h2u(x), oracle.h2u(w), None, exact=True)
    rms = np.sqrt((exact ** 2).mean())
    assert (np.abs(got - exact) <= 2.0 ** -10 * np.abs(exact) + 3e-5 * rms).all()


def test_reference_code_reports_what_is_off_the_boundary(ref):
    """Marlin is not built here: the reference's constructor path runs, the first kernel call says so."""
    lin = ref.RefLinear(1024, 256, 8)                     # QuantType::GPTQ_Marlin
    with pytest.raises(Exception):
        lin.forward(np.zeros((1, 1024), np.float16))

"""GPU parity of the two remaining formats of the W4 operator family:
  * native AWQ: the checkpoint tensors consumed as stored -- (K, N/8) words in AutoAWQ's interleaved nibble order --
    by zl_awq_dequantize / zl_awq_gemm, against the oracle's restatement of SURVEY A.9 (W16 = rn16(fp16(q - z) s), K tiles
    dealt round-robin to 32 splits, fp16 partials summed in fp32);
  * W4A8 (W4_INT8_ALGO): per-row int8 re-quantisation of the dequantised W4 matrix at load, int8 x int8 -> int32 forward,
    fp32-scale back -- integer results bit-exact."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.view(dtype)


@pytest.mark.parametrize("k,n,g", [(256, 128, 128), (4096, 4096, 128), (1056, 192, 32), (2048, 448, 64)])
def test_awq_dequantize_bit_exact(oracle, dev, k, n, g):
    from zhilight_amd import ops
    rng = np.random.default_rng(k + n)
    qw, qz, sc, _, _ = synth.awq_hf(rng, k, n, g)
    got = ops.awq_dequantize(_t(qw.view(np.int32), dev), _t(qz.view(np.int32), dev), _t(sc, dev, torch.float16), g)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), oracle.awq_dequantize(qw, qz, sc, g))


@pytest.mark.parametrize("m", [1, 3, 8, 9, 17])
@pytest.mark.parametrize("k,n,g", [(4096, 4096, 128), (1056, 192, 32), (14336, 256, 128)])
def test_awq_gemm_native_layout(oracle, dev, m, k, n, g):
    """against the reference-faithful restatement (fp16 split partials): at most one fp16 ulp apart on a small fraction
    of the outputs (the four waves of a split add their fp32 partial sums in a different association than the oracle's
    single chain), and inside north_star's 1e-3 of the exact product of the same W16"""
    from zhilight_amd import ops
    rng = np.random.default_rng(m + k + n)
    qw, qz, sc, _, _ = synth.awq_hf(rng, k, n, g)
    x = synth.act(rng, m, k)
    got = ops.awq_gemm(_t(x, dev), _t(qw.view(np.int32), dev), _t(qz.view(np.int32), dev), _t(sc, dev, torch.float16), g)
    gb = got.cpu().numpy().view(np.uint16)
    w16 = oracle.awq_dequantize(qw, qz, sc, g)
    ref = oracle.awq_gemm(oracle.h2u(x), w16, 32)
    rf = oracle.u2h(ref).astype(np.float64)
    d = np.abs(oracle.u2h(gb).astype(np.float64) - rf)
    # a flipped fp16 rounding of ONE split partial (magnitude ~ rms / sqrt(32) .. rms) moves the sum by one ulp of that partial
    # (or, when the sum itself sits on a rounding boundary, by one ulp of the output)
    tol = 2.0 ** -9 * np.maximum(np.abs(rf), np.sqrt((rf ** 2).mean()))
    assert (gb != ref).mean() < 0.02 and (d <= tol).all(), (float((gb != ref).mean()), float(d.max()))
    exact = oracle.awq_gemm(oracle.h2u(x), w16, exact=True)
    assert np.abs(oracle.u2h(gb).astype(np.float64) - exact).max() <= 1e-3 * np.abs(exact).max()


def test_awq_native_equals_the_rerouted_route(oracle, dev):
    """the same checkpoint through the default AWQ route (re-tiled once into ZLW4M, matrix-core kernel) and through the
    native-layout kernel: both within fp16 output rounding of the exact product of the SAME dequantised matrix"""
    from zhilight_amd import ops
    rng = np.random.default_rng(3)
    k, n, g = 2048, 512, 128
    qw, qz, sc, _, _ = synth.awq_hf(rng, k, n, g)
    x = synth.act(rng, 4, k)
    native = ops.awq_gemm(_t(x, dev), _t(qw.view(np.int32), dev), _t(qz.view(np.int32), dev), _t(sc, dev, torch.float16), g).float().cpu().numpy()
    # Int4GPTQ::preprocess_weight, is_awq branch (linear.cpp:1139-1143): shuffle_awq -> exllama (K/8, N) words, zeros un-shuffled
    qk = ops.transpose_2d(ops.shuffle_awq(_t(qw.view(np.int32), dev), True))
    zk = ops.transpose_2d(ops.q4_to_q8(ops.awq_un_shuffle(_t(qz.view(np.int32), dev).clone())))
    w = ops.W4MWeight.from_k_major(qk, zk, ops.transpose_2d(_t(sc, dev, torch.float16)), g)
    rerouted = ops.w4a16_gemm_mfma(_t(x, dev), w).float().cpu().numpy()
    exact = oracle.awq_gemm(oracle.h2u(x), oracle.awq_dequantize(qw, qz, sc, g), exact=True)
    rms = np.sqrt((exact ** 2).mean())
    assert np.abs(native - exact).max() <= 3e-3 * rms       # + the split partials' fp16 rounding and W16's own rounding
    assert np.abs(rerouted - exact).max() <= 3e-3 * rms      # the matrix-core kernel does not round s (q - z) to fp16


def test_awq_gemm_error_behaviour(dev):
    from zhilight_amd import ops
    from zhilight_amd._lib import ZLError
    z = lambda *s, dt=torch.int32: torch.zeros(*s, dtype=dt, device=dev)
    with pytest.raises(ZLError, match="OC is not multiple of cta_N = 64"):
        ops.awq_gemm(z(1, 128, dt=torch.float16), z(128, 4), z(1, 4), z(1, 32, dt=torch.float16), 128)
    with pytest.raises(ZLError, match="Group size should be a multiple of 32"):
        ops.awq_gemm(z(1, 128, dt=torch.float16), z(128, 8), z(8, 8), z(8, 64, dt=torch.float16), 16)
    with pytest.raises(ZLError, match="size K mismatch"):
        ops.awq_gemm(z(1, 256, dt=torch.float16), z(128, 8), z(1, 8), z(1, 64, dt=torch.float16), 128)


@pytest.mark.parametrize("m,k,n", [(41, 1024, 1280), (64, 4096, 2048), (300, 2048, 1536)])
def test_w4a8_int8_branch_bit_exact(oracle, dev, m, k, n):
    """gptq_gemm_k_major's W4_INT8 branch (M > W4_A8_M_THRES = 40, N > 1024): int8 codes of the weights and their fp32
    scales, quantised activation rows, the int32 product and the scaled-back outputs are all bit-identical to the oracle."""
    from zhilight_amd import ops
    rng = np.random.default_rng(m + n)
    qw, qz, sc = synth.gptq_hf(rng, k, n, 128)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, 128)
    w16 = oracle.gptq_dequant_k_major(*km)                                     # (N, K) fp16 bits
    rw8, rs = oracle.w4a8_weight_to_int8(w16)
    w = ops.W4Weight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), 128)
    w8, ws = ops.w4a8_weight_to_int8(w.dequant())
    assert np.array_equal(w8.cpu().numpy(), rw8) and np.array_equal(ws.cpu().numpy(), rs)
    x = synth.act(rng, m, k, 2.0)
    y = ops.w4a8_linear(_t(x, dev), w8, ws)
    rq, rsx = oracle.quant_calc_scale(oracle.h2u(x))
    ref = oracle.quant_scale_back_f32(oracle.int8_gemm_nt(rq, rw8), rsx, rs)
    assert np.array_equal(y.cpu().numpy().view(np.uint16), ref)
    # and the approximation itself stays close to the W4A16 product (what the mode trades: ~1e-2 of the output rms)
    exact = oracle.gemm_nt(oracle.h2u(x), w16, exact=True)
    assert np.abs(y.float().cpu().numpy() - exact).max() <= 0.1 * np.sqrt((exact ** 2).mean())


def test_awq_checkpoint_model_native_and_rerouted_routes(oracle, dev, monkeypatch):
    """A whole AWQ checkpoint (HF names, AutoAWQ "gemm" tensors) through LLaMA.encode on both routes of the reference:
    AWQ_USE_EXLLAMA=1 (default: re-tiled at load, fused matrix-core kernels) and AWQ_USE_EXLLAMA=0 (native layout,
    nn::awq::awq_gemm per linear), against the oracle composed from the integers the checkpoint was packed from."""
    from test_gpu_model import OracleModel
    from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig
    rng = np.random.default_rng(31)
    g = 128
    cfg = ModelConfig(num_layers=2, dim_model=1024, num_heads=8, dim_head=128, dim_ff=2048, vocab_size=512, num_kv_heads=2,
                      eps=1e-5, rope_theta=5e5, rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                                                               "high_freq_factor": 4.0, "original_max_position_embeddings": 8192})
    hd, kvd = cfg.num_heads * cfg.dim_head, cfg.num_kv_heads * cfg.dim_head
    sd, km = {}, {}
    tr = lambda a: np.ascontiguousarray(a.T)   # noqa: E731
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        sd[p + "input_layernorm.weight"] = (1 + 0.1 * rng.standard_normal(cfg.dim_model)).astype(np.float16)
        sd[p + "post_attention_layernorm.weight"] = (1 + 0.1 * rng.standard_normal(cfg.dim_model)).astype(np.float16)
        for name, din, dout in (("self_attn.q_proj", cfg.dim_model, hd), ("self_attn.k_proj", cfg.dim_model, kvd),
                                ("self_attn.v_proj", cfg.dim_model, kvd), ("self_attn.o_proj", hd, cfg.dim_model),
                                ("mlp.gate_proj", cfg.dim_model, cfg.dim_ff), ("mlp.up_proj", cfg.dim_model, cfg.dim_ff),
                                ("mlp.down_proj", cfg.dim_ff, cfg.dim_model)):
            qw, qz, sc, _, _ = synth.awq_hf(rng, din, dout, g)
            sc = (np.abs(rng.standard_normal(sc.shape)) * (0.5 / np.sqrt(din)) / 4 + 1e-4).astype(np.float16).view(np.uint16)
            sd[p + name + ".qweight"], sd[p + name + ".qzeros"], sd[p + name + ".scales"] = qw.view(np.int32), qz.view(np.int32), sc.view(np.float16)
            km[p + name] = (tr(oracle.awq_shuffle(qw, True)), tr(oracle.gptq_q4_to_q8(oracle.awq_un_shuffle(qz))), tr(sc))
    sd["model.embed_tokens.weight"] = (rng.standard_normal((cfg.vocab_size, cfg.dim_model)) * 0.5).astype(np.float16)
    sd["model.norm.weight"] = (1 + 0.1 * rng.standard_normal(cfg.dim_model)).astype(np.float16)
    sd["lm_head.weight"] = (rng.standard_normal((cfg.vocab_size, cfg.dim_model)) * 0.05).astype(np.float16)
    quant = QuantConfig.from_hf(dict(quant_method="awq", bits=4, group_size=g, zero_point=True, version="gemm"))
    om = OracleModel(oracle, cfg, {k: v for k, v in sd.items() if not k.endswith((".qweight", ".qzeros", ".scales"))}, g, 2, 64)
    om.km = km
    tokens = rng.integers(0, cfg.vocab_size, 2).astype(np.int32)
    ref, _ = om.step(tokens, [0, 0], flavour="E")
    outs = {}
    for route in ("1", "0"):
        monkeypatch.setenv("AWQ_USE_EXLLAMA", route)
        model = LLaMA(cfg, quant, dev).load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        ctx = model.new_context(2, 64, 0)
        ctx.tokens.copy_(torch.from_numpy(tokens))
        outs[route] = model.encode(ctx).float().cpu().numpy().astype(np.float64)
        scale = np.abs(ref).max()
        # native route: W16 carries one more fp16 rounding (rn16(d s)) and the split partials another
        assert np.abs(outs[route] - ref).max() <= (1e-3 if route == "1" else 3e-3) * scale, (route, np.abs(outs[route] - ref).max() / scale)
    assert np.abs(outs["0"] - outs["1"]).max() <= 3e-3 * np.abs(ref).max()


@pytest.mark.parametrize("dtype", [0, 1])
def test_fp8_casts_bit_exact(oracle, dev, dtype):
    """nn::fp8::calc_scale + T_KERNEL_cvt_half_fp8 (fp8_util.cu:56-78, 100-195): the per-tensor scale and every E4M3FN code equal
    the oracle's (whose cast is pinned by an exhaustive nearest-even search over all fp16 values, tests/test_oracle_selfcheck.py);
    subnormal codes, saturation at 448 and both activation types."""
    from zhilight_amd import ops
    rng = np.random.default_rng(17 + dtype)
    x = (rng.standard_normal((67, 512)) * np.exp(rng.uniform(-9, 3, (67, 512)))).astype(np.float32)
    x[0, :8] = [0.0, -0.0, 1e-4, -2e-3, 448.0, -470.0, 3e4, 0.0175]
    bits = oracle.f32_to_bf16(x) if dtype else oracle.h2u(x.astype(np.float16))
    tx = torch.from_numpy(bits.view(np.int16)).to(dev).view(torch.bfloat16 if dtype else torch.float16)
    for mx in (448.0, 256.0):
        s_ref = oracle.fp8_calc_scale(bits, mx, dtype)
        codes, s = ops.fp8_dynamic_scaled_quant(tx, mx)
        assert s.item() == s_ref
        assert np.array_equal(codes.cpu().numpy(), oracle.fp8_cvt_half(bits, s_ref, dtype))
    # a fixed scale of 1 exercises the saturation and the subnormal grid directly
    one = torch.ones(1, dtype=torch.float32, device=dev)
    assert np.array_equal(ops.fp8_cvt_half(tx, one).cpu().numpy(), oracle.fp8_cvt_half(bits, 1.0, dtype))


@pytest.mark.parametrize("m,k,n", [(41, 1024, 1280), (64, 4096, 2048), (300, 2048, 1536)])
def test_w4a8_fp8_branch(oracle, dev, m, k, n):
    """gptq_gemm_k_major's W4_FP8 branch (M > W4_A8_M_THRES = 40, N > 1024; q_gemm_k_major.cu:1003-1035): weight codes and scale
    (MAX_WEIGHT_E4M3 = 256) and activation codes and scale (MAX_ACT_E4M3 = 448) bit-identical to the oracle; the fp8 x fp8 product
    with fp32 accumulation within fp16 output rounding of the fp64 sum of the same codes (the reference's accumulation order is
    cuBLASLt's)."""
    from zhilight_amd import ops
    rng = np.random.default_rng(m + n + 1)
    qw, qz, sc = synth.gptq_hf(rng, k, n, 128)
    km = oracle.gptq_prepare_k_major(qw, qz, sc, 128)
    w16 = oracle.gptq_dequant_k_major(*km)                                     # (N, K) fp16 bits
    ws_ref = oracle.fp8_calc_scale(w16, 256.0)
    w8_ref = oracle.fp8_cvt_half(w16, ws_ref)
    w = ops.W4Weight.from_k_major(_t(km[0].view(np.int32), dev), _t(km[1], dev), _t(km[2], dev, torch.float16), 128)
    w8, ws = ops.w4a8_weight_to_fp8(w.dequant())
    assert ws.item() == ws_ref and np.array_equal(w8.cpu().numpy(), w8_ref)
    x = synth.act(rng, m, k, 2.0)
    xs_ref = oracle.fp8_calc_scale(oracle.h2u(x), 448.0)
    a8_ref = oracle.fp8_cvt_half(oracle.h2u(x), xs_ref)
    a8, sa = ops.fp8_dynamic_scaled_quant(_t(x, dev), 448.0)
    assert sa.item() == xs_ref and np.array_equal(a8.cpu().numpy(), a8_ref)
    y = ops.w4a8_fp8_linear(_t(x, dev), w8, ws).float().cpu().numpy().astype(np.float64)
    ref = oracle.u2h(oracle.fp8_gemm_nt(a8_ref, w8_ref, xs_ref, ws_ref)).astype(np.float64)
    assert (np.abs(y - ref) <= 2.0 ** -10 * np.abs(ref) + 1e-5 * np.abs(ref).max()).all()
    exact = oracle.gemm_nt(oracle.h2u(x), w16, exact=True)
    assert np.sqrt(((y - exact) ** 2).mean()) <= 0.08 * np.sqrt((exact ** 2).mean())   # what the mode trades: ~5 % rms (3-bit mantissas on both sides)

"""ctypes/numpy front-end of the CPU ORACLE (oracle/zl_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py.  The product package (zhilight_amd/) never imports this module.

fp16 / bf16 tensors travel as numpy uint16 arrays (raw bits); helpers `h2u`/`u2h` convert between
numpy float16 and the raw view.  dtype codes: 0 = fp16, 1 = bf16.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzl_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (make).  Building the checker is not using it."""
    src = os.path.join(_HERE, "zl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.zlo_f16_to_f32.restype = C.c_float
        _lib.zlo_bf16_to_f32.restype = C.c_float
        _lib.zlo_f32_to_f16.restype = C.c_uint16
        _lib.zlo_f32_to_f16.argtypes = [C.c_float]
        _lib.zlo_f64_to_f16.restype = C.c_uint16
        _lib.zlo_f64_to_f16.argtypes = [C.c_double]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i(x):
    return C.c_int64(int(x))


def _f(x):
    return C.c_float(float(x))


def h2u(a):
    """float16 ndarray -> uint16 raw bits (contiguous)."""
    return np.ascontiguousarray(a, dtype=np.float16).view(np.uint16)


def u2h(a):
    return np.ascontiguousarray(a, dtype=np.uint16).view(np.float16)


def bf16_to_f32(a):
    return (np.ascontiguousarray(a, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    return ((u + 0x7FFF + lsb) >> 16).astype(np.uint16)


def to_f32(a, dtype=0):
    return bf16_to_f32(a) if dtype else u2h(a).astype(np.float32)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ----------------------------------------------------------------------------- a4 layout
def gptq_shuffle(qweight):
    q = _c(qweight, np.uint32).copy()
    lib().zlo_gptq_shuffle(_p(q), _i(q.shape[0]), _i(q.shape[1]))
    return q


def gptq_increase_zero(qzeros):
    q = _c(qzeros, np.uint32).copy()
    lib().zlo_gptq_increase_zero(_p(q), _i(q.size))
    return q


def gptq_q4_to_q8(qzeros):
    q = _c(qzeros, np.uint32)
    out = np.empty(q.shape[:-1] + (q.shape[-1] * 8,), np.uint8)
    lib().zlo_gptq_q4_to_q8(_p(q), _p(out), _i(q.size))
    return out


def awq_un_shuffle(q):
    q = _c(q, np.uint32).copy()
    lib().zlo_awq_un_shuffle(_p(q), _i(q.shape[0]), _i(q.shape[1]))
    return q


def awq_shuffle(qweight, use_exllama=True):
    q = _c(qweight, np.uint32)
    k, n8 = q.shape
    out = np.empty((k // 8, n8 * 8), np.uint32)
    lib().zlo_awq_shuffle(_p(q), _p(out), _i(k), _i(n8 * 8), C.c_int(int(use_exllama)))
    return out


def gptq_prepare_k_major(qweight_hf, qzeros_hf, scales_hf, group_size):
    """HF (K/8,N) int32, (K/G,N/8) int32, (K/G,N) fp16-bits -> k-major (N,K/8), (N,K/G) u8, (N,K/G)."""
    qw, qz, sc = _c(qweight_hf, np.uint32), _c(qzeros_hf, np.uint32), _c(scales_hf, np.uint16)
    k, n = qw.shape[0] * 8, qw.shape[1]
    ng = k // group_size
    o_qw, o_qz, o_sc = np.empty((n, k // 8), np.uint32), np.empty((n, ng), np.uint8), np.empty((n, ng), np.uint16)
    lib().zlo_gptq_prepare_k_major(_p(qw), _p(qz), _p(sc), _i(k), _i(n), _i(group_size), _p(o_qw), _p(o_qz), _p(o_sc))
    return o_qw, o_qz, o_sc


def gptq_dequant_hf_naive(qweight_hf, qzeros_hf, scales_hf, group_size, g_idx=None):
    qw, qz, sc = _c(qweight_hf, np.uint32), _c(qzeros_hf, np.uint32), _c(scales_hf, np.uint16)
    k, n = qw.shape[0] * 8, qw.shape[1]
    gi = None if g_idx is None else _c(g_idx, np.int32)
    out = np.empty((n, k), np.float64)
    lib().zlo_gptq_dequant_hf_naive(_p(qw), _p(qz), _p(sc), _p(gi), _i(k), _i(n), _i(group_size), _p(out))
    return out


def gptq_reconstruct(qweight, qzeros_plus1, scales, g_idx=None):
    """nn::gptq::reconstruct_gptq (src/nn/quant/gptq/q_gemm.cu:641-676): the legacy (K/8, N) weight -- checkpoint word order,
    qzeros (K/G, N/8) as they sit on the device AFTER increase_zero (utils.cu:61-88) -- as fp16 bits (K, N):
        out[k][n] = hmul(int2half(q[k][n] - zero[g][n]), scale[g][n]),   g = g_idx[k] (None: k // group_size).
    q - zero is an integer of magnitude <= 15 (exact in fp16); __hmul is one fp16 rounding of the exact product."""
    qw, qz, sc = _c(qweight, np.uint32), _c(qzeros_plus1, np.uint32), _c(scales, np.uint16).view(np.float16)
    k, n, groups = qw.shape[0] * 8, qw.shape[1], qz.shape[0]
    q = np.zeros((k, n), np.int32)
    for j in range(8):
        q[j::8] = (qw >> np.uint32(4 * j)) & 0xF
    z = np.zeros((groups, n), np.int32)
    for j in range(8):
        z[:, j::8] = (qz >> np.uint32(4 * j)) & 0xF
    g = (np.arange(k) // (k // groups)) if g_idx is None else np.asarray(g_idx, np.int64)
    d = (q - z[g]).astype(np.float32)
    return (d * sc[g].astype(np.float32)).astype(np.float16).view(np.uint16)       # fp32 product of two halves is exact


def gptq_gemm_legacy_exact(x, qweight, qzeros_plus1, scales, g_idx=None):
    """the product nn::gptq::gptq_gemm stands for (q_gemm.cu:874-918, kernels :104-251 / :481-590) in fp64:
    y[m][n] = sum_k x[m][k] (q[k][n] - zero[g(k)][n]) scale[g(k)][n].  The reference's kernels add fp16 partials per 128-k
    block with atomicAdd in retirement order (non-deterministic), so there is no single R value to restate: implementations are
    held to the exact sum."""
    qw, qz, sc = _c(qweight, np.uint32), _c(qzeros_plus1, np.uint32), _c(scales, np.uint16).view(np.float16)
    k, n, groups = qw.shape[0] * 8, qw.shape[1], qz.shape[0]
    q = np.zeros((k, n), np.int32)
    for j in range(8):
        q[j::8] = (qw >> np.uint32(4 * j)) & 0xF
    z = np.zeros((groups, n), np.int32)
    for j in range(8):
        z[:, j::8] = (qz >> np.uint32(4 * j)) & 0xF
    g = (np.arange(k) // (k // groups)) if g_idx is None else np.asarray(g_idx, np.int64)
    w = (q - z[g]).astype(np.float64) * sc[g].astype(np.float64)
    return u2h(x).astype(np.float64) @ w


# ----------------------------------------------------------------------------- a2/a3/a5 GEMM
def gptq_gemm_k_major(x, qw, qz, sc, bias=None, sym=False, add_c=None):
    x, qw, qz, sc = _c(x, np.uint16), _c(qw, np.uint32), _c(qz, np.uint8), _c(sc, np.uint16)
    m, k = x.shape
    n = qw.shape[0]
    g = k // sc.shape[1]
    y = np.zeros((m, n), np.uint16) if add_c is None else _c(add_c, np.uint16).copy()
    b = None if bias is None else _c(bias, np.uint16)
    lib().zlo_gptq_gemm_k_major(_p(x), _p(qw), _p(qz), _p(sc), _p(b), _p(y), _i(m), _i(n), _i(k), _i(g),
                                C.c_int(int(sym)), C.c_int(0 if add_c is None else 1))
    return y


def gptq_moe_up(x, km1, km2, ids, n_shared=0, shared_base=0, sym=False, exp_parallel=False, world=1, rank=0):
    """km1 / km2: (qw (E, N, K/8) u32, qz (E, N, K/G) u8, sc (E, N, K/G) u16) stacks of gate / up; ids (M, top_k) int32"""
    x = _c(x, np.uint16)
    q1, z1, s1 = (_c(a, t) for a, t in zip(km1, (np.uint32, np.uint8, np.uint16)))
    q2, z2, s2 = (_c(a, t) for a, t in zip(km2, (np.uint32, np.uint8, np.uint16)))
    ids = _c(ids, np.int32)
    m, k = x.shape
    n = q1.shape[1]
    g = k // s1.shape[2]
    top_k = ids.shape[1]
    out = np.zeros((m, top_k + n_shared, n), np.uint16)
    lib().zlo_gptq_moe_up(_p(x), _p(q1), _p(z1), _p(s1), _p(q2), _p(z2), _p(s2), _p(ids), _p(out), _i(m), _i(n), _i(k), _i(g),
                          C.c_int(int(sym)), C.c_int(top_k), C.c_int(n_shared), C.c_int(shared_base), C.c_int(int(exp_parallel)),
                          C.c_int(world), C.c_int(rank))
    return out


def gptq_moe_down(a, km, ids, weights, n_shared=0, shared_base=0, sym=False, exp_parallel=False, world=1, rank=0, add_c=None):
    """a (M, top_k + n_shared, K) fp16 bits; km: (E, N, ...) stacks; weights (M, top_k) float32"""
    a = _c(a, np.uint16)
    q, z, s = (_c(t_, d) for t_, d in zip(km, (np.uint32, np.uint8, np.uint16)))
    ids, weights = _c(ids, np.int32), _c(weights, np.float32)
    m, _, k = a.shape
    n = q.shape[1]
    g = k // s.shape[2]
    top_k = ids.shape[1]
    out = np.zeros((m, n), np.uint16) if add_c is None else _c(add_c, np.uint16).copy()
    lib().zlo_gptq_moe_down(_p(a), _p(q), _p(z), _p(s), _p(ids), _p(weights), _p(out), _i(m), _i(n), _i(k), _i(g), C.c_int(int(sym)),
                            C.c_int(top_k), C.c_int(n_shared), C.c_int(shared_base), C.c_int(int(exp_parallel)), C.c_int(world),
                            C.c_int(rank), C.c_int(0 if add_c is None else 1))
    return out


def gptq_gemm_k_major_exact(x, qw, qz, sc, bias=None, sym=False):
    x, qw, qz, sc = _c(x, np.uint16), _c(qw, np.uint32), _c(qz, np.uint8), _c(sc, np.uint16)
    m, k = x.shape
    n = qw.shape[0]
    g = k // sc.shape[1]
    y = np.empty((m, n), np.float64)
    b = None if bias is None else _c(bias, np.uint16)
    lib().zlo_gptq_gemm_k_major_exact(_p(x), _p(qw), _p(qz), _p(sc), _p(b), _p(y), _i(m), _i(n), _i(k), _i(g),
                                      C.c_int(int(sym)))
    return y


def gptq_dequant_k_major(qw, qz, sc):
    qw, qz, sc = _c(qw, np.uint32), _c(qz, np.uint8), _c(sc, np.uint16)
    n, k = qw.shape[0], qw.shape[1] * 8
    out = np.empty((n, k), np.uint16)
    lib().zlo_gptq_dequant_k_major(_p(qw), _p(qz), _p(sc), _p(out), _i(n), _i(k), _i(k // sc.shape[1]))
    return out


def gptq_gemm_fuse_gate_in(x, w1, w2, sym=False):
    x = _c(x, np.uint16)
    qw1, qz1, sc1 = (_c(w1[0], np.uint32), _c(w1[1], np.uint8), _c(w1[2], np.uint16))
    qw2, qz2, sc2 = (_c(w2[0], np.uint32), _c(w2[1], np.uint8), _c(w2[2], np.uint16))
    m, k = x.shape
    n = qw1.shape[0]
    y = np.empty((m, n), np.uint16)
    lib().zlo_gptq_gemm_fuse_gate_in(_p(x), _p(qw1), _p(qz1), _p(sc1), _p(qw2), _p(qz2), _p(sc2), _p(y),
                                     _i(m), _i(n), _i(k), _i(k // sc1.shape[1]), C.c_int(int(sym)))
    return y


# ----------------------------------------------------------------------------- norm
def rmsnorm(x, w, eps, scale=1.0, x2=None, dtype=0):
    x, w = _c(x, np.uint16), _c(w, np.uint16)
    rows, dim = x.shape
    out = np.empty_like(x)
    x2c = None if x2 is None else _c(x2, np.uint16)
    out_sum = None if x2 is None else np.empty_like(x)
    lib().zlo_rmsnorm(_p(x), _p(w), _p(out), _i(rows), _i(dim), _f(eps), _f(scale), _p(x2c), _p(out_sum),
                      C.c_int(dtype))
    return (out, out_sum) if x2 is not None else out


def rmsnorm_exact(x, w, eps, scale=1.0, x2=None, dtype=0):
    x, w = _c(x, np.uint16), _c(w, np.uint16)
    rows, dim = x.shape
    out = np.empty((rows, dim), np.float64)
    x2c = None if x2 is None else _c(x2, np.uint16)
    lib().zlo_rmsnorm_exact(_p(x), _p(w), _p(out), _i(rows), _i(dim), _f(eps), _f(scale), _p(x2c), C.c_int(dtype))
    return out


# ----------------------------------------------------------------------------- rope
def rope_cos_sin(pos, d, base, neox=True, llama3=None):
    pos = _c(pos, np.int32)
    s = pos.size
    cs, sn = np.empty((s, d), np.float32), np.empty((s, d), np.float32)
    if llama3 is None:
        lib().zlo_rope_cos_sin(_p(pos), _p(cs), _p(sn), _i(s), _i(d), _f(base), C.c_int(int(neox)))
    else:
        factor, low, high, old = llama3
        lib().zlo_rope_cos_sin_llama3(_p(pos), _p(cs), _p(sn), _i(s), _i(d), _f(base), _f(factor), _f(low),
                                      _f(high), _f(old), C.c_int(int(neox)))
    return cs, sn


def rope_cos_sin_dynamic(pos, d, base, factor, max_pos, seq_len=None, neox=True):
    pos = _c(pos, np.int32)
    s = pos.size
    sl = None if seq_len is None else _c(seq_len, np.int32)
    cs, sn = np.empty((s, d), np.float32), np.empty((s, d), np.float32)
    lib().zlo_rope_cos_sin_dynamic(_p(pos), _p(sl) if sl is not None else None, _p(cs), _p(sn), _i(s), _i(d), _f(base),
                                   _f(factor), _f(max_pos), C.c_int(int(neox)))
    return cs, sn


def yarn_params(base, dim_head, original_max_position, factor, beta_fast=32, beta_slow=1, attn_factor=1.0, deepseek=False,
                mscale=0.0, mscale_all_dim=0.0):
    factor, mscale, mscale_all_dim = (C.c_float(v).value for v in (factor, mscale, mscale_all_dim))
    lo, hi, m = C.c_float(), C.c_float(), C.c_float()
    lib().zlo_yarn_params(C.c_double(base), C.c_int(dim_head), C.c_int(original_max_position), C.c_double(factor),
                          C.c_int(beta_fast), C.c_int(beta_slow), C.c_double(attn_factor), C.c_int(int(deepseek)),
                          C.c_double(mscale), C.c_double(mscale_all_dim), C.byref(lo), C.byref(hi), C.byref(m))
    return lo.value, hi.value, m.value


def rope_cos_sin_yarn(pos, d, base, factor, low, high, mscale, neox=True):
    pos = _c(pos, np.int32)
    s = pos.size
    cs, sn = np.empty((s, d), np.float32), np.empty((s, d), np.float32)
    lib().zlo_rope_cos_sin_yarn(_p(pos), _p(cs), _p(sn), _i(s), _i(d), _f(base), _f(factor), _f(low), _f(high), _f(mscale),
                                C.c_int(int(neox)))
    return cs, sn


def head_norm(x, w, heads, d, eps, mode=0, dtype=0):
    x, w = _c(x, np.uint16), _c(w, np.uint16)
    rows = x.shape[0]
    out = np.empty_like(x)
    lib().zlo_head_norm(_p(x), _p(w), _p(out), _i(rows), _i(heads), _i(d), _i(x.shape[1]), _i(x.shape[1]), _f(eps),
                        C.c_int(mode), C.c_int(dtype))
    return out


def rotary_embedding_qk(pos, x, h, hkv, d, theta, dtype=0):
    pos, x = _c(pos, np.int32), _c(x, np.uint16)
    s = pos.size
    q, k, v = np.empty((s, h * d), np.uint16), np.empty((s, hkv * d), np.uint16), np.empty((s, hkv * d), np.uint16)
    lib().zlo_rotary_embedding_qk(_p(pos), _p(x), _p(q), _p(k), _p(v), _i(s), _i(h), _i(hkv), _i(d), _f(theta),
                                  C.c_int(dtype))
    return q, k, v


def rope_qk_cache(cs, sn, x, h, hkv, d, neox=True, dtype=0):
    cs, sn, x = _c(cs, np.float32), _c(sn, np.float32), _c(x, np.uint16)
    s = cs.shape[0]
    q, k, v = np.empty((s, h * d), np.uint16), np.empty((s, hkv * d), np.uint16), np.empty((s, hkv * d), np.uint16)
    lib().zlo_rope_qk_cache(_p(cs), _p(sn), _p(x), _p(q), _p(k), _p(v), _i(s), _i(h), _i(hkv), _i(d),
                            C.c_int(int(neox)), C.c_int(dtype))
    return q, k, v


# ----------------------------------------------------------------------------- kv + attention
def _ptr_array(bufs):
    arr = (C.c_void_p * len(bufs))()
    for i, b in enumerate(bufs):
        arr[i] = b.ctypes.data
    return arr


def copy_to_rag_buffer2(placement, buf_lens, k_src, v_src, k_bufs, v_bufs, bshd=True):
    """In place on the per-task numpy uint16 buffers in k_bufs / v_bufs."""
    placement, buf_lens = _c(placement, np.int32), _c(buf_lens, np.int32)
    k_src, v_src = _c(k_src, np.uint16), _c(v_src, np.uint16)
    b, len_q, hkv, d = k_src.shape
    lib().zlo_copy_to_rag_buffer2(_p(placement), _p(buf_lens), _p(k_src), _p(v_src), _ptr_array(k_bufs),
                                  _ptr_array(v_bufs), _i(b), _i(len_q), _i(hkv), _i(d), C.c_int(int(bshd)))


def mqa_rag_buffer(q, buf_lens, k_bufs, v_bufs, mask, hkv, scale, bshd=True, dtype=0, num_split=0, exact=False):
    q, buf_lens, mask = _c(q, np.uint16), _c(buf_lens, np.int32), _c(mask, np.int8)
    b, len_q, h, d = q.shape
    kp, vp = _ptr_array(k_bufs), _ptr_array(v_bufs)
    if exact:
        out = np.empty(q.shape, np.float64)
        lib().zlo_mqa_rag_buffer_exact(_p(q), _p(buf_lens), kp, vp, _p(mask), _p(out), _i(b), _i(len_q), _i(h),
                                       _i(hkv), _i(d), _f(scale), C.c_int(int(bshd)), C.c_int(dtype))
        return out
    out = np.empty_like(q)
    if num_split:
        lib().zlo_mqa_rag_buffer_split_kv(_p(q), _p(buf_lens), kp, vp, _p(mask), _p(out), _i(b), _i(len_q), _i(h),
                                          _i(hkv), _i(d), _f(scale), C.c_int(int(bshd)), C.c_int(dtype),
                                          C.c_int(num_split))
    else:
        lib().zlo_mqa_rag_buffer(_p(q), _p(buf_lens), kp, vp, _p(mask), _p(out), _i(b), _i(len_q), _i(h), _i(hkv),
                                 _i(d), _f(scale), C.c_int(int(bshd)), C.c_int(dtype))
    return out


# ----------------------------------------------------------------------------- element-wise
def element_add_scale(a, b, scale=1.0, scale_residual=True, dtype=0):
    a, b = _c(a, np.uint16), _c(b, np.uint16)
    c = np.empty_like(a)
    lib().zlo_element_add_scale(_p(a), _p(b), _p(c), _i(a.size), _f(scale), C.c_int(int(scale_residual)),
                                C.c_int(dtype))
    return c


def silu_mul(a, b, dtype=0):
    a, b = _c(a, np.uint16), _c(b, np.uint16)
    out = np.empty_like(a)
    lib().zlo_silu_mul(_p(a), _p(b), _p(out), _i(a.size), C.c_int(dtype))
    return out


def gelu_mul(a, b, dtype=0):
    a, b = _c(a, np.uint16), _c(b, np.uint16)
    out = np.empty_like(a)
    lib().zlo_gelu_mul(_p(a), _p(b), _p(out), _i(a.size), C.c_int(dtype))
    return out


# ----------------------------------------------------------------------------- embedding / dense gemm
def embedding(ids, weight, scale=1.0, begin=0, end=None, dtype=0):
    ids, weight = _c(ids, np.int32), _c(weight, np.uint16)
    end = weight.shape[0] + begin if end is None else end
    out = np.empty((ids.size, weight.shape[1]), np.uint16)
    lib().zlo_embedding(_p(ids), _p(weight), _p(out), _i(ids.size), _i(weight.shape[1]), C.c_int32(begin),
                        C.c_int32(end), _f(scale), C.c_int(dtype))
    return out


def gemm_nt_exact_blas(x, w, bias=None, dtype=0):
    """zlo_gemm_nt_exact's value -- x (M, K) . w (N, K)^T + bias in fp64, every product of two T values exact -- through numpy's fp64
    matmul instead of the C triple loop (0.3 GFLOP/s per core: a 2048-token prompt through one Qwen2-72B-shaped layer took 12 minutes).
    BLAS sums in another order than the loop; in fp64 that moves the 16th digit of a sum that is rounded to T right after."""
    xf = to_f32(_c(x, np.uint16), dtype).astype(np.float64)
    wf = to_f32(_c(w, np.uint16), dtype).astype(np.float64)
    y = xf @ wf.T
    if bias is not None:
        y = y + to_f32(_c(bias, np.uint16), dtype).astype(np.float64)[None, :]
    return y


def gemm_nt(x, w, bias=None, alpha=1.0, dtype=0, exact=False):
    x, w = _c(x, np.uint16), _c(w, np.uint16)
    m, k = x.shape
    n = w.shape[0]
    b = None if bias is None else _c(bias, np.uint16)
    if exact:
        y = np.empty((m, n), np.float64)
        lib().zlo_gemm_nt_exact(_p(x), _p(w), _p(b), _p(y), _i(m), _i(n), _i(k), _f(alpha), C.c_int(dtype))
    else:
        y = np.empty((m, n), np.uint16)
        lib().zlo_gemm_nt(_p(x), _p(w), _p(b), _p(y), _i(m), _i(n), _i(k), _f(alpha), C.c_int(dtype))
    return y


# ----------------------------------------------------------------------------- int8
def quant_calc_scale(x, dtype=0):
    x = _c(x, np.uint16)
    m, k = x.shape
    q, s = np.empty((m, k), np.int8), np.empty((m,), np.float32)
    lib().zlo_quant_calc_scale(_p(x), _p(q), _p(s), _i(m), _i(k), C.c_int(dtype))
    return q, s


def rmsnorm_quant(x, w, eps, scale=1.0, dtype=0):
    x, w = _c(x, np.uint16), _c(w, np.uint16)
    rows, dim = x.shape
    out, q, s = np.empty_like(x), np.empty((rows, dim), np.int8), np.empty((rows,), np.float32)
    lib().zlo_rmsnorm_quant(_p(x), _p(w), _p(out), _p(q), _p(s), _i(rows), _i(dim), _f(eps), _f(scale),
                            C.c_int(dtype))
    return out, q, s


def int8_gemm_nt(a, b):
    a, b = _c(a, np.int8), _c(b, np.int8)
    m, k = a.shape
    n = b.shape[0]
    c = np.empty((m, n), np.int32)
    lib().zlo_int8_gemm_nt(_p(a), _p(b), _p(c), _i(m), _i(n), _i(k))
    return c


def quant_scale_back(c, sx, sy, dtype=0):
    c, sx, sy = _c(c, np.int32), _c(sx, np.float32), _c(sy, np.uint16)
    m, n = c.shape
    out = np.empty((m, n), np.uint16)
    lib().zlo_quant_scale_back(_p(c), _p(sx), _p(sy), _p(out), _i(m), _i(n), C.c_int(dtype))
    return out


def quant_back_act_mul(a, asx, asy, b, bsx, bsy, act="silu", dtype=0):
    a, b = _c(a, np.int32), _c(b, np.int32)
    m, n = a.shape
    out = np.empty((m, n), np.uint16)
    lib().zlo_quant_back_act_mul(_p(a), _p(_c(asx, np.float32)), _p(_c(asy, np.uint16)), _p(b),
                                 _p(_c(bsx, np.float32)), _p(_c(bsy, np.uint16)), _p(out), _i(m), _i(n),
                                 C.c_int(0 if act == "silu" else 1), C.c_int(dtype))
    return out


def quant_scale_back3(c, sx, sy, dim_q, dim_kv, dtype=0):
    c = _c(c, np.int32)
    m, n = c.shape
    q, k, v = np.empty((m, dim_q), np.uint16), np.empty((m, dim_kv), np.uint16), np.empty((m, dim_kv), np.uint16)
    lib().zlo_quant_scale_back3(_p(c), _p(_c(sx, np.float32)), _p(_c(sy, np.uint16)), _p(q), _p(k), _p(v), _i(m), _i(n),
                                _i(dim_q), _i(dim_kv), C.c_int(dtype))
    return q, k, v


def quant_back_element_add_scale(a, sx, sy, b, scale, dtype=0):
    a = _c(a, np.int32)
    m, n = a.shape
    out = np.empty((m, n), np.uint16)
    lib().zlo_quant_back_element_add_scale(_p(a), _p(_c(sx, np.float32)), _p(_c(sy, np.uint16)), _p(_c(b, np.uint16)),
                                           C.c_float(scale), _p(out), _i(m), _i(n), C.c_int(dtype))
    return out


def quant_back_transpose(inp, sx, sy, dtype=0):
    inp = _c(inp, np.int32)
    b, t, h, d = inp.shape
    out = np.empty((b, h, t, d), np.uint16)
    lib().zlo_quant_back_transpose(_p(inp), _p(_c(sx, np.float32)), _p(_c(sy, np.uint16)), _p(out), _i(b), _i(t), _i(h),
                                   _i(d), C.c_int(dtype))
    return out


def quant_back_copy_to_buffer(src, sx, sy, placement, dst, dtype=0):
    """src (batch, len_kv, heads, d) int32; dst (batch, heads, len_buf, d) uint16, updated in place."""
    src = _c(src, np.int32)
    b, t, h, d = src.shape
    assert dst.flags["C_CONTIGUOUS"] and dst.dtype == np.uint16
    len_buf = dst.shape[2]
    pl = None if placement is None else _c(placement, np.int32)
    lib().zlo_quant_back_copy_to_buffer(_p(src), _p(_c(sx, np.float32)), _p(_c(sy, np.uint16)), _p(pl), _p(dst), _i(b),
                                        _i(t), _i(h), _i(d), _i(len_buf), _i(t * h * d), _i(h * len_buf * d),
                                        _i(0 if pl is None or pl.ndim == 1 else pl.shape[1]), C.c_int(dtype))
    return dst


def quant_calc_scale_zp(x, q_zero=128, dtype=0):
    """INT8 KV cache rows: u8 codes with a zero point + fp32 scale per row."""
    x = _c(x, np.uint16)
    m, k = x.shape
    q, s = np.empty((m, k), np.uint8), np.empty((m,), np.float32)
    lib().zlo_quant_calc_scale_zp(_p(x), _p(q), _p(s), _i(m), _i(k), C.c_int(q_zero), C.c_int(dtype))
    return q, s


def mqa_rag_buffer_quant(q, buf_lens, k_bufs, v_bufs, k_scales, v_scales, mask, hkv, scale, bshd=True, dtype=0):
    """exact (fp64) decode attention over u8 K/V buffers + fp32 per-(key, kv head) scales."""
    q, buf_lens, mask = _c(q, np.uint16), _c(buf_lens, np.int32), _c(mask, np.int8)
    b, len_q, h, d = q.shape
    out = np.empty(q.shape, np.float64)
    lib().zlo_mqa_rag_buffer_quant_exact(_p(q), _p(buf_lens), _ptr_array(k_bufs), _ptr_array(v_bufs), _ptr_array(k_scales),
                                         _ptr_array(v_scales), _p(mask), _p(out), _i(b), _i(len_q), _i(h), _i(hkv), _i(d),
                                         _f(scale), C.c_int(int(bshd)), C.c_int(dtype))
    return out


# ---- native AWQ (A.9) and W4A8 ---------------------------------------------------------------------------------------
def awq_dequantize(qweight, qzeros, scales, group_size):
    """dequantize_weights: AWQ on-disk tensors -> W16 (K, N) fp16 bits, W16 = rn16(fp16(q - z) * s)"""
    qweight, qzeros, scales = _c(qweight, np.uint32), _c(qzeros, np.uint32), _c(scales, np.uint16)
    k, n = qweight.shape[0], qweight.shape[1] * 8
    out = np.empty((k, n), np.uint16)
    lib().zlo_awq_dequantize(_p(qweight), _p(qzeros), _p(scales), _p(out), _i(k), _i(n), _i(group_size))
    return out


def awq_gemm(x, w16, split_k_iters=32, exact=False):
    """awq_gemm on the dequantised matrix w16 (K, N): split-K with fp16 partials (R) or the exact fp64 product"""
    x, w16 = _c(x, np.uint16), _c(w16, np.uint16)
    m, k = x.shape
    n = w16.shape[1]
    if exact:
        y = np.empty((m, n), np.float64)
        lib().zlo_awq_gemm_exact(_p(x), _p(w16), _p(y), _i(m), _i(n), _i(k))
        return y
    y = np.empty((m, n), np.uint16)
    lib().zlo_awq_gemm(_p(x), _p(w16), _p(y), _i(m), _i(n), _i(k), _i(split_k_iters))
    return y


def w4a8_weight_to_int8(w16):
    """calc_w4a8_scale + KERNEL_dequant<int8_t,1>: W16 (N, K) fp16 bits -> (w8 int8 (N, K), scale fp32 (N))"""
    w16 = _c(w16, np.uint16)
    n, k = w16.shape
    w8, sc = np.empty((n, k), np.int8), np.empty((n,), np.float32)
    lib().zlo_w4a8_weight_to_int8(_p(w16), _p(w8), _p(sc), _i(n), _i(k))
    return w8, sc


def quant_scale_back_f32(c, sx, sy):
    c, sx, sy = _c(c, np.int32), _c(sx, np.float32), _c(sy, np.float32)
    m, n = c.shape
    out = np.empty((m, n), np.uint16)
    lib().zlo_quant_scale_back_f32(_p(c), _p(sx), _p(sy), _p(out), _i(m), _i(n))
    return out


# ---- INT8-compressed tensor-parallel reduce (model_context.cpp:244-326, quant_reduce_kernel.cu)
def quant_group_32(x, dtype=0):
    x = _c(x, np.uint16)
    groups = x.size // 32
    q, s = np.empty(x.shape, np.int8), np.empty(groups, np.uint16)
    lib().zlo_quant_group_32(_p(x), _p(q), _p(s), _i(groups), C.c_int(dtype))
    return q, s


def dequant_sum_quant_g32(my, q_others, scale_others, dtype=0):
    """my (M, 32) T bits; q_others (WS - 1, M, 32) int8; scale_others (WS - 1, M) T bits -> (q_sum (M, 32), scale_sum (M))"""
    my, q_others, scale_others = _c(my, np.uint16), _c(q_others, np.int8), _c(scale_others, np.uint16)
    groups, world = my.size // 32, q_others.shape[0] + 1
    q, s = np.empty(my.shape, np.int8), np.empty(groups, np.uint16)
    lib().zlo_dequant_sum_quant_g32(_p(my), _p(q_others), _p(scale_others), _p(q), _p(s), _i(groups), C.c_int(world), C.c_int(dtype))
    return q, s


def dequant_group_32(q, scale, dtype=0):
    q, scale = _c(q, np.int8), _c(scale, np.uint16)
    out = np.empty(q.shape, np.uint16)
    lib().zlo_dequant_group_32(_p(q), _p(scale), _p(out), _i(q.size // 32), C.c_int(dtype))
    return out


def reduce_tp_int8(parts, dtype=0):
    """ModelContext::reduce_tp_int8 over `parts` (one (n,) T-bit array per rank, n % (32 * WS) == 0): what EVERY rank ends
    with -- slice r of the result is rank r's own (unquantised) slice plus its peers' group-32 codes, re-quantised, then
    dequantised; the peers enter the sum in the order of increasing rank distance (rank + 1, rank + 2, ... mod WS)."""
    ws = len(parts)
    n = parts[0].size
    m = n // ws // 32
    qs = [quant_group_32(p, dtype) for p in parts]
    out = np.empty(n, np.uint16)
    for r in range(ws):
        lo, hi = r * m * 32, (r + 1) * m * 32
        order = [(r + i + 1) % ws for i in range(ws - 1)]
        qo = np.stack([qs[p][0].reshape(-1)[lo:hi].reshape(m, 32) for p in order])
        so = np.stack([qs[p][1][r * m:(r + 1) * m] for p in order])
        q_sum, s_sum = dequant_sum_quant_g32(np.ascontiguousarray(parts[r].reshape(-1)[lo:hi]).reshape(m, 32), qo, so, dtype)
        out[lo:hi] = dequant_group_32(q_sum, s_sum, dtype).reshape(-1)
    return out


# ---- W4A8 with FP8 activations (q_gemm_k_major.cu:1003-1035; fp8_util.cu)
def f32_to_e4m3(x):
    x = np.ascontiguousarray(x, np.float32)
    lib().zlo_f32_to_e4m3.restype = C.c_uint8
    lib().zlo_f32_to_e4m3.argtypes = [C.c_float]
    return np.array([lib().zlo_f32_to_e4m3(float(v)) for v in x.reshape(-1)], np.uint8).reshape(x.shape)


def e4m3_to_f32(c):
    c = np.ascontiguousarray(c, np.uint8)
    e, m = ((c >> 3) & 15).astype(np.int32), (c & 7).astype(np.float64)
    v = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    v = np.where((c & 0x7f) == 0x7f, np.nan, v)
    return np.where(c & 0x80, -v, v)


def fp8_calc_scale(x, max_e4m3=448.0, dtype=0):
    x = _c(x, np.uint16)
    s = C.c_float()
    lib().zlo_fp8_calc_scale(_p(x), _i(x.size), _f(max_e4m3), C.byref(s), C.c_int(dtype))
    return s.value


def fp8_cvt_half(x, scale, dtype=0):
    x = _c(x, np.uint16)
    out = np.empty(x.shape, np.uint8)
    lib().zlo_fp8_cvt_half(_p(x), _i(x.size), _f(scale), _p(out), C.c_int(dtype))
    return out


def fp8_gemm_nt(a, b, scale_a, scale_b):
    a, b = _c(a, np.uint8), _c(b, np.uint8)
    out = np.empty((a.shape[0], b.shape[0]), np.uint16)
    lib().zlo_fp8_gemm_nt(_p(a), _p(b), _f(scale_a), _f(scale_b), _p(out), _i(a.shape[0]), _i(b.shape[0]), _i(a.shape[1]))
    return out


# ---- f4 (config 5, first part): FP8 block linear + MoE router
def fp8_per_token_cast(x, col_major=True, max_e4m3=448.0, dtype=0):
    """(codes (m, n) uint8, scales fp32: (n/128, aligned_m) column-major or (m, n/128)); aligned_m = round_up(m, 4)"""
    x = _c(x, np.uint16)
    m, n = x.shape
    am = (m + 3) // 4 * 4
    out = np.empty((m, n), np.uint8)
    sc = np.zeros((n // 128, am) if col_major else (am, n // 128), np.float32)
    lib().zlo_fp8_per_token_cast(_p(x), _i(n), _p(out), _i(n), _p(sc), _i(am), _i(m), _i(n), C.c_int(int(col_major)), _f(max_e4m3), C.c_int(dtype))
    return out, sc


def fp8_block_dequant(w, scale, dtype=0):
    w, scale = _c(w, np.uint8), _c(scale, np.float32)
    out = np.empty(w.shape, np.uint16)
    lib().zlo_fp8_block_dequant(_p(w), _p(scale), _p(out), _i(w.shape[0]), _i(w.shape[1]), _i(scale.shape[1]), C.c_int(dtype))
    return out


def fp8_block_gemm(a, sa, w, sw, m_indices=None, dtype=1):
    """a (m, k) codes, sa (k/128, aligned_m) fp32, w (G, n, k) or (n, k) codes, sw (G, ceil(n/128), k/128); out (m, n) T bits"""
    a, sa, w, sw = _c(a, np.uint8), _c(sa, np.float32), _c(w, np.uint8), _c(sw, np.float32)
    m, k = a.shape
    n = w.shape[-2]
    out = np.zeros((m, n), np.uint16)
    mi = None if m_indices is None else _c(m_indices, np.int32)
    lib().zlo_fp8_block_gemm(_p(a), _p(sa), _i(sa.shape[1]), _p(w), _p(sw), _p(mi), _p(out), _i(m), _i(n), _i(k), C.c_int(dtype))
    return out


_SCORING = {"": 1, "softmax": 1, "sigmoid": 2, "linear": 3}


def moe_top_k_softmax(logits, k, top_k_ext=None, renormalize=False, weight_scale=1.0, scoring="softmax", dtype=0, num_worker=0):
    logits = _c(logits, np.uint16)
    t, e = logits.shape
    ext = top_k_ext or k
    v, idx = np.zeros((t, ext), np.float32), np.zeros((t, ext), np.int32)
    wl = np.zeros(max(num_worker, 1), np.int32)
    el = np.zeros(e, np.int32)
    lib().zlo_moe_top_k_softmax(_p(logits), _i(t), C.c_int(e), C.c_int(k), C.c_int(ext), C.c_int(int(renormalize)), _f(weight_scale),
                                C.c_int(_SCORING[scoring]), C.c_int(dtype), _p(v), _p(idx), _p(wl) if num_worker else None, _p(el),
                                C.c_int(num_worker))
    return v, idx, wl, el


def moe_group_topk(logits, bias, k, num_group, topk_group, top_k_ext=None, renormalize=True, weight_scale=1.0, scoring="sigmoid", dtype=0,
                   num_worker=0):
    logits = _c(logits, np.uint16)
    t, e = logits.shape
    ext = top_k_ext or k
    b = None if bias is None else _c(bias, np.float32)
    v, idx = np.zeros((t, ext), np.float32), np.zeros((t, ext), np.int32)
    wl = np.zeros(max(num_worker, 1), np.int32)
    el = np.zeros(e, np.int32)
    lib().zlo_moe_group_topk(_p(logits), _p(b), _i(t), C.c_int(e), C.c_int(k), C.c_int(ext), C.c_int(int(renormalize)), _f(weight_scale),
                             C.c_int(_SCORING[scoring]), C.c_int(num_group), C.c_int(topk_group), C.c_int(dtype), _p(v), _p(idx),
                             _p(wl) if num_worker else None, _p(el), C.c_int(num_worker))
    return v, idx, wl, el


# ---- MoE dispatch / combine
def moe_sum_experts(inp, index, weight, dtype=0):
    inp, index, weight = _c(inp, np.uint16), _c(index, np.int32), _c(weight, np.float32)
    seq, k = weight.shape
    out = np.empty((seq, inp.shape[1]), np.uint16)
    lib().zlo_moe_sum_experts(_p(inp), _p(index), _p(weight), _p(out), _i(seq), C.c_int(k), _i(inp.shape[1]), C.c_int(dtype))
    return out


def moe_sum_experts_arr(inputs, experts, index, weight, dim_model, exp_parallel=False, world_size=1, local_rank=0, dtype=0):
    inputs = [None if a is None else _c(a, np.uint16) for a in inputs]
    arr = (C.c_void_p * len(inputs))(*[None if a is None else a.ctypes.data for a in inputs])
    experts, weight = _c(experts, np.int32), _c(weight, np.float32)
    idx = None if index is None else _c(index, np.int32)
    seq, k = weight.shape
    out = np.empty((seq, dim_model), np.uint16)
    lib().zlo_moe_sum_experts_arr(arr, _p(experts), _p(idx), _p(weight), _p(out), _i(seq), C.c_int(k), _i(dim_model), C.c_int(int(exp_parallel)),
                                  C.c_int(world_size), C.c_int(local_rank), C.c_int(dtype))
    return out


def moe_route_shared_lb(exp_ids, worker_load, expert_load, top_k, num_local_experts):
    """in place on copies; returns (exp_ids, worker_load, expert_load)"""
    exp_ids, wl, el = _c(exp_ids, np.int32).copy(), _c(worker_load, np.int32).copy(), _c(expert_load, np.int32).copy()
    base = wl.copy()
    seq, ext = exp_ids.shape
    ws = wl.size
    max_load = (exp_ids.size + ws - 1) // ws
    lib().zlo_moe_route_shared_lb(_p(exp_ids), _p(base), _p(wl), _p(el), C.c_int(max_load), C.c_int(ws), _i(seq), C.c_int(top_k), C.c_int(ext),
                                  C.c_int(num_local_experts))
    return exp_ids, wl, el


def moe_plus_for_sort(exp_ids, num_experts, world_size):
    exp_ids = _c(exp_ids, np.int32)
    out = np.empty_like(exp_ids)
    lib().zlo_moe_plus_for_sort(_p(exp_ids), _p(out), C.c_int(num_experts), C.c_int(world_size), _i(exp_ids.size))
    return out


def moe_calc_reverse_idx(exp_ids, indices, all_loads, num_experts, world_size=1, sorted_by_rank=False):
    exp_ids, indices, all_loads = _c(exp_ids, np.int32), _c(indices, np.int32), _c(all_loads, np.int32)
    off = np.zeros(num_experts, np.int32)
    rev = np.zeros(indices.size, np.int32)
    lib().zlo_moe_calc_reverse_idx(_p(exp_ids), _p(indices), _p(all_loads), C.c_int(num_experts), C.c_int(world_size), C.c_int(int(sorted_by_rank)),
                                   _p(off), _p(rev), _i(indices.size))
    return rev, off


def moe_fill_m_indices(all_loads, block_m, num_experts, rank=0, ws=1):
    all_loads = _c(all_loads, np.int32)
    local = all_loads[rank:num_experts:ws]
    pad = np.zeros(int(local.sum()), np.int32)
    mi = np.zeros(int(((local + block_m - 1) // block_m * block_m).sum()), np.int32)
    lib().zlo_moe_fill_m_indices.restype = C.c_int
    total = lib().zlo_moe_fill_m_indices(_p(all_loads), C.c_int(block_m), C.c_int(num_experts), C.c_int(rank), C.c_int(ws), _p(pad), _p(mi))
    return mi, pad, total


# ---- f4: MLA decode attention over the latent cache
def mla_decode_attn(q_adj, buf_lens, valid_lens, kv_bufs, kv_rank=512, rope_dim=64, scale=1.0, dtype=0, flavour="E"):
    """q_adj (b, h, kv_rank + rope_dim) T bits; kv_bufs: list of (len_buf, kv_rank + rope_dim) T-bit arrays; out (b, h, kv_rank)"""
    q_adj = _c(q_adj, np.uint16)
    b, h, _ = q_adj.shape
    bufs = [_c(a, np.uint16) for a in kv_bufs]
    arr = _ptr_array(bufs)
    bl = _c(buf_lens, np.int32)
    vl = None if valid_lens is None else _c(valid_lens, np.int32)
    out = np.zeros((b, h, kv_rank), np.uint16)
    lib().zlo_mla_decode_attn(_p(q_adj), _p(bl), _p(vl), arr, _p(out), _i(b), _i(h), C.c_int(kv_rank), C.c_int(rope_dim), _f(scale),
                              C.c_int(dtype), C.c_int(0 if flavour == "E" else 1))
    return out


# ---- the batch generator's logit post-processing (numpy restatements; test infrastructure like everything in oracle/) -------------------------------
# src/generator/beam_util.cu:19-128, 130-157, 199-241; 3rd/bmengine/bmengine/functions/softmax.cu:8-30, topk.cu:280-293.  The reference reduces in
# fp32 over 1024 threads; these sum in fp64 -- the tests allow one rounding of T plus the fp32 reduction noise.  x: (rows, n) float64 VALUES of the T logits.
def log_softmax_bias_ref(x, bias, temperature=0.0):
    x = np.asarray(x, np.float64)
    m = np.maximum(x.max(axis=1, keepdims=True), -1e20)
    d = (x - m) / temperature if temperature != 0.0 else (x - m)
    s = np.exp(d).sum(axis=1, keepdims=True) + 1e-20
    return d - np.log(s) + np.asarray(bias, np.float64).reshape(-1, 1)


def softmax_rows_ref(x, temperature=1.0):
    x = np.asarray(x, np.float64)
    m = np.maximum(x.max(axis=1, keepdims=True), -1e20) / temperature
    e = np.exp(x / temperature - m)
    return e / (e.sum(axis=1, keepdims=True) + 1e-20)


def topk_rows_ref(x, top):
    """values (descending) and int32 positions; ties go to the lower index (stable sort of the negated values)"""
    x = np.asarray(x)
    idx = np.argsort(-x.astype(np.float64), axis=1, kind="stable")[:, :top].astype(np.int32)
    return np.take_along_axis(x, idx, axis=1), idx


def repetition_penalty_ref(logits_T, factor_T, presence_T, tokens, batch_ids, rnd):
    """logits_T (rows, vocab) float64 values of T; factor_T / presence_T: the penalties ALREADY rounded to T (the kernel casts them first); rnd rounds a
    float64 array to T.  Sequential like the kernel for distinct (row, token) pairs."""
    out = np.array(logits_T, np.float64)
    for i, (t, b) in enumerate(zip(tokens, batch_ids)):
        l = out[b, t]
        if presence_T is not None and presence_T[i] != 0.0:
            out[b, t] = rnd(np.array([l - presence_T[i]]))[0]
        else:
            out[b, t] = rnd(np.array([l * factor_T[i] if l < 0 else l / factor_T[i]]))[0]
    return out

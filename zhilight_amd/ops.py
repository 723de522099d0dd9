"""Operator-level Python mirror of the reference's `nn::` / `gptq::` / `int8_op::` free functions
(the "bmengine op" surface of SURVEY.md 8b) on top of the C ABI.

PyTorch is used for device memory and streams only: every function takes CUDA tensors, allocates its
output like the reference op does with `ctx.tensor(...)`, and enqueues ONE C-ABI launcher on the
current torch stream.  There is no CPU / eager fallback: without the HIP library these raise.

Names, argument meaning and error behaviour follow the reference (citations on each function;
paths relative to the ZhiLight tree).
"""
import ctypes as C
import os
import math

import torch

from . import _lib
from ._lib import W4Layout, W4Opts, ZLError, check, lib

F16, BF16 = 0, 1
EPI_BIAS, EPI_ADD_C, EPI_RESIDUAL, EPI_SILU_MUL, EPI_SILU_MUL_F32 = 1, 2, 4, 8, 16


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _i(v):
    return C.c_int64(int(v))


def _f(v):
    return C.c_float(float(v))


def _chk_out(out, m, n, dtype, device, what):
    """a caller-supplied output: (m, n) elements of `dtype`, contiguous rows, on the input's device (the launchers write
    m * n elements at out.data_ptr(): a wrong one is a silent overrun)"""
    if out.dtype != dtype or out.device != device or not out.is_contiguous() or out.numel() != m * n:
        raise ZLError(f"{what}: Wrong output size() / dtype / layout")


def _dt(t):
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.bfloat16:
        return BF16
    raise ZLError(f"unsupported activation dtype {t.dtype}")


def _dt_logits(t):
    """router logits: T, or fp32 (ZL_F32 = 2) -- the reference's router Linear writes fp32 logits (feedforward.cpp:285-286)"""
    return 2 if t.dtype == torch.float32 else _dt(t)


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise ZLError("tensors must be contiguous CUDA tensors")


# --------------------------------------------------------------------------------------------------
# a4: load-time layout transforms  (src/nn/quant/gptq/gptq.h:24-49,141-148)
# --------------------------------------------------------------------------------------------------
def gptq_shuffle(qweight):
    """In place on a (K/8, N) int32 tensor -- nn::gptq::gptq_shuffle without act-order."""
    _chk_cuda(qweight)
    check(lib().zl_gptq_shuffle(_p(qweight), _i(qweight.shape[0]), _i(qweight.shape[1]), _stream()), "gptq_shuffle")
    return qweight


def increase_zero(qzeros):
    _chk_cuda(qzeros)
    check(lib().zl_gptq_increase_zero(_p(qzeros), _i(qzeros.numel()), _stream()), "increase_zero")
    return qzeros


def q4_to_q8(qzeros):
    _chk_cuda(qzeros)
    out = torch.empty(qzeros.shape[:-1] + (qzeros.shape[-1] * 8,), dtype=torch.uint8, device=qzeros.device)
    check(lib().zl_gptq_q4_to_q8(_p(qzeros), _p(out), _i(qzeros.numel()), _stream()), "q4_to_q8")
    return out


def transpose_2d(t):
    """functions::Transpose for a 2-D tensor of 1/2/4-byte elements."""
    _chk_cuda(t)
    rows, cols = t.shape
    out = torch.empty((cols, rows), dtype=t.dtype, device=t.device)
    check(lib().zl_transpose_2d(_p(t), _p(out), _i(rows), _i(cols), C.c_int(t.element_size()), _stream()), "transpose")
    return out


def awq_un_shuffle(q):
    _chk_cuda(q)
    check(lib().zl_awq_un_shuffle(_p(q), _i(q.shape[0]), _i(q.shape[1]), _stream()), "un_shuffle")
    return q


def shuffle_awq(qweight, use_exllama=True):
    _chk_cuda(qweight)
    k, n8 = qweight.shape
    out = torch.empty((k // 8, n8 * 8), dtype=qweight.dtype, device=qweight.device)
    check(lib().zl_awq_shuffle(_p(qweight), _p(out), _i(k), _i(n8 * 8), C.c_int(int(use_exllama)), _stream()),
          "shuffle_awq")
    return out


class W4Weight:
    """A W4 (GPTQ / AWQ-as-exllama) linear weight in the gfx950-native ZLW4 layout (include/zhilight_amd.h).

    Plays the role of Int4GPTQ's device-side state after load_state_dict -> preprocess_weight ->
    transpose_weight (src/nn/linear/linear.cpp:1139-1244)."""

    def __init__(self, n, k, group_size, qw, scales, zeros, sym=False, row_interleave=False):
        self.n, self.k, self.group_size = n, k, group_size
        self.qw, self.scales, self.zeros = qw, scales, zeros
        self.sym, self.row_interleave = sym, row_interleave

    @staticmethod
    def layout(n, k, group_size):
        L = W4Layout()
        check(lib().zl_w4_layout(_i(n), _i(k), _i(group_size), C.byref(L)), "w4_layout")
        return L

    @classmethod
    def from_k_major(cls, qweight_km, qzeros_km, scales_km, group_size, sym=False, row_interleave=False):
        """qweight (N,K/8) int32 shuffled words, qzeros (N,K/G) uint8, scales (N,K/G) fp16."""
        _chk_cuda(qweight_km, qzeros_km, scales_km)
        n, k = qweight_km.shape[0], qweight_km.shape[1] * 8
        L = cls.layout(n, k, group_size)
        dev = qweight_km.device
        qw = torch.empty(L.qw_bytes // 4, dtype=torch.int32, device=dev)
        sc = torch.empty(L.scales_bytes // 2, dtype=torch.float16, device=dev)
        zs = torch.empty(L.zeros_bytes // 2, dtype=torch.int16, device=dev)
        check(lib().zl_w4_pack(_p(qweight_km), _p(qzeros_km), _p(scales_km), _i(n), _i(k), _i(group_size),
                               C.c_int(int(row_interleave)), _p(qw), _p(sc), _p(zs), _stream()), "w4_pack")
        return cls(n, k, group_size, qw, sc, zs, sym, row_interleave)

    @classmethod
    def from_hf_gptq(cls, qweight, qzeros, scales, group_size, sym=False, row_interleave=False):
        """HF / AutoGPTQ tensors: qweight (K/8,N) int32, qzeros (K/G,N/8) int32, scales (K/G,N) fp16.
        Same steps as the reference's load path: shuffle, +1 zeros, nibble->byte, transpose x3."""
        qw = gptq_shuffle(qweight.clone())
        qz = q4_to_q8(increase_zero(qzeros.clone()))
        return cls.from_k_major(transpose_2d(qw), transpose_2d(qz), transpose_2d(scales.contiguous()), group_size,
                                sym, row_interleave)

    def dequant(self):
        """(N, K) fp16 = rn16(rn16(q - z) * s) -- nn::gptq::dequant_k_major(out_type=0)."""
        out = torch.empty((self.n, self.k), dtype=torch.float16, device=self.qw.device)
        check(lib().zl_w4_dequant(_p(self.qw), _p(self.scales), _p(self.zeros), _i(self.n), _i(self.k),
                                  _i(self.group_size), _p(out), _stream()), "w4_dequant")
        return out

    def nbytes(self):
        return self.qw.numel() * 4 + self.scales.numel() * 2 + self.zeros.numel() * 2

    @classmethod
    def random(cls, n, k, group_size, device, gen=None, scale_mag=0.005, sym=False, row_interleave=False):
        """Synthetic weight generated directly in the packed layout (benchmarks: no checkpoint I/O)."""
        L = cls.layout(n, k, group_size)
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (L.qw_bytes // 4,), dtype=torch.int32, device=device, generator=gen)
        sc = (torch.rand(L.scales_bytes // 2, device=device, generator=gen) * scale_mag + 1e-4).to(torch.float16)
        zs = torch.randint(-2 ** 15, 2 ** 15 - 1, (L.zeros_bytes // 2,), dtype=torch.int16, device=device, generator=gen)
        return cls(n, k, group_size, qw, sc, zs, sym, row_interleave)



class W4MoEWeight:
    """The experts of one MoE projection, each packed like a W4Weight, stacked `stride` elements apart (what the reference
    keeps as (E, N, K/8) / (E, N, K/G) tensors for its fused MoE kernels, src/nn/feedforward/feedforward.cpp FUSE_GPTQ_MOE).
    gate / up: pass the (2 n_ff, K) [gate; up] k-major tensors of every expert with row_interleave=True."""

    def __init__(self, experts, n, k, group_size, qw, scales, zeros, row_interleave):
        self.experts, self.n, self.k, self.group_size = experts, n, k, group_size
        self.qw, self.scales, self.zeros, self.row_interleave = qw, scales, zeros, row_interleave
        L = W4Weight.layout(n, k, group_size)
        self.stride_bytes = (L.qw_bytes, L.scales_bytes, L.zeros_bytes)

    @classmethod
    def from_k_major(cls, qweights, qzeros, scales, group_size, row_interleave=False):
        """lists (one entry per expert) of qweight (N, K/8) int32, qzeros (N, K/G) uint8, scales (N, K/G) fp16"""
        ws = [W4Weight.from_k_major(q, z, s, group_size, False, row_interleave) for q, z, s in zip(qweights, qzeros, scales)]
        w0 = ws[0]
        return cls(len(ws), w0.n, w0.k, group_size, torch.stack([w.qw for w in ws]).contiguous(),
                   torch.stack([w.scales for w in ws]).contiguous(), torch.stack([w.zeros for w in ws]).contiguous(), row_interleave)

    def nbytes(self):
        return self.qw.numel() * 4 + self.scales.numel() * 2 + self.zeros.numel() * 2


def moe_up(x, w: W4MoEWeight, expert_ids, n_shared=0, exp_parallel=False, world_size=1, rank=0, out=None):
    """nn::gptq::gemm_moe_up (q_gemm_k_major.cu:392-455): out (M, top_k + n_shared, n_ff) = silu(x . gate_e) * (x . up_e) for
    the token's experts; the shared experts are the LAST n_shared of the stack."""
    if x.dtype != torch.float16 or not w.row_interleave:
        raise ZLError("moe_up: half activations, row-interleaved [gate; up] experts")
    _chk_cuda(x, expert_ids)
    m, k = x.shape
    if k != w.k or expert_ids.dtype != torch.int32 or expert_ids.shape[0] != m:
        raise ZLError("moe_up: shape / dtype mismatch")
    top_k, n_ff = expert_ids.shape[1], w.n // 2
    if out is None:
        out = torch.empty((m, top_k + n_shared, n_ff), dtype=torch.float16, device=x.device)
    else:
        _chk_out(out, m * (top_k + n_shared), n_ff, torch.float16, x.device, "moe_up")
    sq, ss, sz = w.stride_bytes
    check(lib().zl_w4a16_moe_up(_p(x), _i(x.stride(0)), _p(w.qw), _p(w.scales), _p(w.zeros), _i(sq), _i(ss), _i(sz), _p(expert_ids),
                                _p(out), _i(m), _i(n_ff), _i(k), _i(w.group_size), C.c_int(top_k), C.c_int(n_shared),
                                C.c_int(w.experts - n_shared), C.c_int(int(exp_parallel)), C.c_int(world_size), C.c_int(rank),
                                _stream()), "w4a16_moe_up")
    return out


def moe_down(a, w: W4MoEWeight, expert_ids, expert_weights, n_shared=0, exp_parallel=False, world_size=1, rank=0, out=None,
             add_c=False):
    """nn::gptq::gemm_moe_down (q_gemm_k_major.cu:457-520): out (M, N) = sum over the token's experts of weight * (a[m, t] . W_e)
    (+ out with add_c); a (M, top_k + n_shared, K)."""
    if a.dtype != torch.float16 or w.row_interleave:
        raise ZLError("moe_down: half activations, plain experts")
    _chk_cuda(a, expert_ids, expert_weights)
    m, t, k = a.shape
    top_k = expert_ids.shape[1]
    if k != w.k or t != top_k + n_shared or expert_ids.dtype != torch.int32 or expert_weights.dtype != torch.float32 or \
            tuple(expert_weights.shape) != tuple(expert_ids.shape) or expert_ids.shape[0] != m:
        raise ZLError("moe_down: shape / dtype mismatch")
    if out is None:
        if add_c:
            raise ZLError("moe_down: add_c needs the output to add to")
        out = torch.empty((m, w.n), dtype=torch.float16, device=a.device)
    else:
        _chk_out(out, m, w.n, torch.float16, a.device, "moe_down")
    sq, ss, sz = w.stride_bytes
    check(lib().zl_w4a16_moe_down(_p(a), _i(k), _p(w.qw), _p(w.scales), _p(w.zeros), _i(sq), _i(ss), _i(sz), _p(expert_ids),
                                  _p(expert_weights), _p(out), _i(m), _i(w.n), _i(k), _i(w.group_size), C.c_int(top_k),
                                  C.c_int(n_shared), C.c_int(w.experts - n_shared), C.c_int(int(exp_parallel)), C.c_int(world_size),
                                  C.c_int(rank), C.c_int(int(add_c)), _stream()), "w4a16_moe_down")
    return out


def w4a16_gemm(x, w, bias=None, residual=None, out=None, norm_weight=None, norm_eps=1e-5, epilogue=0):
    """y = x . dequant(W)^T with optional fused RMSNorm prologue and bias / ADD_C / residual / silu*mul
    epilogue -- nn::gptq::gptq_gemm_k_major (M <= 40 branch) and nn::gptq::gemm_fuse_gate_in
    (src/nn/quant/gptq/q_gemm_k_major.cu:957-1116, 765-829)."""
    if x.dtype != torch.float16:
        raise ZLError("A must be half")  # q_gemm_k_major.cu:989
    _chk_cuda(x, bias, residual, norm_weight)
    x2 = x.reshape(-1, x.shape[-1])
    m, k = x2.shape
    if k != w.k:
        raise ZLError("size K mismatch")
    silu = epilogue & (EPI_SILU_MUL | EPI_SILU_MUL_F32)
    n_out = w.n // 2 if silu else w.n
    if out is None:
        out = torch.empty((m, n_out), dtype=torch.float16, device=x.device)
    elif tuple(out.shape[-2:]) != (m, n_out) and out.numel() != m * n_out:
        raise ZLError("Wrong output size()")
    if bias is not None:
        epilogue |= EPI_BIAS
    check(lib().zl_w4a16_gemm(_p(x2), _i(x2.stride(0)), _p(w.qw), _p(w.scales), _p(w.zeros), _p(bias), _p(residual),
                              _p(out), _i(m), _i(w.n), _i(k), _i(w.group_size), C.c_int(int(w.sym)),
                              _p(norm_weight), _f(norm_eps), C.c_int(epilogue), _stream()), "w4a16_gemm")
    return out


class W4MWeight:
    """The same W4 linear weight in the MFMA-oriented ZLW4M layout (16-row x 128-k tiles + per-row meta)."""

    def __init__(self, n, k, group_size, qw, meta, row_interleave=False):
        self.n, self.k, self.group_size = n, k, group_size
        self.qw, self.meta, self.row_interleave = qw, meta, row_interleave

    @staticmethod
    def layout(n, k, group_size):
        L = W4Layout()
        check(lib().zl_w4m_layout(_i(n), _i(k), _i(group_size), C.byref(L)), "w4m_layout")
        return L

    @classmethod
    def from_k_major(cls, qweight_km, qzeros_km, scales_km, group_size, row_interleave=False):
        _chk_cuda(qweight_km, qzeros_km, scales_km)
        n, k = qweight_km.shape[0], qweight_km.shape[1] * 8
        L = cls.layout(n, k, group_size)
        dev = qweight_km.device
        qw = torch.empty(L.qw_bytes // 4, dtype=torch.int32, device=dev)
        meta = torch.empty(L.scales_bytes // 4, dtype=torch.int32, device=dev)
        check(lib().zl_w4m_pack(_p(qweight_km), _p(qzeros_km), _p(scales_km), _i(n), _i(k), _i(group_size),
                                C.c_int(int(row_interleave)), _p(qw), _p(meta), _stream()), "w4m_pack")
        return cls(n, k, group_size, qw, meta, row_interleave)

    def nbytes(self):
        return self.qw.numel() * 4 + self.meta.numel() * 4

    def to_k_major(self):
        """zl_w4m_unpack: (qweight (N,K/8) int32, qzeros (N,K/G) uint8, scales (N,K/G) fp16) -- the operands of
        gptq_gemm_k_major (q_gemm_k_major.cu:957-1116), in checkpoint row order even for a row-interleaved weight."""
        dev = self.qw.device
        ng = self.k // self.group_size
        qw = torch.empty((self.n, self.k // 8), dtype=torch.int32, device=dev)
        qz = torch.empty((self.n, ng), dtype=torch.uint8, device=dev)
        sc = torch.empty((self.n, ng), dtype=torch.float16, device=dev)
        check(lib().zl_w4m_unpack(_p(self.qw), _p(self.meta), _i(self.n), _i(self.k), _i(self.group_size),
                                  C.c_int(int(self.row_interleave)), _p(qw), _p(qz), _p(sc), _stream()), "w4m_unpack")
        return qw, qz, sc

    @classmethod
    def random(cls, n, k, group_size, device, gen=None, scale_mag=0.005, row_interleave=False):
        """Synthetic weight generated directly in the packed layout (benchmarks: no checkpoint I/O)."""
        L = cls.layout(n, k, group_size)
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (L.qw_bytes // 4,), dtype=torch.int32, device=device, generator=gen)
        gi = group_size // 128                          # the (scale, zero) pair is repeated on each 128-k item of its group
        cnt = L.scales_bytes // 4 // gi
        sc = (torch.rand(cnt, device=device, generator=gen) * scale_mag + 1e-4).to(torch.float16)
        sc = sc.view(torch.int16).to(torch.int32) & 0xffff
        z = torch.randint(0, 16, (cnt,), dtype=torch.int32, device=device, generator=gen)
        meta = (sc | ((0xe400 | z) << 16)).to(torch.int64)
        meta = torch.where(meta >= 2 ** 31, meta - 2 ** 32, meta).to(torch.int32)
        if gi > 1:
            meta = meta.view(-1, 1, 16).expand(-1, gi, 16).reshape(-1).contiguous()
        return cls(n, k, group_size, qw, meta, row_interleave)


def decode_attn_split_len(b, num_kv_heads, max_len_buf):
    """keys per split of the split-KV decode attention kernels for this batch geometry (zl_decode_attn_split_len)"""
    return int(lib().zl_decode_attn_split_len(_i(b), _i(num_kv_heads), _i(max_len_buf)))


def _attn_algo():
    """zl_decode_attn_ex algo: ZL_ATTN_MFMA=0 forces the VALU split-KV kernel (tests cover both)"""
    return 1 if os.environ.get("ZL_ATTN_MFMA", "1") == "0" else 0


_SCRATCH = {}
_SCRATCH_RETIRED = []   # outgrown scratch buffers, kept alive for the graphs that captured them
# tuning overrides of the W4A16 launchers: read HERE, on the host side, per call (the C ABI reads no environment); the
# tests and micro-benchmarks sweep them
_W4_ENV = (("phase_rounds", "ZL_W4_PHASE_ROUNDS"), ("phase_ksplit", "ZL_W4_PHASE_KSPLIT"), ("phase_ksplit_min_m", "ZL_W4_PHASE_KSPLIT_MINM"),
           ("phase_min_m", "ZL_W4_PHASE_MIN_M"), ("phase_max_m", "ZL_W4_PHASE_MAX_M"), ("tiled_min_m", "ZL_W4_TILED_MIN_M"),
           ("tiled_bm", "ZL_W4_TILED_BM"), ("tiled_splitk", "ZL_W4_TILED_SPLITK"), ("mfma_ks", "ZL_MFMA_KS"), ("mfma_rounds", "ZL_MFMA_ROUNDS"), ("tiled_wide", "ZL_W4_TILED_WIDE"),
           ("slab", "ZL_W4_SLAB"), ("slab_min_m", "ZL_W4_SLAB_MIN_M"), ("slab_nw", "ZL_W4_SLAB_NW"), ("slab_gpw", "ZL_W4_SLAB_GPW"), ("slab_r", "ZL_W4_SLAB_R"),
           ("defer_norm", "ZL_DEFER_NORM"))


def w4_scratch(device, m, n):
    """The caller-provided scratch of the K-split W4A16 paths (zl_w4_opts_t::scratch): one zero-initialised buffer per
    device, grown on demand OUTSIDE stream capture (run a step eagerly before capturing it, as bench.py does).  Launches
    that use it must be ordered among themselves -- one compute stream per device at a time, like the reference's engine
    (the dual-stream prompt encode only runs collectives on its second stream); a caller with concurrent GEMM streams
    passes its own buffers through zl_w4a16_gemm_mfma_ex."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    need = int(lib().zl_w4a16_scratch_bytes(_i(m), _i(n)))
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            return buf                                 # too small during capture: the launchers take their unsplit routes
        if buf is not None:
            # a buffer that has been handed out is NEVER freed: a hipGraph captured earlier has its address baked into the
            # K-split launches (arrival counters + fp32 partials) and would scribble over whatever the allocator put there next
            _SCRATCH_RETIRED.append(buf)
        buf = torch.zeros(max(need, 64 << 20), dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf


def w4_scratch_reserve(device, m_max, n_max):
    """size the device's K-split scratch once for the largest (rows, columns) any later call will ask for -- before capturing
    graphs, so that no later growth retires the buffer they hold"""
    return w4_scratch(device, m_max, n_max)


def _w4_opts(device, m, n):
    o = W4Opts()
    buf = w4_scratch(device, m, n) if m > 2 else None   # up to 2 rows no launcher splits over workgroups (3..4: the long-K slab route)
    if buf is not None:
        o.scratch, o.scratch_bytes = buf.data_ptr(), buf.numel()
    for field, env in _W4_ENV:
        v = os.environ.get(env)
        if v:
            setattr(o, field, int(v))
    if os.environ.get("ZL_W4_PHASE_SMALL") == "0":
        o.phase_small_off = 1
    if os.environ.get("ZL_W4_SMALL_ALGO"):             # 1: the fp16-dequant kernels of rounds 1-2 for 1..4 rows
        o.small_algo = int(os.environ["ZL_W4_SMALL_ALGO"])
    return o


def row_ss(x, out=None):
    """zl_row_ss: per row and 16-column tile the sum of squares of x (M, K) -> fp32 (M, K / 16): the statistics a normalising
    W4 launch takes through row_ss= instead of walking the rows again (zl_w4_opts_t::row_ss)"""
    _chk_cuda(x, out)
    if x.dtype != torch.float16 or x.dim() != 2 or x.shape[1] % 16:
        raise ZLError("row_ss: fp16 (M, K), K a multiple of 16")
    m, k = x.shape
    if out is None:
        out = torch.empty((m, k // 16), dtype=torch.float32, device=x.device)
    check(lib().zl_row_ss(_p(x), _i(x.stride(0)), _i(m), _i(k), _p(out), _stream()), "row_ss")
    return out


def w4_row_ss_routes(m, w_producers, w_consumers, device):
    """does every producing projection (plain / residual epilogue) leave the rows' statistics and every normalising one take them?
    (the route questions of zl_w4a16_gemm_mfma_ex, asked without launching; consumers: (weight, rope) pairs)"""
    dummy = C.c_void_p(16)
    for w in w_producers:
        o = _w4_opts(device, m, w.n)
        o.row_ss_out = dummy
        if not isinstance(w, W4MWeight) or not lib().zl_w4a16_emits_row_ss(_i(m), _i(w.n), _i(w.k), _i(w.group_size), C.c_int(EPI_RESIDUAL), C.byref(o)):
            return False
    for w, rope in w_consumers:
        o = _w4_opts(device, m, w.n)
        o.row_ss = dummy
        if not isinstance(w, W4MWeight) or not lib().zl_w4a16_takes_row_ss(_i(m), _i(w.n), _i(w.k), _i(w.group_size), C.c_int(int(rope)), C.byref(o)):
            return False
    return True


def w4a16_gemm_mfma(x, w, bias=None, residual=None, out=None, norm_weight=None, norm_eps=1e-5, epilogue=0, row_ss=None, row_ss_out=None):
    """MFMA flavour of w4a16_gemm (fp32 accumulation = the numerics of the reference's M > 40 branch).
    row_ss (M, K / 16) fp32 with norm_weight: the rows' statistics (ops.row_ss or a producing launch's row_ss_out);
    row_ss_out (M, N / 16): filled with the statistics of the rows this launch stores where its route does that (w4_row_ss_routes)."""
    if x.dtype != torch.float16:
        raise ZLError("A must be half")
    _chk_cuda(x, bias, residual, norm_weight)
    x2 = x.reshape(-1, x.shape[-1])
    m, k = x2.shape
    if k != w.k:
        raise ZLError("size K mismatch")
    silu = epilogue & (EPI_SILU_MUL | EPI_SILU_MUL_F32)
    n_out = w.n // 2 if silu else w.n
    if out is None:
        out = torch.empty((m, n_out), dtype=torch.float16, device=x.device)
    else:
        _chk_out(out, m, n_out, torch.float16, x.device, "w4a16_gemm_mfma")
    if bias is not None:
        epilogue |= EPI_BIAS
    opts = _w4_opts(x.device, m, w.n)
    _chk_cuda(row_ss, row_ss_out)
    if row_ss is not None:
        opts.row_ss = row_ss.data_ptr()
    if row_ss_out is not None:
        opts.row_ss_out = row_ss_out.data_ptr()
    check(lib().zl_w4a16_gemm_mfma_ex(_p(x2), _i(x2.stride(0)), _p(w.qw), _p(w.meta), _p(bias), _p(residual),
                                      _p(out), _i(m), _i(w.n), _i(k), _i(w.group_size), _p(norm_weight), _f(norm_eps),
                                      C.c_int(epilogue), C.byref(opts), _stream()), "w4a16_gemm_mfma")
    return out


def w4a16_gemm_tiled(x, w, bias=None, residual=None, out=None, epilogue=0):
    """The M-tiled MFMA GEMM called directly (w4a16_gemm_mfma forwards to it for M > 64)."""
    if x.dtype != torch.float16:
        raise ZLError("A must be half")
    _chk_cuda(x, bias, residual)
    x2 = x.reshape(-1, x.shape[-1])
    m, k = x2.shape
    if k != w.k:
        raise ZLError("size K mismatch")
    silu = epilogue & (EPI_SILU_MUL | EPI_SILU_MUL_F32)
    if out is None:
        out = torch.empty((m, w.n // 2 if silu else w.n), dtype=torch.float16, device=x.device)
    else:
        _chk_out(out, m, w.n // 2 if silu else w.n, torch.float16, x.device, "w4a16_gemm_tiled")
    if bias is not None:
        epilogue |= EPI_BIAS
    opts = _w4_opts(x.device, max(m, 5), w.n)
    check(lib().zl_w4a16_gemm_tiled_ex(_p(x2), _i(x2.stride(0)), _p(w.qw), _p(w.meta), _p(bias), _p(residual), _p(out),
                                       _i(m), _i(w.n), _i(k), _i(w.group_size), C.c_int(epilogue), C.byref(opts), _stream()),
          "w4a16_gemm_tiled")
    return out


def arange_i32(n, device, start=0, step=1):
    """functions::arange (bmengine init.h:10) as a kernel: int32 start, start + step, ..."""
    out = torch.empty(n, dtype=torch.int32, device=device)
    if n:
        check(lib().zl_arange_i32(_p(out), C.c_int32(start), C.c_int32(step), _i(n), _stream()), "arange_i32")
    return out


def divide_i32(a, divisor):
    """functions::divide on int32 (element.h:25): truncating integer quotient"""
    _chk_cuda(a)
    out = torch.empty_like(a)
    if a.numel():
        check(lib().zl_divide_i32(_p(a), _p(out), C.c_int32(int(divisor)), _i(a.numel()), _stream()), "divide_i32")
    return out


def sort_pairs_i32(keys, values, max_key=0):
    """functions::sort_pair_1d (sort.h:8-12): STABLE sort of (int32 key >= 0, int32 value) pairs by key -> (keys, values)"""
    _chk_cuda(keys, values)
    if keys.dtype != torch.int32 or values.dtype != torch.int32 or keys.numel() != values.numel():
        raise ZLError("sort_pairs_i32: two int32 tensors of one length")
    ko, vo = torch.empty_like(keys), torch.empty_like(values)
    n = keys.numel()
    if n:
        ws = torch.empty(2 * n, dtype=torch.int32, device=keys.device)
        check(lib().zl_sort_pairs_i32(_p(keys), _p(values), _p(ko), _p(vo), _p(ws), _i(n), C.c_int32(int(max_key)), _stream()), "sort_pairs_i32")
    return ko, vo


def scatter_update_dim0(dst, dst_index, src, src_index=None):
    """functions::scatter_update_dim0 (scatter.h:7-13): dst[dst_index[i], :] = src[src_index[i] if given else i, :] in place"""
    _chk_cuda(dst, dst_index, src, src_index)
    if dst.dim() != 2 or src.dim() != 2 or dst.shape[1] != src.shape[1] or dst.dtype != src.dtype or not (dst.is_contiguous() and src.is_contiguous()):
        raise ZLError("scatter_update_dim0: dense (X, D) <- (Y, D) of one dtype")
    if dst_index.numel():
        check(lib().zl_scatter_update_dim0(_p(dst), _p(dst_index), _p(src), _p(src_index), _i(dst_index.numel()), _i(src.shape[1] * src.element_size()),
                                           _i(dst.shape[0]), _i(src.shape[0]), _stream()), "scatter_update_dim0")
    return dst


def experimental_build():
    """True when the loaded library is a ZL_BUILD_EXPERIMENTAL=1 build (digit-plane route, engine's fused launch)"""
    lib()
    return _lib.experimental


def _need_experimental(what):
    if not experimental_build():
        raise ZLError(what + ": only in a ZL_BUILD_EXPERIMENTAL=1 build of libzhilight_amd.so (measured, not faster: DESIGN 5.R4)")


def w4_planes_ok(m, k):
    """what the digit-plane route of the W4A16 linears covers: decode batches of 5..32 rows (1..4 rows: the integer-plane GEMV
    converts inside its own prologue), K a multiple of 128 up to 16384"""
    return experimental_build() and 5 <= m <= 32 and k % 128 == 0 and k <= 16384


def w4_planes(x, norm_weight=None, norm_eps=1e-5, out=None):
    """digit planes of an activation matrix (zl_w4a16_planes): x (M <= 32, K) fp16 -> an opaque uint8 buffer for
    w4_linear_planes / w4_qkv_rope_scatter_planes; with norm_weight the RMSNorm of every row is applied first."""
    _need_experimental("w4_planes")
    if x.dtype != torch.float16:
        raise ZLError("A must be half")
    _chk_cuda(x, norm_weight, out)
    x2 = x.reshape(-1, x.shape[-1])
    m, k = x2.shape
    nbytes = int(lib().zl_w4a16_planes_bytes(_i(m), _i(k)))
    if nbytes < 0:
        check(nbytes, "w4a16_planes_bytes")
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    elif out.numel() < nbytes or out.dtype != torch.uint8:
        raise ZLError("w4_planes: out too small")
    check(lib().zl_w4a16_planes(_p(x2), _i(x2.stride(0)), _i(m), _i(k), _p(norm_weight), _f(norm_eps), _p(out), _stream()),
          "w4a16_planes")
    return out


def w4_linear_planes(planes, m, w, bias=None, residual=None, out=None, epilogue=0):
    """w4a16_gemm_mfma on digit planes (zl_w4a16_gemm_planes): planes = w4_planes(x) of an (m, w.k) matrix."""
    _chk_cuda(planes, bias, residual)
    if not isinstance(w, W4MWeight):
        raise ZLError("w4_linear_planes: a ZLW4M weight")
    silu = epilogue & (EPI_SILU_MUL | EPI_SILU_MUL_F32)
    n_out = w.n // 2 if silu else w.n
    if out is None:
        out = torch.empty((m, n_out), dtype=torch.float16, device=planes.device)
    else:
        _chk_out(out, m, n_out, torch.float16, planes.device, "w4_linear_planes")
    if bias is not None:
        epilogue |= EPI_BIAS
    opts = _w4_opts(planes.device, m, w.n)
    check(lib().zl_w4a16_gemm_planes(_p(planes), _p(w.qw), _p(w.meta), _p(bias), _p(residual), _p(out), _i(m), _i(w.n), _i(w.k),
                                     _i(w.group_size), C.c_int(epilogue), C.byref(opts), _stream()), "w4a16_gemm_planes")
    return out


def w4_qkv_rope_scatter_planes(planes, m, w, cos, sin, placement, buf_lens, k_addrs, v_addrs, num_heads, num_kv_heads, dim_head,
                               bias=None, bshd=True, q_out=None):
    """w4_qkv_rope_scatter on digit planes (the norm went into w4_planes)"""
    _chk_cuda(planes, cos, sin, placement, buf_lens, k_addrs, v_addrs, bias)
    if not isinstance(w, W4MWeight):
        raise ZLError("w4_qkv_rope_scatter_planes: a ZLW4M weight")
    if q_out is None:
        q_out = torch.empty((m, num_heads * dim_head), dtype=torch.float16, device=planes.device)
    check(lib().zl_w4a16_qkv_rope_scatter_planes(_p(planes), _p(w.qw), _p(w.meta), _p(bias), _p(cos), _p(sin), _p(placement),
                                                 _p(buf_lens), _p(k_addrs), _p(v_addrs), _p(q_out), _i(m), _i(num_heads),
                                                 _i(num_kv_heads), _i(dim_head), _i(w.k), _i(w.group_size), C.c_int(int(bshd)),
                                                 _stream()), "w4a16_qkv_rope_scatter_planes")
    return q_out


def w4_linear(x, w, **kw):
    """W4A16 linear on whichever packed layout the weight holds (W4Weight: bit-exact warp-reduce
    arithmetic; W4MWeight: fp32-accumulating MFMA arithmetic)."""
    if isinstance(w, W4MWeight):
        return w4a16_gemm_mfma(x, w, **kw)
    return w4a16_gemm(x, w, **kw)


def gemm_nt_small_m(x, weight, bias=None, alpha=1.0, out=None, norm_weight=None, norm_eps=1e-5, argmax_ws=None):
    """y = T(alpha * x . W^T + bias) -- functions::Gemm(trans_b=True) on the decode path (lm_head).
    With argmax_ws (argmax_workspace) the launch also leaves per-wave greedy candidates for greedy_advance."""
    _chk_cuda(x, weight, bias, norm_weight, argmax_ws)
    x2 = x.reshape(-1, x.shape[-1])
    m, k = x2.shape
    n = weight.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    else:
        _chk_out(out, m, n, x.dtype, x.device, "gemm_nt_small_m")
    if argmax_ws is None:
        check(lib().zl_gemm_nt_small_m(_p(x2), _i(x2.stride(0)), _p(weight), _p(bias), _p(out), _i(m), _i(n), _i(k),
                                       _f(alpha), C.c_int(_dt(x)), _p(norm_weight), _f(norm_eps), _stream()),
              "gemm_nt_small_m")
    else:
        check(lib().zl_gemm_nt_small_m_argmax(_p(x2), _i(x2.stride(0)), _p(weight), _p(bias), _p(out), _i(m), _i(n),
                                              _i(k), _f(alpha), C.c_int(_dt(x)), _p(norm_weight), _f(norm_eps),
                                              _p(argmax_ws), _stream()), "gemm_nt_small_m_argmax")
    return out


def gemm_nt(x, weight, bias=None, alpha=1.0, out=None):
    """functions::Gemm(trans_b=True) for any M on the matrix cores (K % 128 == 0); M <= 4 callers want
    gemm_nt_small_m, which streams the weights at HBM speed."""
    _chk_cuda(x, weight, bias)
    x2 = x.reshape(-1, x.shape[-1])
    m, k = x2.shape
    n = weight.shape[0]
    if weight.shape[1] != k:
        raise ZLError("size K mismatch")
    if out is None:
        out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    else:
        _chk_out(out, m, n, x.dtype, x.device, "gemm_nt")
    check(lib().zl_gemm_nt(_p(x2), _i(x2.stride(0)), _p(weight), _p(bias), _p(out), _i(m), _i(n), _i(k), _f(alpha),
                           C.c_int(_dt(x)), _stream()), "gemm_nt")
    return out


class DenseMWeight:
    """A dense (N, K) fp16 / bf16 matrix in the ZLD16M layout (zl_dense_pack_m): 16-row x 128-k tiles whose MFMA fragments are 1 KiB
    contiguous -- for matrices streamed with 5..32 rows (the lm_head of a decode batch).  A second copy of the matrix: the row-major one
    stays what the embedding, the 1..4-row GEMV and the prompt GEMM read."""

    def __init__(self, weight):
        _chk_cuda(weight)
        n, k = weight.shape
        nbytes = int(lib().zl_dense_m_bytes(_i(n), _i(k)))
        if nbytes < 0:
            check(nbytes, "dense_m_bytes")
        self.n, self.k, self.dtype = n, k, weight.dtype
        self.data = torch.empty(nbytes // 2, dtype=weight.dtype, device=weight.device)
        w = weight if weight.is_contiguous() else weight.contiguous()
        check(lib().zl_dense_pack_m(_p(w), _p(self.data), _i(n), _i(k), _stream()), "dense_pack_m")


def gemm_nt_packed(x, w, bias=None, alpha=1.0, out=None):
    """gemm_nt on a DenseMWeight for up to 32 rows: the same bits, the weights streamed in 1 KiB contiguous fragment loads"""
    _chk_cuda(x, bias)
    x2 = x.reshape(-1, x.shape[-1])
    m, k = x2.shape
    if k != w.k or x.dtype != w.dtype:
        raise ZLError("gemm_nt_packed: size K / dtype mismatch")
    if out is None:
        out = torch.empty((m, w.n), dtype=x.dtype, device=x.device)
    else:
        _chk_out(out, m, w.n, x.dtype, x.device, "gemm_nt_packed")
    check(lib().zl_gemm_nt_packed(_p(x2), _i(x2.stride(0)), _p(w.data), _p(bias), _p(out), _i(m), _i(w.n), _i(k), _f(alpha),
                                  C.c_int(_dt(x)), _stream()), "gemm_nt_packed")
    return out


def gemm_nt_f32(x, weight, alpha=1.0):
    """functions::Gemm(trans_b=True) with set_output_type(kFloat): fp32 output, no rounding to T (the MoE router's logits)"""
    _chk_cuda(x, weight)
    x2 = x.reshape(-1, x.shape[-1])
    m, k = x2.shape
    n = weight.shape[0]
    if weight.shape[1] != k or weight.dtype != x.dtype:
        raise ZLError("size K / dtype mismatch")
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    check(lib().zl_gemm_nt_f32(_p(x2), _i(x2.stride(0)), _p(weight), _p(out), _i(m), _i(n), _i(k), _f(alpha), C.c_int(_dt(x)), _stream()), "gemm_nt_f32")
    return out


def argmax_workspace(m, n, device):
    nbytes = lib().zl_argmax_workspace_bytes(_i(m), _i(n))
    if nbytes < 0:
        check(int(nbytes), "argmax_workspace_bytes")
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


def greedy_advance(argmax_ws, m, n, tokens=None, positions=None, placement=None, valid_lens=None, next_tokens=None):
    """Reduce the lm_head's per-wave candidates to the greedy token of each row and advance the decode
    batch's device state (tokens <- pick, the three counters += 1); returns next_tokens (int64) if given."""
    _chk_cuda(argmax_ws, tokens, positions, placement, valid_lens, next_tokens)
    check(lib().zl_greedy_advance(_p(argmax_ws), _i(m), _i(n), _p(tokens), _p(positions), _p(placement),
                                  _p(valid_lens), _p(next_tokens), _stream()), "greedy_advance")
    return next_tokens


def argmax_advance(logits, tokens=None, positions=None, placement=None, valid_lens=None, next_tokens=None):
    """Greedy pick over whole logit rows (first index of the largest value, as torch.argmax) and the batch state's advance in one
    launch: tokens <- pick (int32), next_tokens <- pick (int64), the three counters += 1."""
    _chk_cuda(tokens, positions, placement, valid_lens, next_tokens)
    if not (logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1 and logits.stride(0) >= logits.shape[1]):
        raise ZLError("argmax_advance: (rows, n) CUDA logits with unit column stride")
    rows, n = logits.shape
    code = {torch.float16: 2, torch.bfloat16: 6, torch.float32: 1}[logits.dtype]
    check(lib().zl_argmax_advance(_p(logits), C.c_int(code), _i(rows), _i(n), _i(logits.stride(0)), _p(tokens), _p(positions), _p(placement),
                                  _p(valid_lens), _p(next_tokens), _stream()), "argmax_advance")
    return next_tokens if next_tokens is not None else tokens


# --------------------------------------------------------------------------------------------------
# the batch generator's logit post-processing (src/generator/beam_util.cu, bmengine functions/{softmax,topk}.cu): csrc/sampling_ops.hip
# --------------------------------------------------------------------------------------------------
_ELEM = {torch.float16: 2, torch.bfloat16: 6, torch.float32: 1}


def _rows2d(logits, what):
    if not (logits.is_cuda and logits.dim() == 2 and logits.is_contiguous() and logits.dtype in _ELEM):
        raise ZLError(what + ": contiguous (rows, n) CUDA logits of fp16 / bf16 / fp32")
    return logits.shape


def log_softmax_bias(logits, bias=None, temperature=0.0, out=None):
    """beam_utility::log_softmax_bias: T((x - max) / temperature - log(sum) + bias[row]); temperature 0 = the form without the division"""
    rows, n = _rows2d(logits, "log_softmax_bias")
    out = torch.empty_like(logits) if out is None else out
    check(lib().zl_log_softmax_bias(_p(logits), _p(bias), _p(out), _i(rows), _i(n), _f(temperature), C.c_int(_ELEM[logits.dtype]), _stream()), "log_softmax_bias")
    return out


def softmax_rows(logits, temperature=1.0, out=None):
    """functions::softmax: T(exp(x / t - max / t) / sum)"""
    rows, n = _rows2d(logits, "softmax_rows")
    out = torch.empty_like(logits) if out is None else out
    check(lib().zl_softmax_rows(_p(logits), _p(out), _i(rows), _i(n), _f(temperature), C.c_int(_ELEM[logits.dtype]), _stream()), "softmax_rows")
    return out


def topk_rows(x, top):
    """functions::TopK::forward: (values (rows, top) descending in x's dtype, positions int32); ties go to the lower index"""
    rows, n = _rows2d(x, "topk_rows")
    v = torch.empty((rows, top), dtype=x.dtype, device=x.device)
    i = torch.empty((rows, top), dtype=torch.int32, device=x.device)
    check(lib().zl_topk_rows(_p(x), _p(v), _p(i), _i(rows), _i(n), C.c_int(top), C.c_int(_ELEM[x.dtype]), _stream()), "topk_rows")
    return v, i


def gather_logits(index, logits):
    """beam_utility::gather_logits: float32 values of the flattened logits at int32 positions"""
    _chk_cuda(index, logits)
    out = torch.empty(index.shape, dtype=torch.float32, device=logits.device)
    check(lib().zl_gather_logits(_p(index), _p(logits), _p(out), _i(index.numel()), C.c_int(_ELEM[logits.dtype]), _stream()), "gather_logits")
    return out


def scatter_logits(values, token_ids, batch_ids, logits, add=False):
    """beam_utility::scatter_update, in place: logits[batch_ids[i], token_ids[i]] = T(values[i]) (add: += in T)"""
    rows, n = _rows2d(logits, "scatter_logits")
    _chk_cuda(values, token_ids, batch_ids)
    check(lib().zl_scatter_logits(_p(values), _p(token_ids), _p(batch_ids), _p(logits), _i(token_ids.numel()), _i(n), C.c_int(int(add)),
                                  C.c_int(_ELEM[logits.dtype]), _stream()), "scatter_logits")
    return logits


def repetition_penalty(factor, tokens, batch_ids, logits, presence=None):
    """beam_utility::beam_repetition_penalty, in place: l = presence != 0 ? l - T(presence) : (l < 0 ? l * T(factor) : l / T(factor))"""
    rows, n = _rows2d(logits, "repetition_penalty")
    _chk_cuda(factor, tokens, batch_ids, presence)
    check(lib().zl_repetition_penalty(_p(factor), _p(presence), _p(tokens), _p(batch_ids), _p(logits), _i(tokens.numel()), _i(n),
                                      C.c_int(_ELEM[logits.dtype]), _stream()), "repetition_penalty")
    return logits


# --------------------------------------------------------------------------------------------------
# a17 / a13 / a14 / a18 / a22
# --------------------------------------------------------------------------------------------------
def rmsnorm(x, weight, eps, scale=1.0, x2=None, out=None, out_sum=None):
    """LayerNorm::forward (rms) / fuse_add (src/nn/layernorm/layernorm.cu:227-302)."""
    _chk_cuda(x, weight, x2)
    xr = x.reshape(-1, x.shape[-1])
    rows, dim = xr.shape
    if out is None:
        out = torch.empty_like(xr)
    if x2 is not None and out_sum is None:
        out_sum = torch.empty_like(xr)
    check(lib().zl_rmsnorm(_p(xr), _p(weight), _p(out), _i(rows), _i(dim), _f(eps), _f(scale), _p(x2), _p(out_sum),
                           C.c_int(_dt(x)), _stream()), "rmsnorm")
    return (out, out_sum) if x2 is not None else out


def rope_cos_sin(pos, dim_head, base, neox=True, llama3=None):
    """RopePreparer::compute_cos_sin (src/nn/position/rope_preparer.cu); llama3 = (factor, low, high, old_len)."""
    _chk_cuda(pos)
    s = pos.numel()
    cs = torch.empty((s, dim_head), dtype=torch.float32, device=pos.device)
    sn = torch.empty_like(cs)
    if llama3 is None:
        check(lib().zl_rope_cos_sin(_p(pos), _p(cs), _p(sn), _i(s), _i(dim_head), _f(base), C.c_int(int(neox)),
                                    _stream()), "rope_cos_sin")
    else:
        fac, low, high, old = llama3
        check(lib().zl_rope_cos_sin_llama3(_p(pos), _p(cs), _p(sn), _i(s), _i(dim_head), _f(base), _f(fac), _f(low),
                                           _f(high), _f(old), C.c_int(int(neox)), _stream()), "rope_cos_sin_llama3")
    return cs, sn


def rope_cos_sin_dynamic(pos, dim_head, base, factor, max_position_embeddings, seq_len=None, neox=True):
    """RotaryEmbedding "dynamic" (NTK) angles (src/nn/position/rotary_embedding.cu:19-61); seq_len (s,) int32 = the position
    the reference reads as the row's sequence length (None: the row's own position)."""
    _chk_cuda(pos)
    s = pos.numel()
    cs = torch.empty((s, dim_head), dtype=torch.float32, device=pos.device)
    sn = torch.empty_like(cs)
    if seq_len is not None:
        _chk_cuda(seq_len)
        if seq_len.numel() != s or seq_len.dtype != torch.int32:
            raise ZLError("rope_cos_sin_dynamic: seq_len must be int32 (s,)")
    check(lib().zl_rope_cos_sin_dynamic(_p(pos), _p(seq_len) if seq_len is not None else None, _p(cs), _p(sn), _i(s),
                                        _i(dim_head), _f(base), _f(factor), _f(max_position_embeddings), C.c_int(int(neox)),
                                        _stream()), "rope_cos_sin_dynamic")
    return cs, sn


def yarn_params(base, dim_head, original_max_position, factor, beta_fast=32, beta_slow=1, attn_factor=1.0, deepseek=False,
                mscale=0.0, mscale_all_dim=0.0):
    """YarnImpl's constructor (rotary_embedding.cu:506-553): (low, high, mscale) in double, narrowed to float."""
    def corr(rot):
        return (dim_head * math.log(original_max_position / (rot * 2 * 3.141592653589793))) / (2 * math.log(base))

    def get_mscale(scale, m=1.0):
        return 1.0 if scale <= 1.0 else 0.1 * m * math.log(scale) + 1.0
    f32 = lambda v: C.c_float(v).value  # noqa: E731  (the reference keeps factor / mscale / attn_factor as float members)
    factor, mscale, mscale_all_dim = f32(factor), f32(mscale), f32(mscale_all_dim)
    low = max(float(math.floor(corr(beta_fast))), 0.0)
    high = min(float(math.ceil(corr(beta_slow))), dim_head - 1.0)
    af = f32(attn_factor)
    m = get_mscale(factor) * af
    if deepseek:
        m = get_mscale(factor, mscale) / get_mscale(factor, mscale_all_dim) * af
    return f32(low), f32(high), f32(m)


def rope_cos_sin_yarn(pos, dim_head, base, factor, low, high, mscale, neox=True):
    """YaRN angles (KERNEL_yarn_rope_neox_style, rotary_embedding.cu:398-447): cos / sin already multiplied by mscale."""
    _chk_cuda(pos)
    s = pos.numel()
    cs = torch.empty((s, dim_head), dtype=torch.float32, device=pos.device)
    sn = torch.empty_like(cs)
    check(lib().zl_rope_cos_sin_yarn(_p(pos), _p(cs), _p(sn), _i(s), _i(dim_head), _f(base), _f(factor), _f(low), _f(high),
                                     _f(mscale), C.c_int(int(neox)), _stream()), "rope_cos_sin_yarn")
    return cs, sn


def head_norm(x, weight, num_heads, dim_head, eps, mode=0, out=None):
    """q_norm / k_norm over dim_head: mode 0 = Qwen3's RMSNorm with one (dim_head) weight (attention.cpp:110-113,871-876),
    mode 1 = KERNEL_layernorm_multi_head with a (heads, dim_head) weight (layernorm.cu:329-353).  x (rows, >= heads*dim_head)
    may be a column slice of a wider row (a q or k window of the fused qkv projection); in place with out=x."""
    _chk_cuda(weight)
    if not x.is_cuda or (out is not None and not out.is_cuda):
        raise ZLError("tensors must be CUDA tensors")
    if x.dim() != 2 or x.stride(1) != 1 or x.shape[1] != num_heads * dim_head:
        raise ZLError("head_norm: x must be (rows, heads * dim_head) with unit column stride")
    if weight.numel() != (dim_head if mode == 0 else num_heads * dim_head) or weight.dtype != x.dtype:
        raise ZLError("head_norm: weight shape / dtype mismatch")
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
    elif out.shape != x.shape or out.stride(1) != 1 or out.dtype != x.dtype:
        raise ZLError("head_norm: out shape mismatch")
    check(lib().zl_head_norm(_p(x), _p(weight), _p(out), _i(x.shape[0]), _i(num_heads), _i(dim_head), _i(x.stride(0)),
                             _i(out.stride(0)), _f(eps), C.c_int(mode), _dt(x), _stream()), "head_norm")
    return out


def rotary_embedding_qk(pos, x, num_heads, num_kv_heads, dim_head, rope_theta):
    """nn::rotary_embedding_qk (src/nn/position/rotary_embedding_fuse.cu:70-123)."""
    _chk_cuda(pos, x)
    s = pos.numel()
    q = torch.empty((s, num_heads * dim_head), dtype=x.dtype, device=x.device)
    k = torch.empty((s, num_kv_heads * dim_head), dtype=x.dtype, device=x.device)
    v = torch.empty_like(k)
    check(lib().zl_rotary_embedding_qk(_p(pos), _p(x), _p(q), _p(k), _p(v), _i(s), _i(num_heads), _i(num_kv_heads),
                                       _i(dim_head), _f(rope_theta), C.c_int(_dt(x)), _stream()), "rotary_embedding_qk")
    return q, k, v


def rope_qk_cache(cos, sin, x, num_heads, num_kv_heads, dim_head, neox=True):
    """nn::rope_qk_cache (src/nn/position/rotary_embedding_fuse_cache.cu:65-125)."""
    _chk_cuda(cos, sin, x)
    s = cos.shape[0]
    q = torch.empty((s, num_heads * dim_head), dtype=x.dtype, device=x.device)
    k = torch.empty((s, num_kv_heads * dim_head), dtype=x.dtype, device=x.device)
    v = torch.empty_like(k)
    check(lib().zl_rope_qk_cache(_p(cos), _p(sin), _p(x), _p(q), _p(k), _p(v), _i(s), _i(num_heads), _i(num_kv_heads),
                                 _i(dim_head), C.c_int(int(neox)), C.c_int(_dt(x)), _stream()), "rope_qk_cache")
    return q, k, v


def make_ptr_table(tensors, device=None):
    """Device array of raw pointers, one per task -- RagBufferContext::buf_k_addr
    (src/model/rag_buffer_context.h:141-188)."""
    device = device or tensors[0].device
    return torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64, device=device)


def copy_to_rag_buffer2(placement, buf_lens, k_src, v_src, k_addrs, v_addrs, bshd=True):
    """nn::copy_to_rag_buffer2 (src/kvcache/ragged_buffer_kernel.cu:254-300)."""
    _chk_cuda(placement, buf_lens, k_src, v_src, k_addrs, v_addrs)
    b, len_q, hkv, d = k_src.shape
    if placement.dim() != 2:
        raise ZLError("placement is not 2d")
    check(lib().zl_copy_to_rag_buffer2(_p(placement), _p(buf_lens), _p(k_src), _p(v_src), _p(k_addrs), _p(v_addrs),
                                       _i(b), _i(len_q), _i(hkv), _i(d), C.c_int(int(bshd)), _stream()),
          "copy_to_rag_buffer2")


def rope_scatter_decode(cos, sin, qkv, placement, buf_lens, k_addrs, v_addrs, num_heads, num_kv_heads, dim_head,
                        neox=True, bshd=True, q_out=None):
    """rope_qk_cache + copy_to_rag_buffer2 fused for len_q == 1 decode rows."""
    _chk_cuda(cos, sin, qkv, placement, buf_lens, k_addrs, v_addrs)
    b = qkv.shape[0]
    if q_out is None:
        q_out = torch.empty((b, num_heads * dim_head), dtype=qkv.dtype, device=qkv.device)
    check(lib().zl_rope_scatter_decode(_p(cos), _p(sin), _p(qkv), _p(q_out), _p(placement), _p(buf_lens), _p(k_addrs),
                                       _p(v_addrs), _i(b), _i(num_heads), _i(num_kv_heads), _i(dim_head),
                                       C.c_int(int(neox)), C.c_int(int(bshd)), C.c_int(_dt(qkv)), _stream()),
          "rope_scatter_decode")
    return q_out


def w4_qkv_rope_scatter_ok(m, k, dim_head, norm):
    """what zl_w4a16_qkv_rope_scatter covers (otherwise: w4_linear + rope_scatter_decode)"""
    return m <= 32 and dim_head % 32 == 0 and (not norm or k <= 4096 or m > 8) and (m <= 16 or k <= 8192)


def w4_qkv_rope_scatter(x, w, cos, sin, placement, buf_lens, k_addrs, v_addrs, num_heads, num_kv_heads, dim_head, bias=None,
                        norm_weight=None, norm_eps=1e-5, bshd=True, q_out=None, row_ss=None):
    """fused qkv projection (W4MWeight w, (H + 2 Hkv) D rows) + neox rotary + KV scatter for decode rows; returns the
    rotated q (M, H*D).  Bit-identical to w4_linear(..) followed by rope_scatter_decode(..)."""
    _chk_cuda(x, cos, sin, placement, buf_lens, k_addrs, v_addrs, bias, norm_weight)
    if not isinstance(w, W4MWeight) or x.dtype != torch.float16:
        raise ZLError("w4_qkv_rope_scatter: fp16 activations and a ZLW4M weight")
    m, k = x.shape
    if q_out is None:
        q_out = torch.empty((m, num_heads * dim_head), dtype=x.dtype, device=x.device)
    opts = _w4_opts(x.device, m, w.n)
    if row_ss is not None:
        _chk_cuda(row_ss)
        opts.row_ss = row_ss.data_ptr()
    check(lib().zl_w4a16_qkv_rope_scatter_ex(_p(x), _i(x.stride(0)), _p(w.qw), _p(w.meta), _p(bias), _p(norm_weight),
                                             _f(norm_eps), _p(cos), _p(sin), _p(placement), _p(buf_lens), _p(k_addrs),
                                             _p(v_addrs), _p(q_out), _i(m), _i(num_heads), _i(num_kv_heads), _i(dim_head),
                                             _i(k), _i(w.group_size), C.c_int(int(bshd)), C.byref(opts), _stream()),
          "w4a16_qkv_rope_scatter")
    return q_out


def decode_attn_workspace(b, len_q, h, d, max_len_buf, device):
    nbytes = lib().zl_decode_attn_workspace_bytes(_i(b), _i(len_q), _i(h), _i(d), _i(max_len_buf))
    if nbytes < 0:
        check(int(nbytes), "decode_attn_workspace_bytes")
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


def multi_query_attention_rag_buffer(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, mask, scale, max_len_buf,
                                     num_kv_heads, valid_lens=None, bshd=True, out=None, workspace=None):
    """nn::multi_query_attention_rag_buffer / attention_qkv_rag_buffer
    (src/nn/attention/attention_kernel.cu:1252-1457, 1150-1213).  batch_q (B, len_q, H, D)."""
    _chk_cuda(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, mask, valid_lens)
    b, len_q, h, d = batch_q.shape
    if out is None:
        out = torch.empty_like(batch_q)
    if workspace is None:
        workspace = decode_attn_workspace(b, len_q, h, d, max_len_buf, batch_q.device)
    check(lib().zl_decode_attn_ex(_p(batch_q), _p(buf_lens), _p(key_buf_addrs), _p(val_buf_addrs), _p(mask),
                               _p(valid_lens), _p(out), _p(workspace), _i(b), _i(len_q), _i(h), _i(num_kv_heads),
                               _i(d), _f(scale), _i(max_len_buf), C.c_int(int(bshd)), C.c_int(_dt(batch_q)),
                               C.c_int(_attn_algo()), _stream()), "decode_attn")
    return out


def decode_attn_la_split_len(b, num_kv_heads, max_len_buf):
    """keys per split zl_decode_attn_la picks for this batch geometry (a multiple of 32)"""
    return int(lib().zl_decode_attn_la_split_len(_i(b), _i(num_kv_heads), _i(max_len_buf)))


def decode_attn_la_workspace(b, h, num_kv_heads, max_len_buf, device, split_len=0):
    """zero-initialised workspace of decode_attention_la (arrival words + split records); split_len = 0: large enough for any
    split length.  One per stream of launches that may overlap."""
    nbytes = lib().zl_decode_attn_la_workspace_bytes(_i(b), _i(h), _i(num_kv_heads), _i(max_len_buf), _i(split_len))
    if nbytes < 0:
        check(int(nbytes), "decode_attn_la_workspace_bytes")
    return torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=device)


def decode_attention_la(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, valid_lens, scale, max_len_buf, num_kv_heads, workspace,
                        out=None, bshd=True, split_len=0, half=False):
    """multi_query_attention_rag_buffer for decode rows (len_q = 1, prefix visibility) as ONE launch: the split merge is done by
    the last-arriving workgroup of every (task, kv head) pair (zl_decode_attn_la).  batch_q (B, 1, H, 128) or (B, H, 128);
    workspace from decode_attn_la_workspace (zeroed once).  Returns out, shaped like batch_q."""
    _chk_cuda(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, valid_lens, workspace)
    b, h, d = batch_q.shape[0], batch_q.shape[-2], batch_q.shape[-1]
    if batch_q.dim() == 4 and batch_q.shape[1] != 1:
        raise ZLError("decode_attention_la: one query row per task")
    if out is None:
        out = torch.empty_like(batch_q)
    check(lib().zl_decode_attn_la(_p(batch_q), _p(buf_lens), _p(key_buf_addrs), _p(val_buf_addrs), _p(valid_lens), _p(out),
                                  _p(workspace), _i(b), _i(h), _i(num_kv_heads), _i(d), _f(scale), _i(max_len_buf),
                                  C.c_int(int(bshd)), C.c_int(_dt(batch_q)), _i(split_len),
                                  C.c_int(int(bool(half)) | (int(os.environ.get("ZL_ATTN_LA_WAVES", "0") or 0) << 8)), _stream()),
          "decode_attn_la")
    return out


def attn_merge_plan(b, num_heads, num_kv_heads, dim_head, max_len_buf, w, dtype=torch.float16):
    """(split_len, max_splits) when zl_decode_attn_splits + zl_w4a16_gemm_attn_merge cover this decode batch and
    projection weight, else None (callers use multi_query_attention_rag_buffer + w4_linear).  ZL_ATTN_MERGE_MAX_B
    (default 1) bounds the batch: every workgroup of the projection merges all rows, which stops paying early."""
    if not isinstance(w, W4MWeight) or dim_head != 128 or num_heads % num_kv_heads or num_heads // num_kv_heads > 16:
        return None
    if b > min(4, int(os.environ.get("ZL_ATTN_MERGE_MAX_B", "1"))) or w.k != num_heads * 128 or w.k > 4096:
        return None
    split_len = int(lib().zl_decode_attn_split_len(_i(b), _i(num_kv_heads), _i(max_len_buf)))
    max_splits = (max_len_buf + split_len - 1) // split_len
    cus = lib().zl_device_cu_count()
    if max_splits > 16 or (w.n + 15) // 16 > 2 * (cus if cus > 0 else 256):
        return None
    return split_len, max_splits, _merge_half_dtype(dtype)     # [2]: fp16 partial records (decided ONCE, for both launches)


def _merge_half_dtype(dtype):
    return _merge_half() and dtype == torch.float16


def _merge_half(t=None):
    """half-precision split partials + the integer-plane merging projection (zl_decode_attn_splits_h /
    zl_w4a16_gemm_attn_merge_h) unless the round-2 kernels are asked for (ZL_W4_SMALL_ALGO=1) or the rows are not fp16"""
    return os.environ.get("ZL_W4_SMALL_ALGO", "0") != "1" and os.environ.get("ZL_ATTN_MERGE_HALF", "1") != "0" and \
        (t is None or t.dtype == torch.float16)


def decode_attention_splits(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, valid_lens, scale, max_len_buf, num_kv_heads,
                            workspace, bshd=True):
    """multi_query_attention_rag_buffer without its merge launch: the split-KV partials stay in `workspace` for
    w4_attn_out_merge.  batch_q (B, 1, H, 128) or (B, H, 128)."""
    _chk_cuda(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, valid_lens, workspace)
    b, h, d = batch_q.shape[0], batch_q.shape[-2], batch_q.shape[-1]
    if _merge_half(batch_q):
        check(lib().zl_decode_attn_splits_h(_p(batch_q), _p(buf_lens), _p(key_buf_addrs), _p(val_buf_addrs), _p(valid_lens),
                                            _p(workspace), _i(b), _i(h), _i(num_kv_heads), _i(d), _f(scale), _i(max_len_buf),
                                            C.c_int(int(bshd)), _stream()), "decode_attn_splits_h")
        return workspace
    check(lib().zl_decode_attn_splits(_p(batch_q), _p(buf_lens), _p(key_buf_addrs), _p(val_buf_addrs), _p(valid_lens),
                                      _p(workspace), _i(b), _i(h), _i(num_kv_heads), _i(d), _f(scale), _i(max_len_buf),
                                      C.c_int(int(bshd)), C.c_int(_dt(batch_q)), _stream()), "decode_attn_splits")
    return workspace


def decode_attention_splits_mask(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, mask, scale, max_len_buf, num_kv_heads, workspace,
                                 bshd=True):
    """decode_attention_splits with the reference's int8 visibility mask (one row of buf_lens[b] entries per task) instead of prefix
    lengths (zl_decode_attn_splits_h_mask): half-precision records of EVERY split of the buffer; the merging projection
    (w4_attn_out_merge) and decode_attention_combine_h take them with valid_lens = buf_lens.  fp16 only."""
    _chk_cuda(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, mask, workspace)
    b, h, d = batch_q.shape[0], batch_q.shape[-2], batch_q.shape[-1]
    if batch_q.dtype != torch.float16:
        raise ZLError("decode_attention_splits_mask: fp16 rows")
    check(lib().zl_decode_attn_splits_h_mask(_p(batch_q), _p(buf_lens), _p(key_buf_addrs), _p(val_buf_addrs), _p(mask), _p(workspace),
                                             _i(b), _i(h), _i(num_kv_heads), _i(d), _f(scale), _i(max_len_buf), C.c_int(int(bshd)),
                                             _stream()), "decode_attn_splits_h_mask")


def decode_attention_combine_h(workspace, buf_lens, valid_lens, b, h, num_kv_heads, max_len_buf, out=None):
    """the merge of half-precision split records as a launch of its own (zl_decode_attn_combine_h): bit for bit the rows
    w4_attn_out_merge's prologue multiplies.  valid_lens None: every split of the buffer (the mask form's records)."""
    _chk_cuda(workspace, buf_lens, valid_lens)
    if out is None:
        out = torch.empty((b, 1, h, 128), dtype=torch.float16, device=workspace.device)
    check(lib().zl_decode_attn_combine_h(_p(workspace), _p(buf_lens), _p(valid_lens), _p(out), _i(b), _i(h), _i(num_kv_heads),
                                         _i(max_len_buf), _stream()), "decode_attn_combine_h")
    return out


def w4_attn_out_merge(workspace, buf_lens, valid_lens, plan, b, w, bias=None, residual=None, out=None, epilogue=0):
    """attn_out projection whose activation rows are merged from the decode attention's split-KV partials in the GEMV
    prologue (zl_w4a16_gemm_attn_merge): bit-identical to the merge launch + w4a16_gemm_mfma."""
    _chk_cuda(workspace, buf_lens, valid_lens, bias, residual)
    split_len, max_splits = plan[0], plan[1]
    if out is None:
        out = torch.empty((b, w.n), dtype=torch.float16, device=workspace.device)
    if bias is not None:
        epilogue |= EPI_BIAS
    # the partial layout was decided by decode_attention_splits from the query dtype; the plan carries that decision
    # (a bare _merge_half() here read fp16 records out of an fp32 workspace for non-fp16 queries)
    half = plan[2] if len(plan) > 2 else _merge_half()
    if half:
        opts = _w4_opts(workspace.device, b, w.n)
        check(lib().zl_w4a16_gemm_attn_merge_h_ex(_p(workspace), _p(buf_lens), _p(valid_lens), _i(split_len), _i(max_splits),
                                                  _p(w.qw), _p(w.meta), _p(bias), _p(residual), _p(out), _i(b), _i(w.n), _i(w.k),
                                                  _i(w.group_size), C.c_int(epilogue), C.byref(opts), _stream()),
              "w4a16_gemm_attn_merge_h")
        return out
    check(lib().zl_w4a16_gemm_attn_merge(_p(workspace), _p(buf_lens), _p(valid_lens), _i(split_len), _i(max_splits), _p(w.qw),
                                         _p(w.meta), _p(bias), _p(residual), _p(out), _i(b), _i(w.n), _i(w.k),
                                         _i(w.group_size), C.c_int(epilogue), _stream()), "w4a16_gemm_attn_merge")
    return out


_ENGINE_STATE = {}


def engine_state(device):
    """granules + epoch + error words of the fused decode launches (zl_w4a16_attn_out_gate_up): one set per device, shared by
    every layer (the epoch differs per layer and per step)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _ENGINE_STATE.get(key)
    if st is None:
        st = dict(granules=torch.zeros(4 * 4096 * 4, dtype=torch.uint8, device=device),
                  epoch=torch.ones(1, dtype=torch.int32, device=device), err=torch.zeros(1, dtype=torch.int32, device=device))
        _ENGINE_STATE[key] = st
    return st


def engine_epoch_advance(device, by=256):
    """once per decode step, ahead of the step's fused launches"""
    _need_experimental("engine_epoch_advance")
    st = engine_state(device)
    check(lib().zl_engine_epoch_advance(_p(st["epoch"]), C.c_uint32(by), _stream()), "engine_epoch_advance")


def w4_attn_out_gate_up(workspace, buf_lens, valid_lens, plan, b, w_o, hidden, w_ff, norm_weight, norm_eps, act, layer_index,
                        bias_o=None, bias_ff=None):
    """zl_w4a16_attn_out_gate_up: hidden += attn_out(merge(partials)); act = silu(w_in(ln(hidden))) * w_gated(ln(hidden)) in one
    launch.  Returns False (nothing launched) outside the launcher's range."""
    _chk_cuda(workspace, buf_lens, valid_lens, hidden, norm_weight, act, bias_o, bias_ff)
    split_len, max_splits = plan[0], plan[1]
    if (len(plan) > 2 and not plan[2]) or not experimental_build():
        return False
    st = engine_state(hidden.device)
    rc = lib().zl_w4a16_attn_out_gate_up(_p(workspace), _p(buf_lens), _p(valid_lens), _i(split_len), _i(max_splits), _p(w_o.qw),
                                         _p(w_o.meta), _p(bias_o), _p(hidden), _p(w_ff.qw), _p(w_ff.meta), _p(bias_ff),
                                         _p(norm_weight), _f(norm_eps), _p(act), _i(b), _i(w_o.n), _i(w_o.k), _i(w_ff.n),
                                         _i(w_o.group_size), _p(st["granules"]), _p(st["epoch"]), C.c_uint32(layer_index),
                                         _p(st["err"]), _stream())
    if rc in (-2, -4):                                    # ZL_ESHAPE / ZL_ELIMIT: the caller takes the two launches
        return False
    check(rc, "w4a16_attn_out_gate_up")
    return True


def decode_attention_fused(cos, sin, qkv, placement, buf_lens, valid_lens, k_addrs, v_addrs, num_heads, num_kv_heads,
                           dim_head, scale, max_len_buf, neox=True, bshd=True, out=None, workspace=None):
    """rope_qk_cache + copy_to_rag_buffer2 + multi_query_attention_rag_buffer for len_q == 1 decode rows in
    one launch pair (partial + combine); returns the attention output (B, H*D)."""
    _chk_cuda(cos, sin, qkv, placement, buf_lens, valid_lens, k_addrs, v_addrs)
    b = qkv.shape[0]
    if out is None:
        out = torch.empty((b, num_heads * dim_head), dtype=qkv.dtype, device=qkv.device)
    if workspace is None:
        workspace = decode_attn_workspace(b, 1, num_heads, dim_head, max_len_buf, qkv.device)
    check(lib().zl_decode_attn_fused(_p(cos), _p(sin), _p(qkv), _p(placement), _p(buf_lens), _p(valid_lens), _p(k_addrs),
                                     _p(v_addrs), _p(out), _p(workspace), _i(b), _i(num_heads), _i(num_kv_heads),
                                     _i(dim_head), _f(scale), _i(max_len_buf), C.c_int(int(neox)), C.c_int(int(bshd)),
                                     C.c_int(_dt(qkv)), _stream()), "decode_attn_fused")
    return out


def quant_calc_scale_zp(x, q_zero=128):
    """int8_op::quant_calc_scale(ctx, x, 127, q_zero) (src/nn/quant/int8/quant_kernel.cu:49-80): the INT8 KV cache's
    row quantiser -> (u8 codes (M, K), fp32 scale (M))."""
    _chk_cuda(x)
    xr = x.reshape(-1, x.shape[-1])
    m, k = xr.shape
    q = torch.empty((m, k), dtype=torch.uint8, device=x.device)
    s = torch.empty((m,), dtype=torch.float32, device=x.device)
    check(lib().zl_quant_calc_scale_zp(_p(xr), _p(q), _p(s), _i(m), _i(k), C.c_int(q_zero), C.c_int(_dt(x)), _stream()),
          "quant_calc_scale_zp")
    return q, s


def quant_copy_to_rag_buffer(placement, buf_lens, k_src, v_src, k_addrs, v_addrs, k_scale_addrs, v_scale_addrs, len_q=1,
                             bshd=True):
    """quant_calc_scale(127, 128) + copy_to_rag_buffer2 for the codes and the scales (attention.cpp:656-676).
    k_src / v_src (B*len_q, Hkv, D)."""
    _chk_cuda(placement, buf_lens, k_src, v_src, k_addrs, v_addrs, k_scale_addrs, v_scale_addrs)
    tokens, hkv, d = k_src.shape
    check(lib().zl_quant_copy_to_rag_buffer(_p(placement), _p(buf_lens), _p(k_src), _p(v_src), _p(k_addrs), _p(v_addrs),
                                            _p(k_scale_addrs), _p(v_scale_addrs), _i(tokens // len_q), _i(len_q), _i(hkv),
                                            _i(d), C.c_int(int(bshd)), C.c_int(_dt(k_src)), _stream()),
          "quant_copy_to_rag_buffer")


def rope_quant_scatter_decode(cos, sin, qkv, placement, buf_lens, k_addrs, v_addrs, k_scale_addrs, v_scale_addrs,
                              num_heads, num_kv_heads, dim_head, neox=True, bshd=True, q_out=None):
    """rope_qk_cache + quantise + scatter of the new K/V rows for len_q == 1 decode rows; returns the rotated q."""
    _chk_cuda(cos, sin, qkv, placement, buf_lens, k_addrs, v_addrs, k_scale_addrs, v_scale_addrs)
    b = qkv.shape[0]
    if q_out is None:
        q_out = torch.empty((b, num_heads * dim_head), dtype=qkv.dtype, device=qkv.device)
    check(lib().zl_rope_quant_scatter_decode(_p(cos), _p(sin), _p(qkv), _p(q_out), _p(placement), _p(buf_lens),
                                             _p(k_addrs), _p(v_addrs), _p(k_scale_addrs), _p(v_scale_addrs), _i(b),
                                             _i(num_heads), _i(num_kv_heads), _i(dim_head), C.c_int(int(neox)),
                                             C.c_int(int(bshd)), C.c_int(_dt(qkv)), _stream()),
          "rope_quant_scatter_decode")
    return q_out


def multi_query_attention_rag_buffer_quant(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, scale_k_addrs, scale_v_addrs,
                                           mask, scale, max_len_buf, num_kv_heads, valid_lens=None, bshd=True, out=None,
                                           workspace=None):
    """nn::multi_query_attention_rag_buffer with scale_key_addrs / scale_val_addrs (INT8 KV cache;
    src/nn/attention/attention_kernel.cu:1384-1416).  batch_q (B, len_q, H, D)."""
    _chk_cuda(batch_q, buf_lens, key_buf_addrs, val_buf_addrs, scale_k_addrs, scale_v_addrs, mask, valid_lens)
    b, len_q, h, d = batch_q.shape
    if out is None:
        out = torch.empty_like(batch_q)
    if workspace is None:
        workspace = decode_attn_workspace(b, len_q, h, d, max_len_buf, batch_q.device)
    check(lib().zl_decode_attn_quant_ex(_p(batch_q), _p(buf_lens), _p(key_buf_addrs), _p(val_buf_addrs), _p(scale_k_addrs),
                                     _p(scale_v_addrs), _p(mask), _p(valid_lens), _p(out), _p(workspace), _i(b),
                                     _i(len_q), _i(h), _i(num_kv_heads), _i(d), _f(scale), _i(max_len_buf),
                                     C.c_int(int(bshd)), C.c_int(_dt(batch_q)), C.c_int(_attn_algo()), _stream()), "decode_attn_quant")
    return out


def prefill_attention(q, k_buf, v_buf, pos0, num_kv_heads, scale, bshd=True, out=None, groups=None):
    """Causal attention of one task's prompt chunk (attn_encode_group -> flash attention in the reference,
    src/nn/attention/attention.cpp:442-622).  q (s_q, H, D); k_buf / v_buf the task's buffers (len_buf, Hkv, D)
    [bshd] or (Hkv, len_buf, D), already holding the chunk's rows at pos0 .. pos0 + s_q - 1."""
    _chk_cuda(q, k_buf, v_buf)
    if q.dtype not in (torch.float16, torch.bfloat16):
        raise ZLError("prefill_attention: fp16 / bf16 only")
    s_q, h, d = q.shape
    len_buf = k_buf.shape[0] if bshd else k_buf.shape[1]
    if out is None:
        out = torch.empty_like(q)
    groups = int(os.environ.get("ZL_PREFILL_GROUPS", "0") or 0) if groups is None else groups     # 0: the launcher's choice
    check(lib().zl_prefill_attn_ex(_p(q), _p(k_buf), _p(v_buf), _p(out), _i(s_q), _i(pos0), _i(h), _i(num_kv_heads), _i(d),
                                   _f(scale), _i(len_buf), C.c_int(int(bshd)), C.c_int(_dt(q)), C.c_int(groups), _stream()), "prefill_attn")
    return out


def element_add_scale(a, b, scale=1.0, scale_residual=True, out=None):
    """nn::element_add_scale_out (src/nn/block/block_kernel.cu:19-50)."""
    _chk_cuda(a, b)
    if a.shape != b.shape:
        raise ZLError("a,b shape mismatch")
    if out is None:
        out = torch.empty_like(a)
    check(lib().zl_element_add_scale(_p(a), _p(b), _p(out), _i(a.numel()), _f(scale), C.c_int(int(scale_residual)),
                                     C.c_int(_dt(a)), _stream()), "element_add_scale")
    return out


def permute_input(x, perm, out=None):
    """nn::gptq::permute_input (src/nn/quant/gptq/gptq.h:155-159): out[..., i] = x[..., perm[i]] (perm int32, any length)."""
    _chk_cuda(x, perm)
    if perm.dtype != torch.int32:
        raise ZLError("perm dtype mismatch")
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    k = perm.numel()
    if out is None:
        out = torch.empty((x2.shape[0], k), dtype=x.dtype, device=x.device)
    check(lib().zl_permute_input(_p(x2), _i(x2.stride(0)), _p(perm), _p(out), _i(x2.shape[0]), _i(k), _stream()), "permute_input")
    return out


def gate_mul(inp, in2, gate_type="silu", out=None):
    """nn::gate_mul_inplace (src/nn/linear/activation_kernel.cu:82-106); out defaults to in place."""
    _chk_cuda(inp, in2)
    if gate_type not in ("silu", "gelu"):
        raise ZLError("Unsupported gate type: " + gate_type)
    out = inp if out is None else out
    check(lib().zl_gate_mul(_p(inp), _p(in2), _p(out), _i(inp.numel()), C.c_int(0 if gate_type == "silu" else 1),
                            C.c_int(_dt(inp)), _stream()), "gate_mul")
    return out


def embedding(ids, weight, scale=1.0, begin=0, end=None):
    """RawEmbedding::forward (src/nn/embedding/embedding.cu:260-272)."""
    _chk_cuda(ids, weight)
    if ids.dtype != torch.int32:
        raise ZLError("ids dtype mismatch")
    end = begin + weight.shape[0] if end is None else end
    out = torch.empty(tuple(ids.shape) + (weight.shape[1],), dtype=weight.dtype, device=weight.device)
    check(lib().zl_embedding(_p(ids), _p(weight), _p(out), _i(ids.numel()), _i(weight.shape[1]), C.c_int32(begin),
                             C.c_int32(end), _f(scale), C.c_int(_dt(weight)), _stream()), "embedding")
    return out


def embedding_rope(ids, weight, scale, pos, dim_head, base, neox=True, llama3=None):
    """embedding + rope_cos_sin in one launch (the first two kernels of a decode step): (hidden, cos, sin), bit-identical to the
    two calls"""
    _chk_cuda(ids, weight, pos)
    if ids.dtype != torch.int32 or pos.numel() != ids.numel():
        raise ZLError("ids dtype mismatch / one position per token")
    s = ids.numel()
    out = torch.empty(tuple(ids.shape) + (weight.shape[1],), dtype=weight.dtype, device=weight.device)
    cs = torch.empty((s, dim_head), dtype=torch.float32, device=pos.device)
    sn = torch.empty_like(cs)
    fac, low, high, old = llama3 if llama3 is not None else (1.0, 1.0, 1.0, 1.0)
    check(lib().zl_embedding_rope(_p(ids), _p(weight), _p(out), _i(s), _i(weight.shape[1]), C.c_int32(0), C.c_int32(weight.shape[0]), _f(scale),
                                  C.c_int(_dt(weight)), _p(pos), _p(cs), _p(sn), _i(dim_head), _f(base), C.c_int(int(neox)),
                                  C.c_int(int(llama3 is not None)), _f(fac), _f(low), _f(high), _f(old), _stream()), "embedding_rope")
    return out, cs, sn


# --------------------------------------------------------------------------------------------------
# a8..a11: INT8
# --------------------------------------------------------------------------------------------------
W8_BACK, W8_BACK_ADD, W8_ACT_SILU, W8_ACT_GELU = 0, 1, 2, 3


class W8MWeight:
    """int8 weight (N, K) in the ZLW8M streaming layout of zl_w8a8_gemm_phase + its per-row scale (T, interleaved with
    the rows when row_interleave)."""

    def __init__(self, n, k, qw, scale, row_interleave=False):
        self.n, self.k, self.qw, self.scale, self.row_interleave = n, k, qw, scale, row_interleave

    @classmethod
    def from_rows(cls, w_int8, scale, row_interleave=False):
        _chk_cuda(w_int8, scale)
        n, k = w_int8.shape
        nbytes = lib().zl_w8m_bytes(_i(n), _i(k))
        if nbytes < 0:
            check(int(nbytes), "w8m_bytes")
        qw = torch.empty(nbytes, dtype=torch.uint8, device=w_int8.device)
        check(lib().zl_w8m_pack(_p(w_int8.contiguous()), _i(n), _i(k), C.c_int(int(row_interleave)), _p(qw), _stream()), "w8m_pack")
        if row_interleave:
            scale = torch.stack([scale[:n // 2], scale[n // 2:]], dim=1).reshape(-1)
        return cls(n, k, qw, scale.contiguous(), row_interleave)

    def nbytes(self):
        return self.qw.numel() + self.scale.numel() * 2


def w8a8_gemm_phase(xq, sx, w: W8MWeight, epilogue=W8_BACK, addend=None, scale=1.0, out=None, dtype=torch.float16):
    """Int8Linear GEMM + scale-back in one launch for 1..32 rows (zl_w8a8_gemm_phase)."""
    _chk_cuda(xq, sx, addend, out)
    m, k = xq.shape
    if k != w.k:
        raise ZLError("w8a8_gemm_phase: K mismatch")
    gated = epilogue in (W8_ACT_SILU, W8_ACT_GELU)
    if gated != w.row_interleave:
        raise ZLError("w8a8_gemm_phase: gated epilogues need a row-interleaved weight (and only they do)")
    if out is None:
        out = torch.empty((m, w.n // 2 if gated else w.n), dtype=dtype, device=xq.device)
    rounds = int(os.environ.get("ZL_W8_PHASE_ROUNDS", "0") or 0)     # tests sweep the instantiations
    check(lib().zl_w8a8_gemm_phase_ex(_p(xq), _p(sx), _p(w.qw), _p(w.scale), _p(addend), _p(out), _i(m), _i(w.n), _i(k), _f(scale),
                                      C.c_int(epilogue), C.c_int(_dt(out)), C.c_int(rounds), _stream()), "w8a8_gemm_phase")
    return out


def w8a8_qkv_rope_scatter(xq, sx, w: W8MWeight, cos, sin, placement, buf_lens, k_addrs, v_addrs, num_heads, num_kv_heads,
                          dim_head, bshd=True, q_out=None, dtype=torch.float16):
    """INT8 fused qkv projection + scale-back + neox rotary + KV scatter for decode rows; returns the rotated q (M, H*D)."""
    _chk_cuda(xq, sx, cos, sin, placement, buf_lens, k_addrs, v_addrs)
    m, k = xq.shape
    if k != w.k or w.n != (num_heads + 2 * num_kv_heads) * dim_head or w.row_interleave:
        raise ZLError("w8a8_qkv_rope_scatter: weight shape")
    if q_out is None:
        q_out = torch.empty((m, num_heads * dim_head), dtype=dtype, device=xq.device)
    check(lib().zl_w8a8_qkv_rope_scatter(_p(xq), _p(sx), _p(w.qw), _p(w.scale), _p(cos), _p(sin), _p(placement), _p(buf_lens),
                                         _p(k_addrs), _p(v_addrs), _p(q_out), _i(m), _i(num_heads), _i(num_kv_heads),
                                         _i(dim_head), _i(k), C.c_int(int(bshd)), C.c_int(_dt(q_out)), _stream()),
          "w8a8_qkv_rope_scatter")
    return q_out


def quant_calc_scale(x):
    """int8_op::quant_calc_scale (src/nn/quant/int8/quant_kernel.cu:49-103) -> (int8 (M,K), fp32 scale (M))."""
    _chk_cuda(x)
    xr = x.reshape(-1, x.shape[-1])
    m, k = xr.shape
    q = torch.empty((m, k), dtype=torch.int8, device=x.device)
    s = torch.empty((m,), dtype=torch.float32, device=x.device)
    check(lib().zl_quant_calc_scale(_p(xr), _p(q), _p(s), _i(m), _i(k), C.c_int(_dt(x)), _stream()), "quant_calc_scale")
    return q, s


def layernorm_quant(x, weight, eps, scale=1.0):
    """int8_op::layernorm_quant (src/nn/quant/int8/quant_kernel.cu:153-227) -> (T out, int8, fp32 scale)."""
    _chk_cuda(x, weight)
    xr = x.reshape(-1, x.shape[-1])
    rows, dim = xr.shape
    out = torch.empty_like(xr)
    q = torch.empty((rows, dim), dtype=torch.int8, device=x.device)
    s = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(lib().zl_rmsnorm_quant(_p(xr), _p(weight), _p(out), _p(q), _p(s), _i(rows), _i(dim), _f(eps), _f(scale),
                                 C.c_int(_dt(x)), _stream()), "layernorm_quant")
    return out, q, s


def int8_gemm_nt(a, b, out=None):
    """int8 (M,K) x int8 (N,K)^T -> int32 (M,N): the IMMA GEMM of Int8Linear::forward."""
    _chk_cuda(a, b)
    m, k = a.shape
    n = b.shape[0]
    c = out if out is not None else torch.empty((m, n), dtype=torch.int32, device=a.device)
    check(lib().zl_int8_gemm_nt(_p(a), _p(b), _p(c), _i(m), _i(n), _i(k), _stream()), "int8_gemm_nt")
    return c


def quant_scale_back(c, scale_x, scale_y, dtype=torch.float16, out=None):
    """int8_op::quant_scale_back (src/nn/quant/int8/quant_kernel.cu:248-306)."""
    _chk_cuda(c, scale_x, scale_y)
    m, n = c.shape
    if out is None:
        out = torch.empty((m, n), dtype=dtype, device=c.device)
    check(lib().zl_quant_scale_back(_p(c), _p(scale_x), _p(scale_y), _p(out), _i(m), _i(n), C.c_int(_dt(out)),
                                    _stream()), "quant_scale_back")
    return out


def quant_back_act_mul(a, a_sx, a_sy, b, b_sx, b_sy, act="silu", dtype=torch.float16):
    """int8_op::quant_back_act_mul (src/nn/quant/int8/quant_kernel.cu:616-676)."""
    _chk_cuda(a, a_sx, a_sy, b, b_sx, b_sy)
    m, n = a.shape
    out = torch.empty((m, n), dtype=dtype, device=a.device)
    check(lib().zl_quant_back_act_mul(_p(a), _p(a_sx), _p(a_sy), _p(b), _p(b_sx), _p(b_sy), _p(out), _i(m), _i(n),
                                      C.c_int(0 if act == "silu" else 1), C.c_int(_dt(out)), _stream()),
          "quant_back_act_mul")
    return out


def quant_scale_back3(c, scale_x, scale_y, dim_q, dim_kv):
    """int8_op::quant_scale_back3 (src/nn/quant/int8/quant_kernel.cu:311-384): fused qkv int32 -> q, k, v."""
    _chk_cuda(c, scale_x, scale_y)
    m, n = c.shape
    if dim_q + 2 * dim_kv != n:
        raise ZLError("Wrong size")
    mk = lambda w: torch.empty((m, w), dtype=scale_y.dtype, device=c.device)  # noqa: E731
    q, k, v = mk(dim_q), mk(dim_kv), mk(dim_kv)
    check(lib().zl_quant_scale_back3(_p(c), _p(scale_x), _p(scale_y), _p(q), _p(k), _p(v), _i(m), _i(n), _i(dim_q),
                                     _i(dim_kv), C.c_int(_dt(q)), _stream()), "quant_scale_back3")
    return q, k, v


def quant_back_element_add_scale(a, scale_x, scale_y, b, scale=1.0, out=None):
    """int8_op::quant_back_element_add_scale (quant_kernel.cu:530-583): T((back(a) + float(b)) * scale);
    out may alias b (element-wise)."""
    _chk_cuda(a, scale_x, scale_y, b)
    m, n = a.shape
    if out is None:
        out = torch.empty((m, n), dtype=b.dtype, device=a.device)
    check(lib().zl_quant_back_element_add_scale(_p(a), _p(scale_x), _p(scale_y), _p(b), _f(scale), _p(out), _i(m), _i(n),
                                                C.c_int(_dt(out)), _stream()), "quant_back_element_add_scale")
    return out


def quant_back_transpose(h_q, scale_x, scale_y):
    """int8_op::quant_back_transpose (quant_kernel.cu:475-527): (B, len_q, H, D) int32 -> (B, H, len_q, D)."""
    _chk_cuda(h_q, scale_x, scale_y)
    if h_q.dim() != 4:
        raise ZLError("input is not 4d")
    b, t, h, d = h_q.shape
    out = torch.empty((b, h, t, d), dtype=scale_y.dtype, device=h_q.device)
    check(lib().zl_quant_back_transpose(_p(h_q), _p(scale_x), _p(scale_y), _p(out), _i(b), _i(t), _i(h), _i(d),
                                        C.c_int(_dt(out)), _stream()), "quant_back_transpose")
    return out


def quant_back_copy_to_buffer(src, scale_x, scale_y, placement, dst):
    """int8_op::quant_back_copy_to_buffer (quant_kernel.cu:389-469): src (B, len_kv, H, D) or (len_kv, H, D) int32
    scattered into dst (B, H, len_buf, D) / (H, len_buf, D) at placement (None = identity, < 0 = skipped)."""
    _chk_cuda(src, scale_x, scale_y, placement, dst)
    if not ((src.dim() == 3 and dst.dim() == 3) or (src.dim() == 4 and dst.dim() == 4)):
        raise ZLError("src and dst must be 3/4-dimensional")
    batch = 1 if src.dim() == 3 else src.shape[0]
    len_kv, h, d = src.shape[-3:]
    len_buf = dst.shape[-2]
    if dst.shape[-3] != h or dst.shape[-1] != d:
        raise ZLError("dim mismatch")
    src_stride = 0 if src.dim() == 3 else src.stride(0)
    dst_stride = 0 if dst.dim() == 3 else dst.stride(0)
    place_stride = 0 if placement is None or placement.dim() == 1 else placement.stride(0)
    check(lib().zl_quant_back_copy_to_buffer(_p(src), _p(scale_x), _p(scale_y), _p(placement), _p(dst), _i(batch),
                                             _i(len_kv), _i(h), _i(d), _i(len_buf), _i(src_stride), _i(dst_stride),
                                             _i(place_stride), C.c_int(_dt(dst)), _stream()), "quant_back_copy_to_buffer")
    return dst


# --------------------------------------------------------------------------------------------------
# a7  AWQ checkpoints in their on-disk layout (no repack) and f3  W4A8
# --------------------------------------------------------------------------------------------------
def awq_dequantize(qweight, qzeros, scales, group_size):
    """nn::awq::awq_dequantize (src/nn/quant/awq/gemm_kernels.cu:277-330): (K, N/8) int32 -> W16 (K, N) fp16."""
    _chk_cuda(qweight, qzeros, scales)
    k, n = qweight.shape[0], qweight.shape[1] * 8
    out = torch.empty((k, n), dtype=torch.float16, device=qweight.device)
    check(lib().zl_awq_dequantize(_p(qweight), _p(qzeros), _p(scales), _p(out), _i(k), _i(n), _i(group_size), _stream()),
          "awq_dequantize")
    return out


def awq_gemm(x, qweight, qzeros, scales, group_size, split_k_iters=32, out=None):
    """nn::awq::awq_gemm (gemm_kernels.cu:404-468) on the AWQ tensors as stored: split-K with fp16 partials summed in fp32."""
    if x.dtype != torch.float16:
        raise ZLError("in_feats must be half")
    _chk_cuda(x, qweight, qzeros, scales)
    x2 = x.reshape(-1, x.shape[-1])
    m, k = x2.shape
    n = qweight.shape[1] * 8
    if qweight.shape[0] != k:
        raise ZLError("size K mismatch")
    if n % 64:
        raise ZLError("OC is not multiple of cta_N = 64")
    if group_size % 32:
        raise ZLError("Group size should be a multiple of 32")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float16, device=x.device)
    ws = torch.empty(int(lib().zl_awq_gemm_workspace_bytes(_i(min(m, 8)), _i(n), _i(split_k_iters))), dtype=torch.uint8, device=x.device)
    check(lib().zl_awq_gemm(_p(x2), _i(x2.stride(0)), _p(qweight), _p(qzeros), _p(scales), _p(out), _p(ws), _i(m), _i(n), _i(k),
                            _i(group_size), _i(split_k_iters), _stream()), "awq_gemm")
    return out


def w4a8_weight_to_int8(w16):
    """Int4GPTQ::calc_w4a8_scale + dequant_k_major(out_type 1): W16 (N, K) fp16 -> (w8 int8 (N, K), scale fp32 (N))."""
    _chk_cuda(w16)
    n, k = w16.shape
    w8 = torch.empty((n, k), dtype=torch.int8, device=w16.device)
    sc = torch.empty((n,), dtype=torch.float32, device=w16.device)
    check(lib().zl_w4a8_weight_to_int8(_p(w16), _p(w8), _p(sc), _i(n), _i(k), _stream()), "w4a8_weight_to_int8")
    return w8, sc


def quant_scale_back_f32(c, sx, sy, out=None):
    """quant_scale_back with an fp32 per-row weight scale (the W4A8 forward): half(float(c) * sx[m] * sy[n])"""
    _chk_cuda(c, sx, sy)
    m, n = c.shape
    if out is None:
        out = torch.empty((m, n), dtype=torch.float16, device=c.device)
    check(lib().zl_quant_scale_back_f32(_p(c), _p(sx), _p(sy), _p(out), _i(m), _i(n), _stream()), "quant_scale_back_f32")
    return out


def w4a8_linear(x, w8, w_scale, out=None):
    """gptq_gemm_k_major's W4_INT8 branch (q_gemm_k_major.cu:1036-1073): quantise the activation rows, int8 x int8 -> int32
    on the matrix cores, scale back with the fp32 weight scale."""
    xq, sx = quant_calc_scale(x)
    return quant_scale_back_f32(int8_gemm_nt(xq, w8), sx, w_scale, out=out)


# ---- INT8-compressed tensor-parallel reduce (ModelContext::reduce_tp_int8, src/model/model_context.cpp:244-326)
def quant_group_32(x):
    """int8_op::quant_group_32: x (..., 32 * g) T -> (codes int8 same shape, scales T (groups,))"""
    _chk_cuda(x)
    if x.numel() % 32:
        raise ZLError("[quant_group_32]")
    groups = x.numel() // 32
    q = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    s = torch.empty(groups, dtype=x.dtype, device=x.device)
    check(lib().zl_quant_group_32(_p(x), _p(q), _p(s), _i(groups), C.c_int(_dt(x)), _stream()), "quant_group_32")
    return q, s


def dequant_sum_quant_g32(my, q_others, scale_others):
    """int8_op::dequant_sum_quant_g32: my (M, 32) T, q_others (WS - 1, M, 32) int8, scale_others (WS - 1, M) T"""
    _chk_cuda(my, q_others, scale_others)
    groups, world = my.numel() // 32, q_others.shape[0] + 1
    q = torch.empty(my.shape, dtype=torch.int8, device=my.device)
    s = torch.empty(groups, dtype=my.dtype, device=my.device)
    check(lib().zl_dequant_sum_quant_g32(_p(my), _p(q_others), _p(scale_others), _p(q), _p(s), _i(groups), C.c_int(world),
                                         C.c_int(_dt(my)), _stream()), "dequant_sum_quant_g32")
    return q, s


def dequant_group_32(q, scale, out=None):
    _chk_cuda(q, scale)
    if out is None:
        out = torch.empty(q.shape, dtype=scale.dtype, device=q.device)
    check(lib().zl_dequant_group_32(_p(q), _p(scale), _p(out), _i(q.numel() // 32), C.c_int(_dt(scale)), _stream()), "dequant_group_32")
    return out


# ---- W4A8 with FP8 activations (gptq_gemm_k_major, W4_FP8_ALGO: q_gemm_k_major.cu:1003-1035; fp8_util.cu)
def fp8_calc_scale(x, max_e4m3=448.0):
    """nn::fp8::calc_scale: a (1,) fp32 DEVICE tensor max|x| / max_e4m3"""
    _chk_cuda(x)
    s = torch.empty(1, dtype=torch.float32, device=x.device)
    check(lib().zl_fp8_calc_scale(_p(x), _i(x.numel()), _f(max_e4m3), _p(s), C.c_int(_dt(x)), _stream()), "fp8_calc_scale")
    return s


def fp8_cvt_half(x, scale):
    """E4M3FN codes (uint8) of T(x) * T(1 / scale), round to nearest even, saturating (T_KERNEL_cvt_half_fp8)"""
    _chk_cuda(x, scale)
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib().zl_fp8_cvt_half(_p(x), _p(scale), _p(out), _i(x.numel()), C.c_int(_dt(x)), _stream()), "fp8_cvt_half")
    return out


def fp8_dynamic_scaled_quant(x, max_e4m3=448.0):
    """nn::fp8::dynamic_scaled_quant: (codes, scale)"""
    s = fp8_calc_scale(x, max_e4m3)
    return fp8_cvt_half(x, s), s


def w4a8_weight_to_fp8(w16, max_weight_e4m3=256.0):
    """Int4GPTQ::calc_w4a8_scale (W4_FP8_ALGO) + dequant_k_major(out_type 2): W16 (N, K) fp16 -> (w8 E4M3 codes, scale (1,) fp32)"""
    return fp8_dynamic_scaled_quant(w16, max_weight_e4m3)


def fp8_gemm_nt(a8, b8, scale_a, scale_b, out=None):
    _chk_cuda(a8, b8, scale_a, scale_b)
    m, k = a8.shape
    n = b8.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float16, device=a8.device)
    check(lib().zl_fp8_gemm_nt(_p(a8), _p(b8), _p(scale_a), _p(scale_b), _p(out), _i(m), _i(n), _i(k), _stream()), "fp8_gemm_nt")
    return out


def w4a8_fp8_linear(x, w8, w_scale, max_act_e4m3=448.0, out=None):
    """gptq_gemm_k_major's W4_FP8 branch: per-tensor dynamic E4M3 quantisation of the activations, fp8 x fp8 GEMM, scales folded
    into the fp32 accumulators, one rounding to fp16"""
    a8, sa = fp8_dynamic_scaled_quant(x, max_act_e4m3)
    return fp8_gemm_nt(a8, w8, sa, w_scale, out=out)


# ---- f4, first part: FP8 128x128-block linear (config 5) and the MoE router ----------------------------------------------
def fp8_per_token_cast(x, scale_col_major=True, max_e4m3=448.0):
    """nn::fp8::per_token_cast_to_fp8: (codes (m, n) uint8, scales fp32 (n/128, aligned_m) column-major -- what fp8_block_gemm
    reads -- or (aligned_m, n/128)); aligned_m = round_up(m, 4) like the reference's TMA alignment"""
    if not (x.is_cuda and x.dim() == 2 and x.stride(1) == 1):        # rows may be strided (Fp8Block::support_uncontinuous_input)
        raise ZLError("FP8 block_scale: input is not a 2D CUDA tensor with dense rows")
    if x.shape[1] % 128:
        raise ZLError("FP8 block_scale: input.size(1) can't divide 128")
    m, n = x.shape
    am = (m + 3) // 4 * 4
    out = torch.empty((m, n), dtype=torch.uint8, device=x.device)
    sc = torch.zeros((n // 128, am) if scale_col_major else (am, n // 128), dtype=torch.float32, device=x.device)
    check(lib().zl_fp8_per_token_cast(_p(x), _i(x.stride(0)), _p(out), _i(n), _p(sc), _i(am), _i(m), _i(n), C.c_int(int(scale_col_major)),
                                      _f(max_e4m3), C.c_int(_dt(x)), _stream()), "fp8_per_token_cast")
    return out, sc


def fp8_block_dequant(w8, scale, dtype=torch.bfloat16):
    """nn::fp8::dequant_fp8_block_weight: (N, K) codes x (ceil(N/128), ceil(K/128)) fp32 -> (N, K) T"""
    _chk_cuda(w8, scale)
    out = torch.empty(w8.shape, dtype=dtype, device=w8.device)
    check(lib().zl_fp8_block_dequant(_p(w8), _p(scale), _p(out), _i(w8.shape[0]), _i(w8.shape[1]), _i(scale.shape[1]),
                                     C.c_int(0 if dtype == torch.float16 else 1), _stream()), "fp8_block_dequant")
    return out


class Fp8BlockMWeight:
    """The e4m3 codes of a block-scaled weight -- (N, K) or (G, N, K) -- in the ZLF8M layout (zl_fp8_block_pack): fragment loads of
    1 KiB contiguous.  Pass it as `w8` to fp8_block_gemm / fp8_block_linear (up to 32 rows per launch, or the grouped form)."""

    def __init__(self, w8):
        _chk_cuda(w8)
        if w8.dtype != torch.uint8 or w8.dim() not in (2, 3):
            raise ZLError("Fp8BlockMWeight: (N, K) or (G, N, K) uint8 codes")
        self.groups = w8.shape[0] if w8.dim() == 3 else 1
        self.n, self.k = w8.shape[-2], w8.shape[-1]
        nbytes = int(lib().zl_fp8_block_packed_bytes(_i(self.n), _i(self.k), _i(self.groups)))
        if nbytes < 0:
            check(nbytes, "fp8_block_packed_bytes")
        self.data = torch.empty(nbytes, dtype=torch.uint8, device=w8.device)
        w = w8 if w8.is_contiguous() else w8.contiguous()
        check(lib().zl_fp8_block_pack(_p(w), _p(self.data), _i(self.n), _i(self.k), _i(self.groups), _stream()), "fp8_block_pack")


def fp8_block_gemm(a8, a_scale, w8, w_scale, m_indices=None, dtype=torch.bfloat16, out=None):
    """deep_gemm_fp8_block_h20_group: a8 (m, k) codes with column-major scales (k/128, aligned_m); w8 (n, k) or (G, n, k) codes
    with scales (ceil(n/128), k/128) or (G, ...); m_indices (m,) int32 = the expert of every row (grouped form)"""
    if isinstance(w8, Fp8BlockMWeight):
        _chk_cuda(a8, a_scale, w_scale, m_indices)
        m, k = a8.shape
        if out is None:
            out = torch.zeros((m, w8.n), dtype=dtype, device=a8.device)
        check(lib().zl_fp8_block_gemm_group_packed(_p(a8), _p(a_scale), _i(a_scale.shape[1]), _p(w8.data), _p(w_scale), _p(m_indices), _p(out), _i(m),
                                                   _i(w8.n), _i(k), C.c_int(w8.groups), C.c_int(0 if out.dtype == torch.float16 else 1), _stream()),
              "fp8_block_gemm_packed")
        return out
    _chk_cuda(a8, a_scale, w8, w_scale, m_indices)
    m, k = a8.shape
    n = w8.shape[-2]
    groups = w8.shape[0] if w8.dim() == 3 else 1
    if out is None:
        out = torch.zeros((m, n), dtype=dtype, device=a8.device)
    check(lib().zl_fp8_block_gemm_group(_p(a8), _p(a_scale), _i(a_scale.shape[1]), _p(w8), _p(w_scale), _p(m_indices), _p(out), _i(m), _i(n),
                                        _i(k), C.c_int(groups), C.c_int(0 if out.dtype == torch.float16 else 1), _stream()), "fp8_block_gemm")
    return out


def fp8_block_linear(x, w8, w_scale, out=None):
    """Fp8Block::forward (linear.cpp:1863-1905): per-token 1x128 activation quantisation + the block-scaled GEMM"""
    a8, sa = fp8_per_token_cast(x)
    return fp8_block_gemm(a8, sa, w8, w_scale, dtype=x.dtype, out=out)


_SCORING = {"": 1, "softmax": 1, "sigmoid": 2, "linear": 3}


def moe_top_k_softmax(logits, top_k, top_k_ext=None, norm_topk_prob=False, weight_scale=1.0, scoring_func="softmax", worker_load=None,
                      expert_load=None, num_worker=1):
    """nn::top_k_softmax: (weights fp32 (tokens, top_k_ext), expert ids int32 (tokens, top_k_ext))"""
    _chk_cuda(logits, worker_load, expert_load)
    t, e = logits.shape
    ext = top_k_ext or top_k
    v = torch.empty((t, ext), dtype=torch.float32, device=logits.device)
    idx = torch.zeros((t, ext), dtype=torch.int32, device=logits.device)
    check(lib().zl_moe_top_k_softmax(_p(logits), _i(t), C.c_int(e), C.c_int(top_k), C.c_int(ext), C.c_int(int(norm_topk_prob)), _f(weight_scale),
                                     C.c_int(_SCORING[scoring_func]), C.c_int(_dt_logits(logits)), _p(v), _p(idx), _p(worker_load), _p(expert_load),
                                     C.c_int(num_worker), _stream()), "moe_top_k_softmax")
    return v, idx


def moe_group_topk(logits, score_correction_bias, num_group, topk_group, top_k, top_k_ext=None, norm_topk_prob=True, weight_scale=1.0,
                   scoring_func="sigmoid", worker_load=None, expert_load=None, num_worker=1):
    """nn::group_topk_softmax (DeepSeek-V3 group-limited routing)"""
    _chk_cuda(logits, score_correction_bias, worker_load, expert_load)
    t, e = logits.shape
    ext = top_k_ext or top_k
    v = torch.empty((t, ext), dtype=torch.float32, device=logits.device)
    idx = torch.zeros((t, ext), dtype=torch.int32, device=logits.device)
    check(lib().zl_moe_group_topk(_p(logits), _p(score_correction_bias), _i(t), C.c_int(e), C.c_int(top_k), C.c_int(ext), C.c_int(int(norm_topk_prob)),
                                  _f(weight_scale), C.c_int(_SCORING[scoring_func]), C.c_int(num_group), C.c_int(topk_group), C.c_int(_dt_logits(logits)),
                                  _p(v), _p(idx), _p(worker_load), _p(expert_load), C.c_int(num_worker), _stream()), "moe_group_topk")
    return v, idx


# ---- f4: MoE dispatch / combine of the prompt path (ff_kernel.h:42-96) ----------------------------------------------------------
def moe_sum_experts(inp, index, weights):
    """nn::sum_experts, one concatenated input (seq_len * K, dim_model)"""
    _chk_cuda(inp, index, weights)
    seq, k = weights.shape
    out = torch.empty((seq, inp.shape[1]), dtype=inp.dtype, device=inp.device)
    check(lib().zl_moe_sum_experts(_p(inp), _p(index), _p(weights), _p(out), _i(seq), C.c_int(k), _i(inp.shape[1]), C.c_int(_dt(inp)), _stream()),
          "sum_experts")
    return out


def moe_sum_experts_arr(inputs, experts, index, weights, exp_parallel=False, world_size=1, local_rank=0):
    """nn::sum_experts, one (rows, dim_model) input per expert (None / empty where an expert got no token)"""
    live = [t for t in inputs if t is not None and t.numel()]
    if not live:
        raise ZLError("all inputs is empty")
    _chk_cuda(experts, index, weights, *live)
    dev, dim = live[0].device, live[0].shape[-1]
    ptrs = torch.tensor([0 if (t is None or not t.numel()) else t.data_ptr() for t in inputs], dtype=torch.int64).to(dev)
    seq, k = weights.shape
    out = torch.empty((seq, dim), dtype=live[0].dtype, device=dev)
    check(lib().zl_moe_sum_experts_arr(_p(ptrs), _p(experts), _p(index), _p(weights), _p(out), _i(seq), C.c_int(k), _i(dim), C.c_int(int(exp_parallel)),
                                       C.c_int(world_size), C.c_int(local_rank), C.c_int(_dt(live[0])), _stream()), "sum_experts")
    return out


def moe_route_shared_lb(exp_ids, exp_weights, worker_load, expert_load, top_k, num_local_experts):
    """nn::route_shared_lb: fills exp_ids[:, top_k:] in place, bumps the two load counters"""
    _chk_cuda(exp_ids, exp_weights, worker_load, expert_load)
    if exp_ids.shape != exp_weights.shape or exp_ids.dtype != torch.int32:
        raise ZLError("shape mismatch")
    seq, ext = exp_ids.shape
    ws = worker_load.numel()
    max_load = (exp_ids.numel() + ws - 1) // ws
    base = worker_load.clone()
    check(lib().zl_moe_route_shared_lb(_p(exp_ids), _p(base), _p(worker_load), _p(expert_load), C.c_int(max_load), C.c_int(ws), _i(seq), C.c_int(top_k),
                                       C.c_int(ext), C.c_int(num_local_experts), _stream()), "route_shared_lb")


def moe_plus_for_sort(exp_ids, num_experts, world_size):
    _chk_cuda(exp_ids)
    if num_experts % world_size:
        raise ZLError("num_experts can't divide world_size")
    out = torch.empty_like(exp_ids)
    check(lib().zl_moe_plus_for_sort(_p(exp_ids), _p(out), C.c_int(num_experts), C.c_int(world_size), _i(exp_ids.numel()), _stream()), "plus_for_sort")
    return out


def moe_calc_reverse_idx(exp_ids, indices, all_loads, num_experts, world_size=1, sorted_by_rank=False):
    """nn::calc_reverse_idx: all_loads = the host copy of [expert loads (num_experts) | rank loads (world_size)]"""
    _chk_cuda(exp_ids, indices)
    off = [0] * num_experts
    if sorted_by_rank:
        rank_offset = 0
        for rank in range(world_size):
            o = 0
            for i in range(rank, num_experts, world_size):
                off[i] = rank_offset + o
                o += int(all_loads[i])
            rank_offset += int(all_loads[num_experts + rank])
    else:
        o = 0
        for i in range(num_experts):
            off[i] = o
            o += int(all_loads[i])
    off_t = torch.tensor(off, dtype=torch.int32).to(exp_ids.device)
    rev = torch.empty_like(indices)
    check(lib().zl_moe_calc_reverse_idx(_p(exp_ids), _p(indices), _p(off_t), _p(rev), _i(indices.numel()), _stream()), "calc_reverse_idx")
    return rev


def moe_fill_m_indices_padded_indices(all_loads, block_m, num_experts, device, exp_parallel=False, rank=0, world_size=1):
    """nn::fill_m_indices_padded_indices: (m_indices, padded_indices, total_tokens_aligned)"""
    r, ws = (rank, world_size) if exp_parallel else (0, 1)
    nt = [int(all_loads[j]) for j in range(r, num_experts, ws)]
    offs, aoffs, o, a = [], [], 0, 0
    for n in nt:
        offs.append(o)
        aoffs.append(a)
        o += n
        a += (n + block_m - 1) // block_m * block_m
    aoffs.append(a)
    if a == 0:
        return None, None, 0
    # one device tensor for the three small tables (kept alive in a local: temporaries would be freed -- and their block handed to
    # the next one -- before the launch is even issued)
    tab = torch.tensor(nt + offs + aoffs, dtype=torch.int32).to(device)
    n = len(nt)
    pad = torch.empty(max(o, 1), dtype=torch.int32, device=device)[:o]
    mi = torch.empty(a, dtype=torch.int32, device=device)
    check(lib().zl_moe_fill_m_indices(_p(tab[:n]), _p(tab[n:2 * n]), _p(tab[2 * n:]), _p(pad) if o else _p(mi), _p(mi), C.c_int(n), C.c_int(max(nt)),
                                      C.c_int(block_m), _stream()), "fill_m_indices_padded_indices")
    return mi, pad, a


# ---- f4: MLA decode attention over the latent cache ------------------------------------------------------------------------------
def mla_decode_attention(q_adj, buf_lens, kv_buf_addrs, scale, max_len_buf, valid_lens=None, kv_lora_rank=512, rope_dim=64, workspace=None,
                         algo=0):
    """MLAImpl over the compressed cache for decode rows: q_adj (B, H, 576), kv_buf_addrs (B,) int64 device table of (len_buf, 576)
    buffers -> (B, H, 512).  algo 0: the matrix-core kernel, 1: the VALU kernel (zl_mla_decode_attn_ex)."""
    _chk_cuda(q_adj, buf_lens, kv_buf_addrs, valid_lens)
    b, h, cd = q_adj.shape
    if cd != kv_lora_rank + rope_dim:
        raise ZLError("q_adj: last dimension is kv_lora_rank + rope_dim")
    if workspace is None:
        workspace = torch.empty(int(lib().zl_mla_decode_workspace_bytes(_i(b), _i(h), _i(max_len_buf))), dtype=torch.uint8, device=q_adj.device)
    out = torch.empty((b, h, kv_lora_rank), dtype=q_adj.dtype, device=q_adj.device)
    check(lib().zl_mla_decode_attn_ex(_p(q_adj), _p(buf_lens), _p(valid_lens), _p(kv_buf_addrs), _p(out), _p(workspace), _i(b), _i(h),
                                      _i(kv_lora_rank), _i(rope_dim), _f(scale), _i(max_len_buf), C.c_int(_dt(q_adj)), C.c_int(algo), _stream()),
          "mla_decode_attention")
    return out


def mha_fwd_kvcache_mla(q, kcache, head_size_v, seqlens_k, block_table, softmax_scale, is_causal=False, out=None):
    """ds::mha_fwd_kvcache_mla (ds_flash_mla_api.h:16-30): q (B, len_q, H, 576), kcache (num_blocks, page, 1, 576), seqlens_k (B,) int32,
    block_table (B, max_blocks) int32 -> (out (B, len_q, H, 512), softmax_lse (B, 1, len_q * H) fp32).  Non-causal only (len_q == 1 or
    is_causal False), as every reference call site is."""
    _chk_cuda(q, kcache, seqlens_k, block_table)
    b, len_q, h, cd = q.shape
    if len_q > 1 and is_causal:
        raise ZLError("mha_fwd_kvcache_mla: causal multi-row queries are not on this path")
    if kcache.dim() != 4 or kcache.shape[2] != 1 or kcache.shape[3] != cd or not kcache.is_contiguous() or not q.is_contiguous():
        raise ZLError("mha_fwd_kvcache_mla: kcache is (num_blocks, page_block_size, 1, head_size), contiguous like q")
    if block_table.dim() != 2 or block_table.shape[0] != b or block_table.dtype != torch.int32 or seqlens_k.dtype != torch.int32:
        raise ZLError("mha_fwd_kvcache_mla: block_table (B, max_blocks) int32, seqlens_k (B,) int32")
    page, mb, hh = kcache.shape[1], block_table.shape[1], len_q * h
    workspace = torch.empty(int(lib().zl_mla_decode_workspace_bytes(_i(b), _i(hh), _i(page * mb))), dtype=torch.uint8, device=q.device)
    if out is None:
        out = torch.empty((b, len_q, h, head_size_v), dtype=q.dtype, device=q.device)
    lse = torch.empty((b, 1, hh), dtype=torch.float32, device=q.device)
    block_table = block_table.contiguous()
    check(lib().zl_mla_decode_attn_paged(_p(q), _p(kcache), _p(block_table), _p(seqlens_k), _p(out), _p(lse), _p(workspace), _i(b), _i(hh),
                                         _i(head_size_v), _i(cd - head_size_v), _i(page), _i(mb), _f(softmax_scale), C.c_int(_dt(q)), _stream()),
          "mha_fwd_kvcache_mla")
    return out, lse

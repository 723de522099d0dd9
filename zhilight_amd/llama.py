"""Host-side mirror of the reference's decode ("search") path for LLaMA-like models:
`model::LLaMA::encode` -> N x `nn::EncoderLayer::forward` -> `get_logits`
(src/model/llama.cpp:75-165, src/nn/block/block.cpp:86-143, src/nn/attention/attention.cpp:846-964,
src/nn/feedforward/feedforward.cpp:113-137), with the per-step device state of
`model::DynBatchContext` / `RagBufferContext` (src/model/dyn_batch_context.h:29-200,
src/model/rag_buffer_context.h:141-188).

Only what the decode hot path needs lives here: weight containers, the per-task ragged KV buffers
with their device pointer tables, and the kernel sequence of one step.  PyTorch supplies device
memory and the stream; every arithmetic step is one C-ABI launcher from `ops`.

Kernel sequence per layer (6 launches; the reference issues ~14 for the same math):
  1. w4a16_gemm  [RMSNorm(ln_attn) prologue]  hidden -> fused q|k|v             (project_q/k/v)
  2. decode_attn_fused: rotate q,k (cached cos/sin), k,v -> ragged KV, split-KV softmax(q.K^T).V
  3. combine of the KV splits
  4. w4a16_gemm  [residual epilogue]           attn_out + hidden -> hidden       (attn_out + add)
  5. w4a16_gemm  [RMSNorm(ln_ff) prologue, silu*mul epilogue]  -> act           (w_in, w_gated, gate_mul)
  6. w4a16_gemm  [residual epilogue]           w_out + hidden -> hidden
The roundings are the reference's single-stream path: every linear output, the norm output, the
rotated q/k, the attention output and each residual sum is rounded to fp16 exactly where the
reference materialises an fp16 tensor.
"""
import math
import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from . import parallel


@dataclass
class ModelConfig:
    """Subset of model::ModelConfig (src/model/model_config.hpp:27-130) used on this path; key names
    follow `pydict_to_model_config` (src/py_export/py_model_config.cpp:77-151)."""
    num_layers: int = 32
    dim_model: int = 4096
    num_heads: int = 32
    dim_head: int = 128
    dim_ff: int = 14336
    vocab_size: int = 128256
    num_kv_heads: int = 8
    eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[dict] = None     # {"rope_type": "llama3", "factor":8, "low_freq_factor":1, ...}
    activate_fn: str = "silu"
    scale_emb: float = 1.0
    dim_model_base: int = 0                 # MiniCPM: logits scale = dim_model_base / dim_model
    scale_depth: float = -1.0               # MiniCPM residual scale scale_depth / sqrt(num_layers)
    tie_lm_head: bool = False
    qk_norm: Optional[str] = None           # "head": Qwen3 q_norm / k_norm (dim_head weight); "multi_head": use_qk_norm
    max_position_embeddings: int = 0        # dynamic-NTK rope threshold
    model_type: str = "llama"
    dtype: str = "half"                     # "half" | "bfloat16" (unquantised models; the W4 / int8 routes are fp16)

    @property
    def torch_dtype(self):
        return torch.bfloat16 if self.dtype in ("bfloat16", "bf16") else torch.float16

    @property
    def residual_scale(self):
        """EncoderLayer scale (src/nn/block/block.cpp:55-58): MiniCPM ("cpm_dragonfly") adds scale_depth/sqrt(L)
        times the sub-layer output to the residual; everything else a plain sum."""
        return (self.scale_depth / math.sqrt(self.num_layers)) if self.scale_depth > 0 else 1.0

    @classmethod
    def from_hf(cls, cfg: dict):
        """HF config.json -> ModelConfig (zhilight/config/adapter.py + py_model_config.cpp key names)."""
        hidden, heads = cfg["hidden_size"], cfg["num_attention_heads"]
        return cls(
            num_layers=cfg["num_hidden_layers"], dim_model=hidden, num_heads=heads,
            dim_head=cfg.get("head_dim", hidden // heads), dim_ff=cfg["intermediate_size"],
            vocab_size=cfg["vocab_size"], num_kv_heads=cfg.get("num_key_value_heads", heads),
            eps=cfg.get("rms_norm_eps", 1e-5), rope_theta=cfg.get("rope_theta", 10000.0),
            rope_scaling=cfg.get("rope_scaling"), activate_fn=cfg.get("hidden_act", "silu"),
            scale_emb=cfg.get("scale_emb", 1.0), dim_model_base=cfg.get("dim_model_base", 0),
            scale_depth=cfg.get("scale_depth", -1.0), tie_lm_head=cfg.get("tie_word_embeddings", False),
            qk_norm=("head" if cfg.get("model_type") in ("qwen3", "qwen3_moe") else
                     "multi_head" if cfg.get("use_qk_norm") else None),     # attention.cpp:110-117
            max_position_embeddings=cfg.get("max_position_embeddings", 0), model_type=cfg.get("model_type", "llama"),
            dtype={"bfloat16": "bfloat16", "float16": "half"}.get(cfg.get("torch_dtype", "float16"), "half"))

    @classmethod
    def llama3_8b(cls):
        return cls()

    @classmethod
    def minicpm_2b(cls):
        """openbmb/MiniCPM-2B-sft-bf16 (BASELINE configs[0]): "cpm_dragonfly" scalings, tied lm_head, D = 64."""
        return cls(num_layers=40, dim_model=2304, num_heads=36, dim_head=64, dim_ff=5760, vocab_size=122753, num_kv_heads=36,
                   eps=1e-5, rope_theta=10000.0, scale_emb=12.0, dim_model_base=256, scale_depth=1.4, tie_lm_head=True,
                   dtype="bfloat16")


@dataclass
class QuantConfig:
    """model::QuantConfig (src/model/model_config.hpp:132-177); quant_type 5 = GPTQ W4A16 k-major.
    `awq` marks an AWQ checkpoint taken through the reference's AWQ_USE_EXLLAMA route
    (Int4GPTQ with is_awq, src/nn/linear/linear.cpp:694-698, 1139-1143): the tensors are converted to the
    same k-major operands at load, the kernels are the GPTQ ones."""
    quant_type: int = 5
    group_size: int = 128
    sym: bool = False
    act_order: bool = False
    awq: bool = False

    @classmethod
    def from_hf(cls, qc: dict):
        """QuantConfig.adapt_hf_config (zhilight/quant.py:35-88): gptq -> type 5; awq -> type 5 with the AWQ
        load transform (what the reference does under AWQ_USE_EXLLAMA=1, its faster decode route)."""
        method = qc.get("quant_method", "gptq")
        if method not in ("gptq", "awq"):
            raise ops.ZLError(f"Unsupported quant_method {method}")
        if qc.get("bits", 4) != 4:
            raise ops.ZLError("Only bits=4 is supported")
        if qc.get("is_marlin_format", False):
            raise ops.ZLError(f"Unsupported Marlin {method}")
        act_order = bool(qc.get("desc_act", False))
        if method == "awq":
            if not qc.get("zero_point", True):
                raise ops.ZLError("AWQ checkpoints without zero points are not supported")
            return cls(5, qc.get("group_size", 128), False, False, True)
        return cls(5, qc.get("group_size", 128), qc.get("sym", False), act_order, False)


def hf_name_to_internal(name: str) -> str:
    """LLaMALoader._replace_name (zhilight/loader.py:250-358), LLaMA/GPTQ subset."""
    s = name
    s = re.sub(r"model\.embed_tokens\.weight", "token_embedding.weight", s)
    s = re.sub(r"model\.norm\.weight", "output_layernorm.weight", s)
    s = re.sub(r"model\.layers\.([0-9]+)\.input_layernorm\.", r"layers.\1.ln_attn.", s)
    s = re.sub(r"model\.layers\.([0-9]+)\.post_attention_layernorm\.weight", r"layers.\1.ln_ff.weight", s)
    s = re.sub(r"model\.layers\.([0-9]+)\.self_attn\.([qkv])_proj\.", r"layers.\1.attn.project_\2.", s)
    s = re.sub(r"model\.layers\.([0-9]+)\.self_attn\.o_proj\.", r"layers.\1.attn.attn_out.", s)
    s = re.sub(r"model\.layers\.([0-9]+)\.self_attn\.([qk])_norm\.", r"layers.\1.attn.\2_norm.", s)
    s = re.sub(r"model\.layers\.([0-9]+)\.mlp\.gate_proj\.", r"layers.\1.ff.w_in.", s)
    s = re.sub(r"model\.layers\.([0-9]+)\.mlp\.up_proj\.", r"layers.\1.ff.w_gated.", s)
    s = re.sub(r"model\.layers\.([0-9]+)\.mlp\.down_proj\.", r"layers.\1.ff.w_out.", s)
    return "llama." + s


def _dev_t(a, device, dtype=None):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    t = t.to(device).contiguous()
    return t if dtype is None else t.view(dtype)


def w4_algo(quant) -> str:
    """Which W4A16 kernel a quantized linear runs -- the counterpart of the reference's GPTQ_KERNEL_ALGO
    switch (src/nn/quant/gptq/q_gemm_k_major.cu:1075-1100).  "mfma" (default): dequant -> MFMA with fp32
    accumulation, the arithmetic of the reference's dequant+GEMM branch; "exact": the warp-reduce
    kernel replayed bit for bit.  Set ZL_W4_ALGO=exact to select the latter; group sizes the MFMA tile
    cannot hold (not a multiple of 128) use it automatically."""
    algo = os.environ.get("ZL_W4_ALGO", "mfma").lower()
    if algo not in ("mfma", "exact"):
        raise ops.ZLError(f"ZL_W4_ALGO={algo!r}: expected 'mfma' or 'exact'")
    if quant.group_size % 128 != 0:
        return "exact"
    return algo


class Int4GPTQ:
    """nn::Linear with the Int4GPTQ implementation (src/nn/linear/linear.cpp:638-1244): holds the
    load-time-transformed weight; `fuse` mirrors Linear::fuse / fuse3 (row concatenation)."""

    def __init__(self, name, dim_in, dim_out, quant: QuantConfig):
        self.name, self.dim_in, self.dim_out, self.quant = name, dim_in, dim_out, quant
        self.km = None      # k-major (qweight, qzeros, scales) device tensors until packed
        self.weight: Optional[ops.W4Weight] = None
        self.bias = None
        self.perm = None    # act-order: activation column gather applied in forward()

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str, device, out_perm=None, input_prepermuted=False):
        """out_perm: reorder the OUTPUT columns (w_in / w_gated permuted by w_out's input order, the reference's
        permute_ff_up_out, linear.cpp:1168-1186).  input_prepermuted: the producer already delivers the activations in this
        linear's regrouped row order (w_out behind such w_in / w_gated): no gather in forward (linear.cpp:1188-1210)."""
        qw = _dev_t(sd[prefix + ".qweight"], device, torch.int32)
        qz = _dev_t(sd[prefix + ".qzeros"], device, torch.int32)
        sc = _dev_t(sd[prefix + ".scales"], device, torch.float16)
        if out_perm is not None:
            if self.quant.awq:
                raise ops.ZLError("output permutation is a GPTQ act-order transform")
            idx = out_perm.to(torch.int64)
            qw = qw.index_select(1, idx).contiguous()
            sc = sc.index_select(1, idx).contiguous()
            shifts = torch.arange(8, device=qz.device, dtype=torch.int32) * 4
            zn = ((qz.unsqueeze(-1) >> shifts) & 0xF).reshape(qz.shape[0], -1).index_select(1, idx).reshape(qz.shape[0], -1, 8)
            qz = torch.zeros_like(qz)
            for j in range(8):
                qz |= zn[:, :, j] << (4 * j)
        if self.quant.awq:
            # AWQ on disk: qweight (K, N/8) with the [0,4,1,5,2,6,3,7] nibble order, zeros stored as they are
            # used (no +1).  Int4GPTQ::preprocess_weight, is_awq branch: shuffle_awq -> (K/8, N) exllama
            # words, un_shuffle of the zero nibbles; from there on it is the GPTQ operand set.
            if qw.shape != (self.dim_in, self.dim_out // 8):
                raise ops.ZLError(f"{prefix}: qweight shape {tuple(qw.shape)} != {(self.dim_in, self.dim_out // 8)}")
            qw = ops.transpose_2d(ops.shuffle_awq(qw, True))
            qz = ops.transpose_2d(ops.q4_to_q8(ops.awq_un_shuffle(qz.clone())))
        else:
            if qw.shape != (self.dim_in // 8, self.dim_out):
                raise ops.ZLError(f"{prefix}: qweight shape {tuple(qw.shape)} != {(self.dim_in // 8, self.dim_out)}")
            if self.quant.act_order and prefix + ".g_idx" in sd:
                qw = self._apply_act_order(qw, _dev_t(sd[prefix + ".g_idx"], device, torch.int32), prefix)
            # Int4GPTQ::preprocess_weight + transpose_weight: shuffle, +1 zeros, nibble->byte, transposes
            qw = ops.transpose_2d(ops.gptq_shuffle(qw.clone()))
            qz = ops.transpose_2d(ops.q4_to_q8(ops.increase_zero(qz.clone())))
        self.km = (qw, qz, ops.transpose_2d(sc))
        if input_prepermuted:
            self.perm = None
        if prefix + ".bias" in sd:
            self.bias = _dev_t(sd[prefix + ".bias"], device, torch.float16)
            if out_perm is not None:
                self.bias = self.bias.index_select(0, out_perm.to(torch.int64)).contiguous()

    def _apply_act_order(self, qw, g_idx, prefix):
        """desc_act checkpoints (the reference sorts with argsort(g_idx) and lets its exllama
        kernel read x through that permutation, src/nn/linear/linear.cpp:1144-1147, q_gemm.cu:104-251).  Same
        math, done at the boundary: the weight ROWS are put in group order once at load (then every group is
        128 consecutive k again and zeros/scales are already in group order), and forward() gathers the
        activation columns with the same permutation before the ordinary kernel runs."""
        k, g = self.dim_in, self.quant.group_size
        if g_idx.numel() != k:
            raise ops.ZLError(f"{prefix}: g_idx has {g_idx.numel()} entries, expected {k}")
        perm = torch.argsort(g_idx.to(torch.int64), stable=True)
        if not torch.equal(g_idx.to(torch.int64)[perm], torch.arange(k, device=g_idx.device) // g):
            raise ops.ZLError(f"{prefix}: g_idx is not a regrouping into groups of {g}")
        if torch.equal(perm, torch.arange(k, device=perm.device)):
            return qw                                     # trivial order (static groups)
        shifts = torch.arange(8, device=qw.device, dtype=torch.int32) * 4
        nib = ((qw.unsqueeze(1) >> shifts.view(1, 8, 1)) & 0xF).reshape(k, -1)       # (K, N) nibbles, row 8r + j
        nib = nib.index_select(0, perm).reshape(k // 8, 8, -1)
        packed = torch.zeros_like(qw)
        for j in range(8):
            packed |= nib[:, j, :] << (4 * j)
        self.perm = perm.to(torch.int32)
        return packed

    @staticmethod
    def fuse(name, parts: List["Int4GPTQ"], row_interleave=False):
        if not Int4GPTQ.same_perm(parts):
            raise ops.ZLError("act-order linears with different input permutations cannot be fused")
        out = Int4GPTQ(name, parts[0].dim_in, sum(p.dim_out for p in parts), parts[0].quant)
        out.perm = parts[0].perm                        # shared: q/k/v (gate/up) see the same input, so GPTQ gives them one order
        out.km = tuple(torch.cat([p.km[i] for p in parts], dim=0).contiguous() for i in range(3))
        if any(p.bias is not None for p in parts):
            out.bias = torch.cat([p.bias if p.bias is not None else
                                  torch.zeros(p.dim_out, dtype=torch.float16, device=out.km[0].device) for p in parts])
        out.pack(row_interleave)
        return out

    @staticmethod
    def same_perm(parts):
        first = parts[0].perm
        return all((p.perm is None) == (first is None) and (first is None or torch.equal(p.perm, first)) for p in parts)

    def pack(self, row_interleave=False):
        if w4_algo(self.quant) == "mfma":
            self.weight = ops.W4MWeight.from_k_major(*self.km, self.quant.group_size, row_interleave)
        else:
            self.weight = ops.W4Weight.from_k_major(*self.km, self.quant.group_size, self.quant.sym, row_interleave)
        if row_interleave and self.bias is not None:
            half = self.dim_out // 2
            self.bias = torch.stack([self.bias[:half], self.bias[half:]], dim=1).reshape(-1).contiguous()
        self.km = None
        return self

    def forward(self, x, **kw):
        if self.perm is not None:
            if kw.get("norm_weight") is not None:
                raise ops.ZLError("act-order linears take an already normalised input")
            x = ops.permute_input(x, self.perm)
        return ops.w4_linear(x, self.weight, bias=self.bias, **kw)


class AWQLinear:
    """nn::Linear with the AWQ implementation (src/nn/linear/linear.cpp:1400-1600, what an AWQ checkpoint runs on when
    AWQ_USE_EXLLAMA is not set): the checkpoint tensors stay in their on-disk layout -- qweight (K, N/8), qzeros (K/G, N/8),
    scales (K/G, N) -- and forward is nn::awq::awq_gemm below 256 rows (split-K 32, fp16 partials) or awq_dequantize + a
    fp32-accumulating GEMM from 256 rows on (linear.cpp:1560-1565).  The format-faithful route; the default AWQ route
    (AWQ_USE_EXLLAMA=1, Int4GPTQ above) re-tiles the same tensors once for the matrix-core kernels."""

    def __init__(self, name, dim_in, dim_out, quant: QuantConfig):
        self.name, self.dim_in, self.dim_out, self.quant = name, dim_in, dim_out, quant
        self.qweight = self.qzeros = self.scales = self.bias = None
        self.perm = None

    @property
    def weight(self):
        return self

    def nbytes(self):
        return self.qweight.numel() * 4 + self.qzeros.numel() * 4 + self.scales.numel() * 2

    def load_state_dict(self, sd, prefix, device):
        self.qweight = _dev_t(sd[prefix + ".qweight"], device, torch.int32).contiguous()
        self.qzeros = _dev_t(sd[prefix + ".qzeros"], device, torch.int32).contiguous()
        self.scales = _dev_t(sd[prefix + ".scales"], device, torch.float16).contiguous()
        if self.qweight.shape != (self.dim_in, self.dim_out // 8):
            raise ops.ZLError(f"{prefix}: qweight shape {tuple(self.qweight.shape)} != {(self.dim_in, self.dim_out // 8)}")
        if prefix + ".bias" in sd:
            self.bias = _dev_t(sd[prefix + ".bias"], device, torch.float16)
        return self

    def pack(self, row_interleave=False):
        return self

    def forward(self, x, out=None, residual=None, epilogue=0, **kw):
        if kw.get("norm_weight") is not None:
            raise ops.ZLError("AWQ linears take an already normalised input")
        g = self.quant.group_size
        x2 = x.reshape(-1, x.shape[-1])
        if x2.shape[0] < 256:
            y = ops.awq_gemm(x2, self.qweight, self.qzeros, self.scales, g)
        else:
            w16 = ops.transpose_2d(ops.awq_dequantize(self.qweight, self.qzeros, self.scales, g))      # (N, K)
            y = ops.gemm_nt(x2, w16)
        if self.bias is not None:
            y = y + self.bias                                # add_bias in T arithmetic
        if epilogue & ops.EPI_RESIDUAL:
            return ops.element_add_scale(residual, y, 1.0, True, out=out if out is not None else residual)
        if out is not None:
            out.copy_(y)
            return out
        return y


class NormalLinear:
    """nn::Linear with the NormalLinear implementation (src/nn/linear/linear.cpp:140-428): y = T(x . W^T + bias),
    fp32 accumulation.  Up to 4 rows the wave-per-row GEMV (HBM speed, optional fused RMSNorm), above the MFMA
    GEMM."""

    def __init__(self, name, dim_in, dim_out):
        self.name, self.dim_in, self.dim_out = name, dim_in, dim_out
        self.weight = self.bias = None

    def load_state_dict(self, sd, prefix, device, dtype):
        self.weight = _dev_t(sd[prefix + ".weight"], device).to(dtype).contiguous()
        if tuple(self.weight.shape) != (self.dim_out, self.dim_in):
            raise ops.ZLError(f"{prefix}: weight shape {tuple(self.weight.shape)} != {(self.dim_out, self.dim_in)}")
        if prefix + ".bias" in sd:
            self.bias = _dev_t(sd[prefix + ".bias"], device).to(dtype)
        return self

    @staticmethod
    def fuse(name, parts: List["NormalLinear"]):
        out = NormalLinear(name, parts[0].dim_in, sum(p.dim_out for p in parts))
        out.weight = torch.cat([p.weight for p in parts], dim=0).contiguous()
        if any(p.bias is not None for p in parts):
            out.bias = torch.cat([p.bias if p.bias is not None else torch.zeros(p.dim_out, dtype=out.weight.dtype,
                                                                                  device=out.weight.device) for p in parts])
        return out

    @classmethod
    def random(cls, name, dim_in, dim_out, device, gen, dtype):
        l = cls(name, dim_in, dim_out)
        l.weight = (torch.randn(dim_out, dim_in, device=device, generator=gen) * (0.7 / math.sqrt(dim_in))).to(dtype)
        return l

    def nbytes(self):
        return self.weight.numel() * 2

    def forward(self, x, out=None, norm_weight=None, norm_eps=1e-5):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.shape[0] <= 4 or self.dim_in % 128 != 0:
            return ops.gemm_nt_small_m(x2, self.weight, bias=self.bias, out=out, norm_weight=norm_weight, norm_eps=norm_eps)
        if norm_weight is not None:
            x2 = ops.rmsnorm(x2, norm_weight, norm_eps)
        # 5..32 rows (decode batches of the unquantised models, BASELINE configs[0]): the ZLD16M copy of the matrix -- 1 KiB contiguous
        # fragment loads, 5.95 instead of 5.0 TB/s on the Llama-3 lm_head, the same bits; packed on first use outside stream capture.
        # A second copy of the weights: ZL_DENSE_PACKED=0 keeps only the row-major one
        if x2.shape[0] <= 32 and os.environ.get("ZL_DENSE_PACKED", "1") != "0":
            if getattr(self, "weight_m", None) is None and not torch.cuda.is_current_stream_capturing():
                self.weight_m = ops.DenseMWeight(self.weight)
            if getattr(self, "weight_m", None) is not None:
                return ops.gemm_nt_packed(x2, self.weight_m, bias=self.bias, out=out)
        return ops.gemm_nt(x2, self.weight, bias=self.bias, out=out)


class DenseEncoderLayer:
    """EncoderLayer over NormalLinear (unquantised fp16 / bf16 models, BASELINE configs[0] MiniCPM): fused q|k|v
    projection, separate w_in / w_gated + gate_mul_inplace (src/nn/feedforward/feedforward.cpp:113-137), residual
    adds through element_add_scale_out with the layer scale (src/nn/block/block.cpp:123-140)."""

    def __init__(self, cfg: ModelConfig, quant: QuantConfig, idx: int):
        self.cfg, self.quant, self.idx = cfg, quant, idx
        self.unfused = None

    def load_state_dict(self, sd, prefix, device):
        c, dt = self.cfg, self.cfg.torch_dtype
        hd, kvd = c.num_heads * c.dim_head, c.num_kv_heads * c.dim_head
        self.ln_attn = _dev_t(sd[prefix + ".ln_attn.weight"], device).to(dt)
        self.ln_ff = _dev_t(sd[prefix + ".ln_ff.weight"], device).to(dt)
        lin = lambda sub, din, dout: NormalLinear(prefix + "." + sub, din, dout).load_state_dict(sd, prefix + "." + sub, device, dt)  # noqa: E731
        self.qkv = NormalLinear.fuse(prefix + ".attn.project_qkv",
                                     [lin("attn.project_q", c.dim_model, hd), lin("attn.project_k", c.dim_model, kvd),
                                      lin("attn.project_v", c.dim_model, kvd)])
        self.attn_out = lin("attn.attn_out", hd, c.dim_model)
        self.w_in, self.w_gated = lin("ff.w_in", c.dim_model, c.dim_ff), lin("ff.w_gated", c.dim_model, c.dim_ff)
        self.w_out = lin("ff.w_out", c.dim_ff, c.dim_model)

    def init_random(self, device, gen):
        c, dt = self.cfg, self.cfg.torch_dtype
        hd, kvd = c.num_heads * c.dim_head, c.num_kv_heads * c.dim_head
        self.ln_attn = (1.0 + 0.05 * torch.randn(c.dim_model, device=device, generator=gen)).to(dt)
        self.ln_ff = (1.0 + 0.05 * torch.randn(c.dim_model, device=device, generator=gen)).to(dt)
        self.qkv = NormalLinear.random("qkv", c.dim_model, hd + 2 * kvd, device, gen, dt)
        self.attn_out = NormalLinear.random("attn_out", hd, c.dim_model, device, gen, dt)
        self.w_in = NormalLinear.random("w_in", c.dim_model, c.dim_ff, device, gen, dt)
        self.w_gated = NormalLinear.random("w_gated", c.dim_model, c.dim_ff, device, gen, dt)
        self.w_out = NormalLinear.random("w_out", c.dim_ff, c.dim_model, device, gen, dt)

    def linears(self):
        return [self.qkv, self.attn_out, self.w_in, self.w_gated, self.w_out]

    def weight_bytes(self):
        return sum(l.nbytes() for l in self.linears())

    def project_qkv(self, hidden, eps, out=None):
        return self.qkv.forward(hidden, out=out, norm_weight=self.ln_attn, norm_eps=eps)

    def _add(self, hidden, sub):
        # residual first, T arithmetic: c = a + b * T(scale) for MiniCPM, (a + b) * T(1) otherwise
        rs = self.cfg.residual_scale
        ops.element_add_scale(hidden, sub, rs, scale_residual=(self.cfg.scale_depth <= 0), out=hidden)

    def attn_out_add(self, attn, hidden):
        self._add(hidden, self.attn_out.forward(attn))

    def ff_add(self, hidden, eps, act_buf=None):
        xn = ops.rmsnorm(hidden, self.ln_ff, eps)
        gate = self.w_in.forward(xn, out=act_buf)
        act = ops.gate_mul(gate, self.w_gated.forward(xn), "gelu" if self.cfg.activate_fn.startswith("gelu") else "silu")
        self._add(hidden, self.w_out.forward(act))


class Int8Linear:
    """nn::Linear with the Int8Linear implementation (src/nn/linear/linear.cpp:430-635, quant type 2 = AutoInt8):
    the fp16 weight is quantised per output row at load (quant_calc_scale: scale = amax / 127, stored in T),
    forward = per-row activation quantisation, int8 x int8 -> int32 GEMM, scale back."""

    def __init__(self, name, dim_in, dim_out):
        self.name, self.dim_in, self.dim_out = name, dim_in, dim_out
        self.weight = self.scale = self.bias = None     # int8 (N, K), T (N), T (N)

    def load_state_dict(self, sd, prefix, device):
        w = _dev_t(sd[prefix + ".weight"], device)
        if tuple(w.shape) != (self.dim_out, self.dim_in):
            raise ops.ZLError(f"{prefix}: weight shape {tuple(w.shape)} != {(self.dim_out, self.dim_in)}")
        self.weight, s = ops.quant_calc_scale(w)
        self.scale = s.to(w.dtype)                       # functions::typecast(w_scale, dtype)
        if prefix + ".bias" in sd:
            # the reference concatenates the biases in fuse3 and adds them after the scale-back (linear.cpp:459-461,
            # 631-632); the fused scale-back epilogues of this route carry none yet -- refuse instead of dropping it
            raise ops.ZLError(f"{prefix}: bias tensors are not supported on the int8 route (checkpoints with q/k/v biases, "
                              "e.g. Qwen2: use the GPTQ / fp16 routes)")
        return self

    @staticmethod
    def fuse(name, parts: List["Int8Linear"]):
        out = Int8Linear(name, parts[0].dim_in, sum(p.dim_out for p in parts))
        out.weight = torch.cat([p.weight for p in parts], dim=0).contiguous()
        out.scale = torch.cat([p.scale for p in parts]).contiguous()
        return out

    @classmethod
    def random(cls, name, dim_in, dim_out, device, gen):
        l = cls(name, dim_in, dim_out)
        l.weight = torch.randint(-127, 128, (dim_out, dim_in), dtype=torch.int8, device=device, generator=gen)
        l.scale = (torch.rand(dim_out, device=device, generator=gen) * (0.02 / math.sqrt(dim_in)) + 1e-5).to(torch.float16)
        return l

    def nbytes(self):
        return self.weight.numel() + self.scale.numel() * 2

    def gemm(self, xq):
        """int32 (M, N) = xq . W^T -- the caller fuses the scale-back"""
        return ops.int8_gemm_nt(xq, self.weight)

    def stream_weight(self):
        """the ZLW8M copy for the decode kernel (zl_w8a8_gemm_phase), packed on first use; the (N, K) rows stay for the
        tiled GEMM of prompt chunks and batches beyond 32 rows"""
        if getattr(self, "_w8m", None) is None:
            self._w8m = ops.W8MWeight.from_rows(self.weight, self.scale)
        return self._w8m


class Int8EncoderLayer:
    """EncoderLayer over Int8Linear (BASELINE configs[2]) with the fusions of the reference's
    int8 route: layernorm_quant feeds q/k/v (one shared int8 input, src/nn/linear/linear.cpp:568-583),
    quant_scale_back on the fused qkv product, quant_back_act_mul for the gated FF
    (src/nn/feedforward/feedforward.cpp:163-187), quant_back_element_add_scale for the residual adds."""

    def __init__(self, cfg: ModelConfig, quant: QuantConfig, idx: int):
        self.cfg, self.quant, self.idx = cfg, quant, idx
        self.unfused = None

    def load_state_dict(self, sd, prefix, device):
        c = self.cfg
        hd, kvd = c.num_heads * c.dim_head, c.num_kv_heads * c.dim_head
        self.ln_attn = _dev_t(sd[prefix + ".ln_attn.weight"], device)
        self.ln_ff = _dev_t(sd[prefix + ".ln_ff.weight"], device)
        lin = lambda sub, din, dout: Int8Linear(prefix + "." + sub, din, dout).load_state_dict(sd, prefix + "." + sub, device)  # noqa: E731
        self.qkv = Int8Linear.fuse(prefix + ".attn.project_qkv",
                                   [lin("attn.project_q", c.dim_model, hd), lin("attn.project_k", c.dim_model, kvd),
                                    lin("attn.project_v", c.dim_model, kvd)])
        self.attn_out = lin("attn.attn_out", hd, c.dim_model)
        self.w_in, self.w_gated = lin("ff.w_in", c.dim_model, c.dim_ff), lin("ff.w_gated", c.dim_model, c.dim_ff)
        self.w_out = lin("ff.w_out", c.dim_ff, c.dim_model)

    def init_random(self, device, gen):
        c = self.cfg
        hd, kvd = c.num_heads * c.dim_head, c.num_kv_heads * c.dim_head
        self.ln_attn = (1.0 + 0.05 * torch.randn(c.dim_model, device=device, generator=gen)).to(torch.float16)
        self.ln_ff = (1.0 + 0.05 * torch.randn(c.dim_model, device=device, generator=gen)).to(torch.float16)
        self.qkv = Int8Linear.random("qkv", c.dim_model, hd + 2 * kvd, device, gen)
        self.attn_out = Int8Linear.random("attn_out", hd, c.dim_model, device, gen)
        self.w_in = Int8Linear.random("w_in", c.dim_model, c.dim_ff, device, gen)
        self.w_gated = Int8Linear.random("w_gated", c.dim_model, c.dim_ff, device, gen)
        self.w_out = Int8Linear.random("w_out", c.dim_ff, c.dim_model, device, gen)

    def linears(self):
        return [self.qkv, self.attn_out, self.w_in, self.w_gated, self.w_out]

    def weight_bytes(self):
        return sum(l.nbytes() for l in self.linears())

    # Decode batches (<= 32 rows): the streaming int8 kernel with the scale-back fused in its epilogue
    # (zl_w8a8_gemm_phase: 4 GEMM launches per layer instead of 5 GEMMs + 4 scale-back kernels + split-K memsets;
    # bit-identical results).  More rows: the tiled int8 GEMM + the reference's separate scale-back kernels.
    @staticmethod
    def _stream(rows):
        return rows <= 32 and os.environ.get("ZL_W8_PHASE", "1") != "0"

    def _gated_stream_weight(self):
        if getattr(self, "_w8m_gated", None) is None:
            self._w8m_gated = ops.W8MWeight.from_rows(torch.cat([self.w_in.weight, self.w_gated.weight], dim=0),
                                                      torch.cat([self.w_in.scale, self.w_gated.scale]), row_interleave=True)
        return self._w8m_gated

    def project_qkv(self, hidden, eps, out=None):
        _, xq, sx = ops.layernorm_quant(hidden, self.ln_attn, eps)
        if self._stream(hidden.shape[0]):
            return ops.w8a8_gemm_phase(xq, sx, self.qkv.stream_weight(), ops.W8_BACK, out=out, dtype=hidden.dtype)
        return ops.quant_scale_back(self.qkv.gemm(xq), sx, self.qkv.scale, hidden.dtype, out=out)

    def attn_out_add(self, attn, hidden):
        aq, sa = ops.quant_calc_scale(attn)
        if self._stream(hidden.shape[0]):
            return ops.w8a8_gemm_phase(aq, sa, self.attn_out.stream_weight(), ops.W8_BACK_ADD, addend=hidden, scale=1.0, out=hidden)
        ops.quant_back_element_add_scale(self.attn_out.gemm(aq), sa, self.attn_out.scale, hidden, 1.0, out=hidden)

    def ff_add(self, hidden, eps, act_buf=None):
        _, xq, sx = ops.layernorm_quant(hidden, self.ln_ff, eps)
        if self._stream(hidden.shape[0]):
            act = ops.w8a8_gemm_phase(xq, sx, self._gated_stream_weight(), ops.W8_ACT_SILU, out=act_buf, dtype=hidden.dtype)
            aq, sa = ops.quant_calc_scale(act)
            return ops.w8a8_gemm_phase(aq, sa, self.w_out.stream_weight(), ops.W8_BACK_ADD, addend=hidden, scale=1.0, out=hidden)
        act = ops.quant_back_act_mul(self.w_in.gemm(xq), sx, self.w_in.scale, self.w_gated.gemm(xq), sx, self.w_gated.scale,
                                     "silu", hidden.dtype)
        aq, sa = ops.quant_calc_scale(act)
        ops.quant_back_element_add_scale(self.w_out.gemm(aq), sa, self.w_out.scale, hidden, 1.0, out=hidden)


# decode attention with the split merge inside its launch (zl_decode_attn_la): default of the ZL_ATTN_LA switch
_ATTN_LA_DEFAULT = "auto"


# 9..32 decode rows: RMSNorm deferred into the phase kernel (w4_phase.hip DN) instead of a stand-alone launch: the ZL_DEFER_NORM switch.
# OFF by default -- measured (round 5, profiles/r05_defer_norm.txt): +5 % tokens/s at batch 32, +10 % at batch 16, but every output of the
# qkv and gate|up projections then carries an fp16 rounding the reference does not make (T(x w) rs instead of T(x rs w)), and on the
# synthetic network that is 1e-2 of the largest logit after 4 layers (tests/test_gpu_fullgeom.py [32-4]: 1.35e-2 from R against a
# 6e-3 bar; test_gpu_model batch 20: 1.8e-3 from E against 1e-3).  Parity is the first gate: opt-in only.
_DEFER_NORM_DEFAULT = "0"


def _fused_norm_rows(weight):
    """rows up to which the W4A16 kernel of this weight fuses the RMSNorm: 8 for the phase-pipelined MFMA kernel
    (register-resident activations, K <= 4096; bit-identical to the stand-alone launch), 32 with the deferred norm on
    (ZL_DEFER_NORM=1: rs applied to the fp32 totals, not bit-identical), 4 otherwise"""
    if not (isinstance(weight, ops.W4MWeight) and weight.k <= 4096 and weight.group_size % 128 == 0):
        return 4
    return 32 if os.environ.get("ZL_DEFER_NORM", _DEFER_NORM_DEFAULT) != "0" else 8


class EncoderLayer:
    """nn::EncoderLayer (src/nn/block/block.h:15-63): ln_attn, attn{project_q,k,v,attn_out}, ln_ff,
    ff{w_in,w_gated,w_out}; q/k/v and w_in/w_gated are fused at load (CPM_FUSE_QKV / CPM_FUSE_FF_IN)."""

    def __init__(self, cfg: ModelConfig, quant: QuantConfig, idx: int, tp=None):
        self.cfg, self.quant, self.idx = cfg, quant, idx
        self.ln_attn = self.ln_ff = None
        self.qkv = self.attn_out = self.w_in_gated = self.w_out = None
        self.unfused = None     # act-order checkpoints: [q, k, v, w_in, w_gated] as separate linears
        # tensor parallelism: cfg holds the LOCAL head / ff counts; q/k/v and gate/up are column-parallel,
        # attn_out / w_out row-parallel: their partial outputs are summed over the ranks before the residual add
        self.tp = tp if tp is not None and tp.size > 1 else None

    def load_state_dict(self, sd, prefix, device):
        c, q = self.cfg, self.quant
        hd = c.num_heads * c.dim_head
        kvd = c.num_kv_heads * c.dim_head
        self.ln_attn = _dev_t(sd[prefix + ".ln_attn.weight"], device, torch.float16)
        self.ln_ff = _dev_t(sd[prefix + ".ln_ff.weight"], device, torch.float16)

        tp = self.tp

        # act-order feed-forward: w_in / w_gated OUTPUT columns are stored in w_out's regrouped input order, so the
        # activation reaches w_out already permuted (no gather, and a TP rank's column slice of w_in / w_gated matches its
        # row slice of w_out) -- the reference's permute_ff_up_out (linear.cpp:1168-1210)
        ff_out_perm = None
        g_out = sd.get(prefix + ".ff.w_out.g_idx") if q.act_order else None
        if g_out is not None:
            g64 = _dev_t(g_out, device, torch.int32).to(torch.int64)
            op = torch.argsort(g64, stable=True)
            if not torch.equal(op, torch.arange(op.numel(), device=op.device)):
                ff_out_perm = op.to(torch.int32)

        def lin(sub, din, dout, mode=None, part=None, out_perm=None, prepermuted=False):
            # the checkpoint holds the FULL matrix; a TP rank keeps its column (output rows) or row (input columns) slice
            # (part = (index, count) overrides (rank, size): replicated kv heads)
            idx, cnt = part if part else ((tp.rank, tp.size) if tp else (0, 1))
            full_in, full_out = (din * cnt if mode == "row" else din), (dout * cnt if mode == "column" else dout)
            l = Int4GPTQ(prefix + "." + sub, full_in, full_out, q)
            l.load_state_dict(sd, prefix + "." + sub, device, out_perm=out_perm, input_prepermuted=prepermuted)
            if tp and mode:
                if l.perm is not None and mode == "row":
                    # row-parallel with an input gather left (attn_out): the rank keeps rows [idx K/cnt, ...) of the REGROUPED
                    # order, which are scattered over all heads -- forward all-gathers the attention output first
                    l.tp_gather_perm = l.perm[idx * din:(idx + 1) * din].contiguous()
                    l.perm = None
                l.km = parallel.shard_k_major(*l.km, q.group_size, mode, idx, cnt)
                l.dim_in, l.dim_out = din, dout
                if l.bias is not None and mode == "column":
                    l.bias = l.bias[idx * dout:(idx + 1) * dout].contiguous()
                if l.bias is not None and mode == "row" and tp.rank != 0:
                    l.bias = None                     # added once, by rank 0's partial sum
            return l
        if q.awq and os.environ.get("AWQ_USE_EXLLAMA", "1") == "0":
            # the reference's native AWQ route: every linear on its own, tensors as stored (no q/k/v or gate/up fusion)
            if tp:
                raise ops.ZLError("the native AWQ route is not wired into tensor parallelism (use AWQ_USE_EXLLAMA=1)")

            def alin(sub, din, dout):
                return AWQLinear(prefix + "." + sub, din, dout, q).load_state_dict(sd, prefix + "." + sub, device)
            self.unfused = [alin("attn.project_q", c.dim_model, hd), alin("attn.project_k", c.dim_model, kvd),
                            alin("attn.project_v", c.dim_model, kvd), alin("ff.w_in", c.dim_model, c.dim_ff),
                            alin("ff.w_gated", c.dim_model, c.dim_ff)]
            self.attn_out = alin("attn.attn_out", hd, c.dim_model)
            self.w_out = alin("ff.w_out", c.dim_ff, c.dim_model)
            return
        kv_part = getattr(self, "kv_part", None)
        pq, pk, pv = (lin("attn.project_q", c.dim_model, hd, "column"), lin("attn.project_k", c.dim_model, kvd, "column", kv_part),
                      lin("attn.project_v", c.dim_model, kvd, "column", kv_part))
        w_in = lin("ff.w_in", c.dim_model, c.dim_ff, "column", out_perm=ff_out_perm)
        w_gated = lin("ff.w_gated", c.dim_model, c.dim_ff, "column", out_perm=ff_out_perm)
        self.attn_out = lin("attn.attn_out", hd, c.dim_model, "row").pack()
        self.w_out = lin("ff.w_out", c.dim_ff, c.dim_model, "row", prepermuted=ff_out_perm is not None).pack()
        if not (Int4GPTQ.same_perm([pq, pk, pv]) and Int4GPTQ.same_perm([w_in, w_gated])):
            # act-order with DIFFERENT input orders inside a group (not what GPTQ produces for linears sharing their input,
            # but a legal checkpoint): every linear gathers x itself, q/k/v and gate/up stay separate
            self.unfused = [l.pack() for l in (pq, pk, pv, w_in, w_gated)]
        else:
            self.qkv = Int4GPTQ.fuse(prefix + ".attn.project_qkv", [pq, pk, pv])
            self.w_in_gated = Int4GPTQ.fuse(prefix + ".ff.w_in_gated", [w_in, w_gated], row_interleave=True)

    def init_random(self, device, gen):
        """Synthetic weights of the right shapes, generated directly in the packed layout."""
        c, q = self.cfg, self.quant
        hd, kvd = c.num_heads * c.dim_head, c.num_kv_heads * c.dim_head
        self.ln_attn = (1.0 + 0.05 * torch.randn(c.dim_model, device=device, generator=gen)).to(torch.float16)
        self.ln_ff = (1.0 + 0.05 * torch.randn(c.dim_model, device=device, generator=gen)).to(torch.float16)

        def rnd(name, din, dout, interleave=False):
            l = Int4GPTQ(name, din, dout, q)
            mag = 1.0 / math.sqrt(din) / 4.0
            if w4_algo(q) == "mfma":
                l.weight = ops.W4MWeight.random(dout, din, q.group_size, device, gen, mag, interleave)
            else:
                l.weight = ops.W4Weight.random(dout, din, q.group_size, device, gen, mag, q.sym, interleave)
            return l
        self.qkv = rnd("qkv", c.dim_model, hd + 2 * kvd)
        self.attn_out = rnd("attn_out", hd, c.dim_model)
        self.w_in_gated = rnd("w_in_gated", c.dim_model, 2 * c.dim_ff, True)
        self.w_out = rnd("w_out", c.dim_ff, c.dim_model)

    def linears(self):
        return list(self.unfused) + [self.attn_out, self.w_out] if self.unfused else [self.qkv, self.attn_out, self.w_in_gated, self.w_out]

    def weight_bytes(self):
        return sum(l.weight.nbytes() for l in self.linears())

    # the two fused projections, or their act-order stand-ins (separate RMSNorm, per-linear gather, concat / gate_mul)
    def project_qkv(self, hidden, eps, out=None, normed=None):
        """normed: ln_attn(hidden) already computed by the caller (the dual-stream path's fused add + norm)"""
        if self.unfused is None:
            if self.qkv.perm is not None:                 # act-order: RMSNorm, ONE gather for the fused q|k|v, then the GEMV
                xn = normed if normed is not None else ops.rmsnorm(hidden, self.ln_attn, eps)
                return ops.w4_linear(ops.permute_input(xn, self.qkv.perm), self.qkv.weight, bias=self.qkv.bias, out=out)
            if normed is not None:
                return ops.w4_linear(normed, self.qkv.weight, bias=self.qkv.bias, out=out)
            if hidden.shape[0] > _fused_norm_rows(self.qkv.weight):   # the fused norm prologue normalises every row in every
                xn = ops.rmsnorm(hidden, self.ln_attn, eps)           # workgroup: beyond a few rows a separate launch is cheaper
                return ops.w4_linear(xn, self.qkv.weight, bias=self.qkv.bias, out=out)
            return ops.w4_linear(hidden, self.qkv.weight, bias=self.qkv.bias, out=out, norm_weight=self.ln_attn, norm_eps=eps)
        xn = normed if normed is not None else ops.rmsnorm(hidden, self.ln_attn, eps)
        parts = [l.forward(xn) for l in self.unfused[:3]]
        return torch.cat(parts, dim=1, out=out) if out is not None else torch.cat(parts, dim=1)

    def row_partial(self, lin, x):
        """this rank's partial output of a row-parallel linear.  act-order attn_out: the rank's regrouped rows span all
        heads, so the attention output is all-gathered and read through the rank's slice of the permutation first"""
        gp = getattr(lin, "tp_gather_perm", None)
        if gp is not None:
            x = ops.permute_input(self.tp.all_gather_columns(x), gp)
        return lin.forward(x)

    def _row_parallel_add(self, lin, x, hidden):
        """hidden += sum over TP ranks of lin(x): ModelContext::reduce_sum on the fp16 partial outputs, then the
        residual add in T arithmetic (src/nn/block/block.cpp:123-140, src/model/model_context.cpp:203-242)"""
        part = self.row_partial(lin, x)
        # REDUCE_TP_INT8_THRES (the reference's switch, model_context.cpp:221-227): above that many rows the partial sums travel
        # as group-32 int8 codes (ModelContext::reduce_tp_int8)
        thres = int(os.environ.get("REDUCE_TP_INT8_THRES", "0") or 0)
        compressed = getattr(self.tp, "reduce_tp_int8", None)
        if thres > 0 and compressed is not None and part.shape[0] > thres and part.numel() % (32 * self.tp.size) == 0:
            # the one-shot INT8 exchange takes whole 64-element groups per rank, a contiguous 8-byte-aligned message that fits its
            # buffer; everything else needs the RCCL composition -- and without either transport the fp16 all-reduce below runs
            # (the branch used to be entered with only `oneshot` present and then raised for shapes it does not take)
            one = getattr(self.tp, "oneshot", None)
            fits = (one is not None and part.is_contiguous() and part.data_ptr() % 8 == 0 and part.numel() % (64 * self.tp.size) == 0
                    and part.numel() * 2 <= getattr(self.tp, "oneshot_bytes", 0))
            if fits or getattr(self.tp, "comm", None) is not None:
                ops.element_add_scale(hidden, compressed(part), 1.0, True, out=hidden)
                return
        fused = getattr(self.tp, "all_reduce_add", None)
        if fused is not None and fused(part, hidden) is not None:      # one-shot all-reduce with the residual add in its launch
            return
        self.tp.all_reduce_sum(part)
        ops.element_add_scale(hidden, part, 1.0, True, out=hidden)

    def attn_out_add(self, attn, hidden):
        """hidden += attn_out(attn) in place (linear + element_add_scale fused in the epilogue)"""
        if self.tp:
            return self._row_parallel_add(self.attn_out, attn, hidden)
        self.attn_out.forward(attn, residual=hidden, out=hidden, epilogue=ops.EPI_RESIDUAL)

    def ff_add(self, hidden, eps, act_buf=None):
        """hidden += w_out(silu(w_in(ln(hidden))) * w_gated(ln(hidden))) in place"""
        act = self.ff_in(hidden, eps, out=act_buf)
        if self.tp:
            return self._row_parallel_add(self.w_out, act, hidden)
        self.w_out.forward(act, residual=hidden, out=hidden, epilogue=ops.EPI_RESIDUAL)

    def ff_in(self, hidden, eps, out=None, normed=None):
        if self.unfused is None:
            if self.w_in_gated.perm is not None:
                xn = normed if normed is not None else ops.rmsnorm(hidden, self.ln_ff, eps)
                return ops.w4_linear(ops.permute_input(xn, self.w_in_gated.perm), self.w_in_gated.weight, bias=self.w_in_gated.bias,
                                     out=out, epilogue=ops.EPI_SILU_MUL)
            if normed is not None:
                return ops.w4_linear(normed, self.w_in_gated.weight, bias=self.w_in_gated.bias, out=out, epilogue=ops.EPI_SILU_MUL)
            if hidden.shape[0] > _fused_norm_rows(self.w_in_gated.weight):
                xn = ops.rmsnorm(hidden, self.ln_ff, eps)
                return ops.w4_linear(xn, self.w_in_gated.weight, bias=self.w_in_gated.bias, out=out, epilogue=ops.EPI_SILU_MUL)
            return ops.w4_linear(hidden, self.w_in_gated.weight, bias=self.w_in_gated.bias, out=out, norm_weight=self.ln_ff,
                                 norm_eps=eps, epilogue=ops.EPI_SILU_MUL)
        xn = normed if normed is not None else ops.rmsnorm(hidden, self.ln_ff, eps)
        gate = self.unfused[3].forward(xn, out=out)
        return ops.gate_mul(gate, self.unfused[4].forward(xn), "silu")


@dataclass
class DynBatchContext:
    """Per-step device state of a decode batch (model::DynBatchContext s_token / s_position /
    s_placement / s_len_buf, src/model/dyn_batch_context.h:29-200) + the RagBufferContext tables."""
    tokens: torch.Tensor        # (B) int32
    positions: torch.Tensor     # (B) int32
    placement: torch.Tensor     # (B) int32   slot of the new token in the task's KV buffer
    buf_lens: torch.Tensor      # (B) int32   allocated length of each task's buffer
    valid_lens: torch.Tensor    # (B) int32   visible keys = position + 1 (greedy / sampling decode)
    k_addrs: torch.Tensor       # (num_layers, B) int64 raw pointers
    v_addrs: torch.Tensor
    max_len_buf: int
    kv: List[torch.Tensor] = field(default_factory=list)  # owners: per task (layers, 2, len_buf, Hkv, D)
    # INT8 KV cache (KV_CACHE_DTYPE=int8, src/model/model_context.cpp:61-79): kv holds u8 codes, one fp32
    # scale per (slot, kv head) in kv_scales: per task (layers, 2, len_buf, Hkv)
    kv_quant: bool = False
    ks_addrs: Optional[torch.Tensor] = None
    vs_addrs: Optional[torch.Tensor] = None
    kv_scales: List[torch.Tensor] = field(default_factory=list)
    # host-side bound on the device-side bookkeeping (advance / zl_greedy_advance bump placement with no host round trip):
    # decode steps that still fit the tasks' buffers.  Eager steps raise when it is used up; a caller replaying a captured
    # step bounds its replays itself (the scatter kernels drop a row whose slot lies outside its buffer -- never a stray write)
    steps_left: int = 1 << 60
    unquant_kv: Dict[int, torch.Tensor] = field(default_factory=dict)   # INT8 cache: a chunked prompt's temporary fp16 K/V


class LLaMA:
    """model::LLaMA (src/model/llama.cpp:11-165) restricted to the dynamic-batch decode step."""

    def __init__(self, cfg: ModelConfig, quant: QuantConfig, device="cuda:0", tp=None):
        """tp: parallel.TPGroup (rank, size, process group) for tensor parallelism over the GPUs of a node -- W4 route
        only; heads, kv heads, dim_ff and the vocabulary must divide by the TP degree."""
        if cfg.scale_depth > 0 and quant.quant_type != 0:
            raise ops.ZLError("scale_depth (MiniCPM residual scaling) is only wired into the unquantised layer stack")
        if quant.quant_type != 0 and cfg.torch_dtype != torch.float16:
            raise ops.ZLError("the W4A16 / int8 routes are fp16 (q_gemm_k_major.cu:989 asserts half)")
        self.cfg, self.quant, self.device = cfg, quant, torch.device(device)
        # QuantType (zhilight/quant.py:8-19): 0 NoQuant, 2 AutoInt8, 5 GPTQ (AWQ-as-exllama rides on 5)
        layer_cls = {0: DenseEncoderLayer, 2: Int8EncoderLayer}.get(quant.quant_type, EncoderLayer)
        self.tp = tp if tp is not None and tp.size > 1 else None
        self.full_cfg = cfg
        if self.tp:
            if layer_cls is not EncoderLayer:
                raise ops.ZLError("tensor parallelism is wired into the W4A16 layer stack only")
            t = self.tp.size
            # fewer kv heads than ranks: with ATTN_KV_REP_TP=1 the reference gives rank r kv head r / (TP / Hkv), i.e.
            # groups of TP / Hkv ranks hold (and cache) the same kv head (src/nn/attention/attention.cpp:126-134)
            kv_part = None
            if (cfg.num_kv_heads < t and t % cfg.num_kv_heads == 0 and int(os.environ.get("ATTN_KV_REP_TP", "0")) > 0):
                kv_part = (self.tp.rank // (t // cfg.num_kv_heads), cfg.num_kv_heads)
            if cfg.num_heads % t or (kv_part is None and cfg.num_kv_heads % t) or cfg.dim_ff % t or cfg.vocab_size % t:
                raise ops.ZLError("heads, kv heads, dim_ff and vocab_size must be divisible by the TP degree "
                                  "(fewer kv heads than ranks: ATTN_KV_REP_TP=1)")
            import dataclasses
            cfg = dataclasses.replace(cfg, num_heads=cfg.num_heads // t, num_kv_heads=1 if kv_part else cfg.num_kv_heads // t,
                                      dim_ff=cfg.dim_ff // t)
            self.cfg = cfg                            # the LOCAL geometry drives buffers, KV and kernels
            self.layers = [EncoderLayer(cfg, quant, i, self.tp) for i in range(cfg.num_layers)]
            for l in self.layers:
                l.kv_part = kv_part
        else:
            self.layers = [layer_cls(cfg, quant, i) for i in range(cfg.num_layers)]
        self.token_embedding = self.output_layernorm = self.lm_head = None
        self._bufs = {}

    # ---- weights -------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], hf_names=True):
        sd = {hf_name_to_internal(k) if hf_names else k: v for k, v in state_dict.items()}
        dev = self.device
        dt = self.cfg.torch_dtype
        self.token_embedding = _dev_t(sd["llama.token_embedding.weight"], dev).to(dt).contiguous()
        self.output_layernorm = _dev_t(sd["llama.output_layernorm.weight"], dev).to(dt).contiguous()
        self.lm_head = self.token_embedding if self.cfg.tie_lm_head else _dev_t(sd["llama.lm_head.weight"], dev).to(dt).contiguous()
        if self.tp:                                   # vocab-parallel projection (embedding.cu:353-385)
            v = self.cfg.vocab_size // self.tp.size
            self.lm_head = self.lm_head[self.tp.rank * v:(self.tp.rank + 1) * v].contiguous()
        for i, layer in enumerate(self.layers):
            layer.load_state_dict(sd, f"llama.layers.{i}", dev)
            layer.q_norm = layer.k_norm = None
            if self.cfg.qk_norm:
                layer.q_norm, layer.k_norm = (self._qk_norm_weight(sd[f"llama.layers.{i}.attn.{n}_norm.weight"], n)
                                              for n in ("q", "k"))
        return self

    def _qk_norm_weight(self, w, which):
        """q_norm / k_norm weights (attention.cpp:110-117): (dim_head) for "head"; (heads * dim_head) for "multi_head", of
        which a TP rank keeps its heads' rows (replicated kv heads: the one head it holds)."""
        c, fc = self.cfg, self.full_cfg
        t = _dev_t(w, self.device).to(c.torch_dtype).reshape(-1)
        if c.qk_norm == "head":
            if t.numel() != c.dim_head:
                raise ops.ZLError("q_norm / k_norm weight must have dim_head elements")
            return t.contiguous()
        full = fc.num_heads if which == "q" else fc.num_kv_heads
        loc = c.num_heads if which == "q" else c.num_kv_heads
        if t.numel() != full * c.dim_head:
            raise ops.ZLError("q_norm / k_norm weight must have heads * dim_head elements")
        if self.tp:
            kv_part = getattr(self.layers[0], "kv_part", None)
            first = kv_part[0] if (which == "k" and kv_part) else self.tp.rank * loc
            t = t.view(full, c.dim_head)[first:first + loc]
        return t.contiguous()

    def apply_qk_norm(self, layer, qkv):
        """q_norm / k_norm in place on the q and k windows of the fused projection, before the rotation
        (attention.cpp:864-876 fused, :904-915 separate projections)."""
        c = self.cfg
        if not c.qk_norm:
            return
        hd, kvd = c.num_heads * c.dim_head, c.num_kv_heads * c.dim_head
        mode = 0 if c.qk_norm == "head" else 1
        ops.head_norm(qkv[:, :hd], layer.q_norm, c.num_heads, c.dim_head, c.eps, mode, out=qkv[:, :hd])
        ops.head_norm(qkv[:, hd:hd + kvd], layer.k_norm, c.num_kv_heads, c.dim_head, c.eps, mode, out=qkv[:, hd:hd + kvd])

    def init_random(self, seed=0):
        gen = torch.Generator(device=self.device).manual_seed(seed)
        c, dev = self.cfg, self.device
        dt = c.torch_dtype
        self.token_embedding = (torch.randn(c.vocab_size, c.dim_model, device=dev, generator=gen) * 0.5).to(dt)
        self.output_layernorm = torch.ones(c.dim_model, dtype=dt, device=dev)
        vloc = c.vocab_size // self.tp.size if self.tp else c.vocab_size
        self.lm_head = self.token_embedding if (c.tie_lm_head and not self.tp) else (torch.randn(vloc, c.dim_model, device=dev, generator=gen) * 0.02).to(dt)
        for layer in self.layers:
            layer.init_random(dev, gen)
            layer.q_norm = layer.k_norm = None
            if c.qk_norm:
                nq, nk = (c.dim_head, c.dim_head) if c.qk_norm == "head" else (c.num_heads * c.dim_head, c.num_kv_heads * c.dim_head)
                layer.q_norm = (1.0 + 0.1 * torch.randn(nq, device=dev, generator=gen)).to(dt)
                layer.k_norm = (1.0 + 0.1 * torch.randn(nk, device=dev, generator=gen)).to(dt)
        return self

    def init_synthetic(self, seed=0, sink=None):
        """Synthetic GPTQ checkpoint of this geometry with SURVEY 8(d)'s recipe -- the one tests/synth.py::gptq_hf and
        tests/test_gpu_model.py::_hf_state draw on the host -- generated on the device and taken through the REAL load path
        (EncoderLayer.load_state_dict: shuffle, zero + 1, transposes, fusion, ZLW4M packing), one layer at a time:
          qweight (K/8, N) int32 from uniform nibbles 0..15; qzeros (K/G, N/8) int32 from nibbles 0..14 stored as zero - 1;
          scales (K/G, N) fp16 = |N(0,1)| * (0.5 / sqrt(K)) / 4 + 1e-4 (activations stay O(1): _hf_state's magnitude, = 8(d)'s
          0.02 / 8 at K = 4096 within 25 %); norm weights 1 + 0.1 N(0,1); embedding uniform integers / 128, lm_head uniform
          integers * 0.05 / 64 (tests/test_gpu_fullgeom.py::_state).
        W4 route, no tensor parallelism (bench.py's TP leg keeps init_random: every rank would have to draw the full matrices).
        sink(name -> device tensor): called with every checkpoint tensor under the reference's parameter names ("llama.layers.3.attn.
        project_q.qweight", "llama.token_embedding.weight", ...) before it is consumed -- tools/bench_boundary.py hands the same
        checkpoint to the reference's own model::LLaMA."""
        c, dev, q = self.cfg, self.device, self.quant
        if self.tp or not all(isinstance(l, EncoderLayer) for l in self.layers) or q.act_order or q.awq:
            raise ops.ZLError("init_synthetic: the plain GPTQ layer stack")
        gen = torch.Generator(device=dev).manual_seed(seed)
        g = q.group_size
        i64 = dict(dtype=torch.int64, device=dev, generator=gen)

        def pack8(nib, dim):                            # 8 nibbles along `dim` (stride 8) -> one int32 word, nibble j at bits 4 j
            word = torch.zeros_like(nib.select(dim, 0).unsqueeze(dim).expand(*[s // 8 if i == dim else s for i, s in enumerate(nib.shape)])).contiguous()
            for j in range(8):
                word |= (nib[j::8] if dim == 0 else nib[:, j::8]) << (4 * j)
            return torch.where(word >= 2 ** 31, word - 2 ** 32, word).to(torch.int32)

        def lin(sd, name, din, dout):
            sd[name + ".qweight"] = pack8(torch.randint(0, 16, (din, dout), **i64), 0)
            sd[name + ".qzeros"] = pack8(torch.randint(0, 15, (din // g, dout), **i64), 1)
            sd[name + ".scales"] = (torch.randn(din // g, dout, device=dev, generator=gen).abs() * ((0.5 / math.sqrt(din)) / 4) + 1e-4).to(torch.float16)
        hd, kvd = c.num_heads * c.dim_head, c.num_kv_heads * c.dim_head
        dt = c.torch_dtype
        for i, layer in enumerate(self.layers):
            pfx = f"llama.layers.{i}"
            sd = {pfx + ".ln_attn.weight": (1 + 0.1 * torch.randn(c.dim_model, device=dev, generator=gen)).to(torch.float16),
                  pfx + ".ln_ff.weight": (1 + 0.1 * torch.randn(c.dim_model, device=dev, generator=gen)).to(torch.float16)}
            lin(sd, pfx + ".attn.project_q", c.dim_model, hd)
            lin(sd, pfx + ".attn.project_k", c.dim_model, kvd)
            lin(sd, pfx + ".attn.project_v", c.dim_model, kvd)
            lin(sd, pfx + ".attn.attn_out", hd, c.dim_model)
            lin(sd, pfx + ".ff.w_in", c.dim_model, c.dim_ff)
            lin(sd, pfx + ".ff.w_gated", c.dim_model, c.dim_ff)
            lin(sd, pfx + ".ff.w_out", c.dim_ff, c.dim_model)
            if sink is not None:
                sink(sd)
            layer.load_state_dict(sd, pfx, dev)
            layer.q_norm = layer.k_norm = None
            del sd
        u8 = lambda: torch.randint(-127, 128, (c.vocab_size, c.dim_model), dtype=torch.int8, device=dev, generator=gen)
        self.token_embedding = (u8().to(dt) * (1.0 / 128)).contiguous()
        self.output_layernorm = (1 + 0.1 * torch.randn(c.dim_model, device=dev, generator=gen)).to(dt)
        self.lm_head = self.token_embedding if c.tie_lm_head else (u8().to(dt) * (0.05 / 64)).contiguous()
        if sink is not None:
            sink({"llama.token_embedding.weight": self.token_embedding, "llama.output_layernorm.weight": self.output_layernorm,
                  "llama.lm_head.weight": self.lm_head})
        return self

    # ---- KV state ------------------------------------------------------------------------------
    def new_context(self, batch: int, len_buf: int, start_pos: int, fill_random=False, kv_cache_dtype=None) -> DynBatchContext:
        """`batch` tasks whose first `start_pos` tokens are already in the KV buffers (zero- or
        random-filled here: the reference zero-fills, src/kvcache/transformer_buffer.cu:290-330).
        kv_cache_dtype: None (the model dtype) or "int8"; default from the reference's switch, the environment
        variable KV_CACHE_DTYPE (src/model/model_context.cpp:61-79)."""
        c, dev = self.cfg, self.device
        if kv_cache_dtype is None:
            kv_cache_dtype = os.environ.get("KV_CACHE_DTYPE") or None
        if kv_cache_dtype not in (None, "int8"):
            raise ops.ZLError(f"Unsupported dtype: {kv_cache_dtype}")
        kv, kv_scales = [], []
        for _ in range(batch):
            shape = (c.num_layers, 2, len_buf, c.num_kv_heads, c.dim_head)
            if kv_cache_dtype == "int8":
                t = torch.randint(0, 256, shape, dtype=torch.uint8, device=dev) if fill_random else \
                    torch.full(shape, 128, dtype=torch.uint8, device=dev)
                sc = torch.rand(shape[:-1], dtype=torch.float32, device=dev) * 0.03 + 0.005 if fill_random else \
                    torch.zeros(shape[:-1], dtype=torch.float32, device=dev)
                kv_scales.append(sc)
            else:
                t = torch.randn(shape, dtype=c.torch_dtype, device=dev) if fill_random else torch.zeros(shape, dtype=c.torch_dtype, device=dev)
            kv.append(t)
        quant_kw = {}
        if kv_cache_dtype == "int8":
            quant_kw = dict(
                kv_quant=True, kv_scales=kv_scales,
                ks_addrs=torch.tensor([[t[l, 0].data_ptr() for t in kv_scales] for l in range(c.num_layers)], dtype=torch.int64, device=dev),
                vs_addrs=torch.tensor([[t[l, 1].data_ptr() for t in kv_scales] for l in range(c.num_layers)], dtype=torch.int64, device=dev))
        k_addrs = torch.tensor([[t[l, 0].data_ptr() for t in kv] for l in range(c.num_layers)], dtype=torch.int64, device=dev)
        v_addrs = torch.tensor([[t[l, 1].data_ptr() for t in kv] for l in range(c.num_layers)], dtype=torch.int64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        return DynBatchContext(
            steps_left=len_buf - start_pos,
            tokens=torch.zeros(batch, **i32), positions=torch.full((batch,), start_pos, **i32),
            placement=torch.full((batch,), start_pos, **i32), buf_lens=torch.full((batch,), len_buf, **i32),
            valid_lens=torch.full((batch,), start_pos + 1, **i32), k_addrs=k_addrs, v_addrs=v_addrs,
            max_len_buf=len_buf, kv=kv, **quant_kw)

    def _buffers(self, b):
        if b not in self._bufs:
            c, dev = self.cfg, self.device
            f16 = dict(dtype=c.torch_dtype, device=dev)
            hd, kvd = c.num_heads * c.dim_head, c.num_kv_heads * c.dim_head
            self._bufs[b] = dict(
                hidden=torch.empty(b, c.dim_model, **f16), qkv=torch.empty(b, hd + 2 * kvd, **f16),
                q=torch.empty(b, hd, **f16), attn=torch.empty(b, hd, **f16), act=torch.empty(b, c.dim_ff, **f16),
                logits=torch.empty(b, c.vocab_size, **f16))
        return self._bufs[b]

    # ---- one decode step -----------------------------------------------------------------------
    def encode(self, ctx: DynBatchContext, workspace=None, argmax_ws=None, gemv_only=False, skip_gemv=False):
        """LLaMA::encode for a pure decode ("search") batch: returns logits (B, vocab) fp16.
        gemv_only (bench.py's roofline leg): issue ONLY the four quantised projections of every layer, exactly as the step
        launches them (fused norm / rotary + scatter / split merge / gated activation / residual), on whatever the buffers
        hold -- no embedding, attention or lm_head; returns None.
        skip_gemv (bench.py again): the complement -- everything BUT those projections (embedding, rope table, attention, lm_head,
        greedy bookkeeping), so that step time - this = the projections' time inside the step."""
        c = self.cfg
        b = ctx.tokens.numel()
        if ctx.steps_left <= 0 and not torch.cuda.is_current_stream_capturing():
            raise ops.ZLError("decode step past the end of the KV buffers (placement >= len_buf): grow the buffers first "
                              "(the reference asserts pos_buf < len_buf, ragged_buffer_kernel.cu:194-222)")
        bufs = self._buffers(b)
        if workspace is None:
            workspace = self._bufs.setdefault(("ws", b, ctx.max_len_buf),
                                              ops.decode_attn_workspace(b, 1, c.num_heads, c.dim_head, ctx.max_len_buf, self.device))
        rs = c.rope_scaling
        kind = rs.get("rope_type", rs.get("type")) if rs else None
        if gemv_only:
            hidden = bufs["hidden"]
            cos, sin = self._rope_tables(ctx.positions)
        elif kind in (None, "default", "llama3") and c.dim_head <= 256 and self.token_embedding.shape[0] == c.vocab_size:
            # token_embedding + RopePreparer in one launch (two independent ~4 us kernels, half of each a kernel boundary)
            l3 = None if kind != "llama3" else (rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"], rs["original_max_position_embeddings"])
            hidden, cos, sin = ops.embedding_rope(ctx.tokens, self.token_embedding, c.scale_emb, ctx.positions, c.dim_head, c.rope_theta, True, l3)
        else:
            hidden = ops.embedding(ctx.tokens, self.token_embedding, c.scale_emb)  # token_embedding
            cos, sin = self._rope_tables(ctx.positions)                            # RopePreparer
        scale = 1.0 / math.sqrt(c.dim_head)
        mfma_attn = (c.dim_head == 128 and c.num_heads // c.num_kv_heads <= 16
                     and os.environ.get("ZL_ATTN_MFMA", "1") != "0")
        # fused qkv projection + rotary + KV scatter in the GEMV epilogue (zl_w4a16_qkv_rope_scatter) where it applies
        fuse_qkv_rope = (mfma_attn and not ctx.kv_quant and os.environ.get("ZL_FUSE_QKV_ROPE", "1") != "0"
                         and not c.qk_norm
                         and all(isinstance(l, EncoderLayer) and l.unfused is None and l.qkv.perm is None
                                 and isinstance(l.qkv.weight, ops.W4MWeight)
                                 for l in self.layers)
                         and ops.w4_qkv_rope_scatter_ok(b, c.dim_model, c.dim_head, norm=b <= max(8, _fused_norm_rows(self.layers[0].qkv.weight))))
        fuse_qkv_rope_i8 = (mfma_attn and not ctx.kv_quant and b <= 32 and c.dim_head % 32 == 0 and not c.qk_norm
                            and os.environ.get("ZL_FUSE_QKV_ROPE", "1") != "0"
                            and all(isinstance(l, Int8EncoderLayer) and l._stream(b) for l in self.layers))
        # a few rows: the split merge of the decode attention rides in the attn_out projection's prologue (one launch less)
        merge_plan = None
        if (fuse_qkv_rope and not self.tp and os.environ.get("ZL_ATTN_MERGE", "1") != "0"
                and all(l.attn_out.perm is None for l in self.layers)):
            merge_plan = ops.attn_merge_plan(b, c.num_heads, c.num_kv_heads, c.dim_head, ctx.max_len_buf,
                                             self.layers[0].attn_out.weight, c.torch_dtype)
        # round 5: the split merge INSIDE the attention launch (last-arriving workgroup of a (task, kv head) pair, zl_decode_attn_la)
        # wherever the step would otherwise run the two-launch path (split kernel + merge kernel): same records, same merge
        # arithmetic, one launch less per layer.  Batch 1 keeps the merge in attn_out's prologue: there the in-launch merge was
        # measured 1.4 us per layer SLOWER at the same split length and worse with finer splits (profiles/r05_attn_la_ab.txt).
        # ZL_ATTN_LA: auto (default) / 0 off / 1 everywhere; ZL_ATTN_LA_SPLIT = keys per split (multiple of 32, 0 = the two-launch
        # path's), ZL_ATTN_LA_HALF = 1 half-precision split records
        la = None
        la_mode = os.environ.get("ZL_ATTN_LA", _ATTN_LA_DEFAULT)
        if (mfma_attn and not ctx.kv_quant and c.dim_head == 128 and (la_mode == "1" or (la_mode == "auto" and merge_plan is None))):
            la_split = int(os.environ.get("ZL_ATTN_LA_SPLIT", "0") or 0)
            la_half = os.environ.get("ZL_ATTN_LA_HALF", "0") == "1" and c.torch_dtype == torch.float16
            eff = la_split or ops.decode_attn_la_split_len(b, c.num_kv_heads, ctx.max_len_buf)
            if (ctx.max_len_buf + eff - 1) // eff <= 64:
                # one block per (batch, power-of-two bucket of the buffer length): records are addressed by the launch's own split
                # count, so a block sized for a longer buffer serves every shorter one; a captured graph keeps its block's address,
                # hence buckets instead of regrowing in place -- at most log2 blocks per batch, < 2 x the largest
                bucket = 1 << max(10, (ctx.max_len_buf - 1).bit_length())
                key = ("la_ws", b, bucket)
                if key not in self._bufs:
                    self._bufs[key] = ops.decode_attn_la_workspace(b, c.num_heads, c.num_kv_heads, bucket, self.device)
                la = (self._bufs[key], la_split, la_half)
                merge_plan = None

        # round 6: the rows' statistics travel with the residual stream (zl_w4_opts_t::row_ss / row_ss_out): attn_out and w_out leave
        # per row and 16-column tile the sum of squares of what they store, the normalising projections (qkv, gate|up) read those
        # 256 numbers per row instead of running -- or waiting for -- an RMSNorm pass over the rows: no stand-alone norm launch from
        # 9 rows on, no register-resident norm prologue below.  ZL_ROW_SS=0 off; ZL_ROW_SS_MIN_M: rows it starts at
        stats = None
        if (fuse_qkv_rope and la is not None and not self.tp and os.environ.get("ZL_ROW_SS", "1") != "0"
                and int(os.environ.get("ZL_ROW_SS_MIN_M", "8")) <= b <= 32 and c.dim_model % 1024 == 0
                and all(l.attn_out.perm is None and l.w_out.perm is None and l.w_in_gated.perm is None for l in self.layers)
                and ops.w4_row_ss_routes(b, [self.layers[0].attn_out.weight, self.layers[0].w_out.weight],
                                         [(self.layers[0].qkv.weight, True), (self.layers[0].w_in_gated.weight, False)], self.device)):
            key = ("row_ss", b)
            if key not in self._bufs:
                self._bufs[key] = torch.empty((2, b, c.dim_model // 16), dtype=torch.float32, device=self.device)
            stats = self._bufs[key]

        def attend(li, out):
            """decode attention of layer li over the tasks' buffers: bufs["q"] -> out (B, H * D)"""
            q4 = bufs["q"].view(b, 1, c.num_heads, c.dim_head)
            if la is not None:
                return ops.decode_attention_la(q4, ctx.buf_lens, ctx.k_addrs[li], ctx.v_addrs[li], ctx.valid_lens, scale, ctx.max_len_buf,
                                               c.num_kv_heads, la[0], out=out.view(b, 1, c.num_heads, c.dim_head), split_len=la[1], half=la[2])
            return ops.multi_query_attention_rag_buffer(q4, ctx.buf_lens, ctx.k_addrs[li], ctx.v_addrs[li], None, scale, ctx.max_len_buf,
                                                        c.num_kv_heads, valid_lens=ctx.valid_lens, out=out.view(b, 1, c.num_heads, c.dim_head),
                                                        workspace=workspace)
        # attention split merge + attn_out + residual and ln_ff + gate|up + silu.mul in ONE launch (w4_engine.hip)
        fuse_o_ff = (merge_plan is not None and merge_plan[2] and os.environ.get("ZL_FUSE_O_GATEUP", "0") == "1" and ops.experimental_build()
                     and all(l.w_in_gated.perm is None and isinstance(l.w_in_gated.weight, ops.W4MWeight) for l in self.layers))
        if fuse_o_ff:
            ops.engine_epoch_advance(self.device)
        trace = getattr(self, "trace_hidden", None)          # parity tests: the hidden rows entering every layer (eager runs only)
        for li, layer in enumerate(self.layers):
            if trace is not None and not torch.cuda.is_current_stream_capturing():
                trace.append(hidden.clone())
            if fuse_qkv_rope_i8:
                # INT8 route: layernorm_quant, then the streaming W8A8 kernel with scale-back + rotary + KV scatter fused
                _, xq, sx = ops.layernorm_quant(hidden, layer.ln_attn, c.eps)
                ops.w8a8_qkv_rope_scatter(xq, sx, layer.qkv.stream_weight(), cos, sin, ctx.placement, ctx.buf_lens, ctx.k_addrs[li],
                                          ctx.v_addrs[li], c.num_heads, c.num_kv_heads, c.dim_head, q_out=bufs["q"])
                attend(li, bufs["attn"])
                layer.attn_out_add(bufs["attn"], hidden)
                layer.ff_add(hidden, c.eps, bufs["act"])
                continue
            if fuse_qkv_rope and skip_gemv and la is not None:
                attend(li, bufs["attn"])
                continue
            if fuse_qkv_rope and skip_gemv:
                if not merge_plan:
                    raise ops.ZLError("skip_gemv: the fused W4 decode route with the merging attn_out projection only")
                ops.decode_attention_splits(bufs["q"].view(b, c.num_heads, c.dim_head), ctx.buf_lens, ctx.k_addrs[li],
                                            ctx.v_addrs[li], ctx.valid_lens, scale, ctx.max_len_buf, c.num_kv_heads, workspace)
                continue
            if fuse_qkv_rope and stats is not None:
                if li == 0:
                    ops.row_ss(hidden, out=stats[0])          # (the embedding's rows: the only statistics no projection produced)
                ops.w4_qkv_rope_scatter(hidden, layer.qkv.weight, cos, sin, ctx.placement, ctx.buf_lens, ctx.k_addrs[li],
                                        ctx.v_addrs[li], c.num_heads, c.num_kv_heads, c.dim_head, bias=layer.qkv.bias,
                                        norm_weight=layer.ln_attn, norm_eps=c.eps, q_out=bufs["q"], row_ss=stats[0])
                if not gemv_only:
                    attend(li, bufs["attn"])
                layer.attn_out.forward(bufs["attn"], residual=hidden, out=hidden, epilogue=ops.EPI_RESIDUAL, row_ss_out=stats[1])
                ops.w4_linear(hidden, layer.w_in_gated.weight, bias=layer.w_in_gated.bias, out=bufs["act"], norm_weight=layer.ln_ff,
                              norm_eps=c.eps, epilogue=ops.EPI_SILU_MUL, row_ss=stats[1])
                layer.w_out.forward(bufs["act"], residual=hidden, out=hidden, epilogue=ops.EPI_RESIDUAL, row_ss_out=stats[0])
                continue
            if fuse_qkv_rope:
                fused_norm = b <= max(8, _fused_norm_rows(layer.qkv.weight))    # 9..32 rows: the deferred norm (ZL_DEFER_NORM)
                xin = hidden if fused_norm else ops.rmsnorm(hidden, layer.ln_attn, c.eps)
                ops.w4_qkv_rope_scatter(xin, layer.qkv.weight, cos, sin, ctx.placement, ctx.buf_lens, ctx.k_addrs[li],
                                        ctx.v_addrs[li], c.num_heads, c.num_kv_heads, c.dim_head, bias=layer.qkv.bias,
                                        norm_weight=layer.ln_attn if fused_norm else None, norm_eps=c.eps, q_out=bufs["q"])
                if merge_plan:
                    if not gemv_only:
                        ops.decode_attention_splits(bufs["q"].view(b, c.num_heads, c.dim_head), ctx.buf_lens, ctx.k_addrs[li],
                                                    ctx.v_addrs[li], ctx.valid_lens, scale, ctx.max_len_buf, c.num_kv_heads, workspace)
                    if fuse_o_ff and ops.w4_attn_out_gate_up(workspace, ctx.buf_lens, ctx.valid_lens, merge_plan, b, layer.attn_out.weight,
                                                             hidden, layer.w_in_gated.weight, layer.ln_ff, c.eps, bufs["act"], li,
                                                             bias_o=layer.attn_out.bias, bias_ff=layer.w_in_gated.bias):
                        layer.w_out.forward(bufs["act"], residual=hidden, out=hidden, epilogue=ops.EPI_RESIDUAL)
                        continue
                    ops.w4_attn_out_merge(workspace, ctx.buf_lens, ctx.valid_lens, merge_plan, b, layer.attn_out.weight,
                                          bias=layer.attn_out.bias, residual=hidden, out=hidden, epilogue=ops.EPI_RESIDUAL)
                    layer.ff_add(hidden, c.eps, bufs["act"])
                    continue
                if not gemv_only:
                    attend(li, bufs["attn"])
                layer.attn_out_add(bufs["attn"], hidden)
                layer.ff_add(hidden, c.eps, bufs["act"])
                continue
            if gemv_only:
                raise ops.ZLError("gemv_only: the fused W4 decode route only")
            layer.project_qkv(hidden, c.eps, out=bufs["qkv"])
            self.apply_qk_norm(layer, bufs["qkv"])
            if ctx.kv_quant:
                # attention.cpp:652-676 + :725-745: rotate, quantise the new K/V rows into the u8 cache, attend over codes
                ops.rope_quant_scatter_decode(cos, sin, bufs["qkv"], ctx.placement, ctx.buf_lens, ctx.k_addrs[li], ctx.v_addrs[li],
                                              ctx.ks_addrs[li], ctx.vs_addrs[li], c.num_heads, c.num_kv_heads, c.dim_head,
                                              q_out=bufs["q"])
                ops.multi_query_attention_rag_buffer_quant(
                    bufs["q"].view(b, 1, c.num_heads, c.dim_head), ctx.buf_lens, ctx.k_addrs[li], ctx.v_addrs[li], ctx.ks_addrs[li],
                    ctx.vs_addrs[li], None, scale, ctx.max_len_buf, c.num_kv_heads, valid_lens=ctx.valid_lens,
                    out=bufs["attn"].view(b, 1, c.num_heads, c.dim_head), workspace=workspace)
            elif mfma_attn:
                # rope + KV scatter in one small launch, then the matrix-core decode attention (attention.hip:
                # k_decode_attn_mfma; 11.0 / 18.7 / 39.8 us vs 11.9 / 24.0 / 49.7 us for the fused VALU kernel at batch 1 / 8 / 32)
                ops.rope_scatter_decode(cos, sin, bufs["qkv"], ctx.placement, ctx.buf_lens, ctx.k_addrs[li], ctx.v_addrs[li],
                                        c.num_heads, c.num_kv_heads, c.dim_head, q_out=bufs["q"])
                attend(li, bufs["attn"])
            else:
                ops.decode_attention_fused(cos, sin, bufs["qkv"], ctx.placement, ctx.buf_lens, ctx.valid_lens, ctx.k_addrs[li],
                                           ctx.v_addrs[li], c.num_heads, c.num_kv_heads, c.dim_head, scale, ctx.max_len_buf,
                                           out=bufs["attn"], workspace=workspace)
            layer.attn_out_add(bufs["attn"], hidden)
            layer.ff_add(hidden, c.eps, bufs["act"])
        if gemv_only:
            return None
        self.last_hidden = hidden                      # (B, dim_model) before the output norm: the parity tests read it
        return self._logits(hidden, bufs["logits"], argmax_ws)

    def _logits(self, hidden, out=None, argmax_ws=None):
        """output_layernorm + lm_head (get_logits, src/model/llama.cpp:150-165).  MiniCPM divides the normalised
        row by dim_model / dim_model_base inside the norm (ln_after_enc's scale, llama.cpp:13-21)."""
        c = self.cfg
        rows = hidden.shape[0]
        if self.tp:                                   # local vocabulary slice, then all-gather of the logits
            part = ops.gemm_nt_small_m(hidden, self.lm_head, norm_weight=self.output_layernorm, norm_eps=c.eps) if rows <= 4 else \
                ops.gemm_nt(ops.rmsnorm(hidden, self.output_layernorm, c.eps), self.lm_head)
            full = self.tp.all_gather_columns(part)
            if out is not None:
                out.copy_(full)
                return out
            return full
        ln_scale = (c.dim_model / c.dim_model_base) if c.dim_model_base > 0 else 1.0
        if (rows > 4 and argmax_ws is None and c.dim_model % 128 == 0) or ln_scale != 1.0:
            xn = ops.rmsnorm(hidden, self.output_layernorm, c.eps, ln_scale)
            if rows > 4 and argmax_ws is None and c.dim_model % 128 == 0:
                # more rows than the streaming GEMV takes per pass: the MFMA GEMM; up to 32 rows (decode batches) on the ZLD16M copy of
                # the matrix -- packed on first use, OUTSIDE stream capture (run a step eagerly before capturing it, as bench.py does)
                if rows <= 32 and os.environ.get("ZL_LM_HEAD_PACKED", "1") != "0":
                    if getattr(self, "_lm_head_m", None) is None and not torch.cuda.is_current_stream_capturing():
                        self._lm_head_m = ops.DenseMWeight(self.lm_head)
                    if getattr(self, "_lm_head_m", None) is not None:
                        return ops.gemm_nt_packed(xn, self._lm_head_m, out=out)
                return ops.gemm_nt(xn, self.lm_head, out=out)
            return ops.gemm_nt_small_m(xn, self.lm_head, out=out, argmax_ws=argmax_ws)
        return ops.gemm_nt_small_m(hidden, self.lm_head, out=out, norm_weight=self.output_layernorm, norm_eps=c.eps,
                                   argmax_ws=argmax_ws)

    def _rope_tables(self, pos):
        """cos / sin (rows, dim_head) fp32 of one forward's positions: RopePreparer (rope_preparer.cu:49-160) for plain and
        llama3 frequencies, the RotaryEmbedding variants "dynamic" (NTK) and "yarn" (rotary_embedding.cu:19-61, 398-553)
        as tables for the same fused rotation kernels."""
        c = self.cfg
        rs = c.rope_scaling
        kind = rs.get("rope_type", rs.get("type")) if rs else None
        if kind in (None, "default"):
            return ops.rope_cos_sin(pos, c.dim_head, c.rope_theta, True, None)
        if kind == "llama3":
            return ops.rope_cos_sin(pos, c.dim_head, c.rope_theta, True,
                                    (rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"], rs["original_max_position_embeddings"]))
        if kind == "dynamic":
            if c.max_position_embeddings <= 0:
                raise ops.ZLError("dynamic rope scaling needs max_position_embeddings")
            # the reference reads ONE sequence length per forward: the position of the call's last row (rotary_embedding.cu:36,
            # gridDim.x - 1 of the flat (tokens, dim) view) -- for a decode batch that is the last task's position
            seq_len = pos[-1:].expand(pos.numel()).contiguous()
            return ops.rope_cos_sin_dynamic(pos, c.dim_head, c.rope_theta, rs["factor"], c.max_position_embeddings, seq_len)
        if kind == "yarn":
            deepseek = c.model_type in ("deepseek_v2", "deepseek_v3")
            low, high, msc = ops.yarn_params(c.rope_theta, c.dim_head, rs["original_max_position_embeddings"], rs["factor"],
                                             int(rs.get("beta_fast", 32)), int(rs.get("beta_slow", 1)),
                                             rs.get("attn_factor", 1.0), deepseek, rs.get("mscale", 0.0),
                                             rs.get("mscale_all_dim", 0.0))
            return ops.rope_cos_sin_yarn(pos, c.dim_head, c.rope_theta, rs["factor"], low, high, msc)
        raise ops.ZLError(f"rope_scaling type {kind!r} is not supported (supported: llama3, dynamic, yarn)")

    def _prefill_mask(self, s, len_buf, pos0=0):
        """causal mask + workspace for head sizes the MFMA prefill kernel does not cover"""
        key = ("prefill", s, len_buf, pos0)
        if key not in self._bufs:
            c, dev = self.cfg, self.device
            mask = torch.tril(torch.ones(s, len_buf, dtype=torch.int8, device=dev), diagonal=pos0).contiguous()
            ws = ops.decode_attn_workspace(1, s, c.num_heads, c.dim_head, len_buf, dev)
            self._bufs[key] = (mask, ws)
        return self._bufs[key]

    def prefill(self, ctx: DynBatchContext, task: int, prompt: torch.Tensor, chunk: int = 0):
        """Prompt encode, optionally in chunks of `chunk` tokens (the reference's chunked prefill,
        src/generator/batch_generator.cpp:1048-1084: a long prompt enters the batch piece by piece, each piece
        attending to the KV of the pieces before it)."""
        s = int(prompt.numel())
        if chunk <= 0 or chunk >= s:
            return self._encode_prompt(ctx, task, prompt, 0)
        logits = None
        if ctx.kv_quant:
            # INT8 KV cache: the prompt keeps attending to its UNquantised K/V rows across the chunks, held in temporary
            # buffers for the duration of the prompt (dyn_batch->unquant_key_buf, attention.cpp:497-510) while the codes go
            # to the cache; the decode steps that follow read the cache
            c = self.cfg
            ctx.unquant_kv[task] = torch.zeros((c.num_layers, 2, (s + 63) // 64 * 64, c.num_kv_heads, c.dim_head),
                                               dtype=c.torch_dtype, device=self.device)
        try:
            for p0 in range(0, s, chunk):
                logits = self._encode_prompt(ctx, task, prompt[p0:p0 + chunk], p0)
        finally:
            ctx.unquant_kv.pop(task, None)
        return logits

    def _encode_prompt(self, ctx: DynBatchContext, task: int, prompt: torch.Tensor, pos0: int):
        """LLaMA::encode's switch (src/model/llama.cpp:102-110): with DUAL_STREAM=1, more than one TP rank and more
        than DUAL_STREAM_THRESHOLD (1024) tokens in the piece, the layers run as dual_stream_encode."""
        if (self.tp and int(os.environ.get("DUAL_STREAM", "0")) > 0 and not ctx.kv_quant
                and int(prompt.numel()) > int(os.environ.get("DUAL_STREAM_THRESHOLD", "1024"))):
            return self._prefill_dual_stream(ctx, task, prompt, pos0)
        return self._prefill_chunk(ctx, task, prompt, pos0)

    def _prefill_dual_stream(self, ctx: DynBatchContext, task: int, prompt: torch.Tensor, pos0: int):
        """EncoderLayer::impl::dual_stream_encode (src/nn/block/block.cpp:205-441) on two HIP streams: the piece is
        cut in DUAL_STREAM_NUM_SPLIT (2) parts of round_up(ceil(S / 2), 16) rows; every row-parallel partial output
        (attn_out, w_out) is all-reduced on a second, higher-priority stream while the main stream computes the other
        part, so the xGMI transfer of one half hides behind the GEMMs / attention of the other.  Part k of a layer
        starts with add_fuse_ln (layernorm.cu:227-302: c = T(hidden + reduced), norm of the fp32 sum) once its own
        reduce has landed; the later part attends to the earlier part's K/V through the cache, like a prefill chunk.
        Ordering is stream events only (main -> reduce before the collective, reduce -> main before the add); the
        partial stays referenced until the main stream has consumed it, so no allocator hand-over is needed."""
        c, dev = self.cfg, self.device
        s = int(prompt.numel())
        if s < 1 or pos0 + s + 1 > ctx.max_len_buf:
            raise ops.ZLError("prompt does not fit the task's KV buffer")
        num_split = max(1, int(os.environ.get("DUAL_STREAM_NUM_SPLIT", "2")))
        round_up = max(1, int(os.environ.get("DUAL_STREAM_SPLIT_ROUND_UP", "16")))
        part = -(-(-(-s // num_split)) // round_up) * round_up
        bounds = [(a, min(a + part, s)) for a in range(0, s, part)]
        tokens = prompt.to(device=dev, dtype=torch.int32).contiguous()
        pos = torch.arange(pos0, pos0 + s, dtype=torch.int32, device=dev)
        hidden_all = ops.embedding(tokens, self.token_embedding, c.scale_emb)
        cos, sin = self._rope_tables(pos)
        buf_lens = ctx.buf_lens[task:task + 1]
        scale = 1.0 / math.sqrt(c.dim_head)
        main = torch.cuda.current_stream(dev)
        if "reduce_stream" not in self._bufs:
            self._bufs["reduce_stream"] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("DUAL_STREAM_PRIORITY", "-1")))
        red = self._bufs["reduce_stream"]
        main_ev = [torch.cuda.Event() for _ in bounds]
        done_ev = [torch.cuda.Event() for _ in bounds]
        hidden = [hidden_all[a:b] for a, b in bounds]
        pending = [None] * len(bounds)

        def reduce_async(k, partial):
            main_ev[k].record(main)
            red.wait_event(main_ev[k])
            with torch.cuda.stream(red):
                self.tp.all_reduce_sum(partial)
                done_ev[k].record(red)
            pending[k] = partial

        def reduced(k):
            partial, pending[k] = pending[k], None
            main.wait_event(done_ev[k])
            return partial

        for li, layer in enumerate(self.layers):
            ka, va = ctx.k_addrs[li][task:task + 1], ctx.v_addrs[li][task:task + 1]
            for k, (a, b) in enumerate(bounds):
                n = b - a
                if li == 0:
                    xn = ops.rmsnorm(hidden[k], layer.ln_attn, c.eps)
                else:
                    xn, hidden[k] = ops.rmsnorm(hidden[k], layer.ln_attn, c.eps, x2=reduced(k))
                qkv = layer.project_qkv(None, c.eps, normed=xn)
                self.apply_qk_norm(layer, qkv)
                q, kr, v = ops.rope_qk_cache(cos[a:b], sin[a:b], qkv, c.num_heads, c.num_kv_heads, c.dim_head, True)
                ops.copy_to_rag_buffer2(pos[a:b].view(1, n), buf_lens, kr.view(1, n, c.num_kv_heads, c.dim_head),
                                        v.view(1, n, c.num_kv_heads, c.dim_head), ka, va)
                if c.dim_head == 128:
                    att = ops.prefill_attention(q.view(n, c.num_heads, c.dim_head), ctx.kv[task][li, 0], ctx.kv[task][li, 1],
                                                pos0 + a, c.num_kv_heads, scale)
                else:
                    mask, ws = self._prefill_mask(n, ctx.max_len_buf, pos0 + a)
                    att = ops.multi_query_attention_rag_buffer(q.view(1, n, c.num_heads, c.dim_head), buf_lens, ka, va, mask,
                                                               scale, ctx.max_len_buf, c.num_kv_heads, workspace=ws)
                reduce_async(k, layer.row_partial(layer.attn_out, att.view(n, -1)))
            for k in range(len(bounds)):
                xn, hidden[k] = ops.rmsnorm(hidden[k], layer.ln_ff, c.eps, x2=reduced(k))
                reduce_async(k, layer.row_partial(layer.w_out, layer.ff_in(None, c.eps, normed=xn)))
        for k in range(len(bounds)):
            hidden[k] = ops.element_add_scale(hidden[k], reduced(k), 1.0, True)
        self.dual_stream_runs = getattr(self, "dual_stream_runs", 0) + 1
        logits = self._prompt_logits_and_pick(ctx, task, hidden[-1][-1:])
        ctx.positions[task] = pos0 + s
        ctx.placement[task] = pos0 + s
        ctx.valid_lens[task] = pos0 + s + 1
        ctx.steps_left = min(ctx.steps_left, ctx.max_len_buf - (pos0 + s))
        return logits

    def _prefill_chunk(self, ctx: DynBatchContext, task: int, prompt: torch.Tensor, pos0: int):
        """The "encode part" of a task (LLaMA::encode with len_q = prompt length for one task:
        src/model/llama.cpp:75-165, Attention::impl::NormalImpl::dynamic_batch_forward encode branch,
        src/nn/attention/attention.cpp:846-964 / attn_encode_group :442-622): runs the whole prompt through
        the layers, fills the task's KV buffers at slots 0..S-1, leaves the task ready for decode steps
        (tokens <- greedy first token, positions = placement = S, valid_lens = S + 1) and returns the logits of
        the last prompt position (1, vocab).  Sequence: separate RMSNorm, W4A16 GEMM (M = S: the M-tiled MFMA
        kernel, the arithmetic of the reference's M > 40 dequant + GEMM branch), rope_qk_cache,
        copy_to_rag_buffer2, causal MFMA attention (prefill_attention; other head sizes: the mask form of
        multi_query_attention_rag_buffer)."""
        c, dev = self.cfg, self.device
        s = int(prompt.numel())
        if s < 1 or pos0 + s + 1 > ctx.max_len_buf:
            raise ops.ZLError("prompt does not fit the task's KV buffer")
        unq = ctx.unquant_kv.get(task) if ctx.kv_quant else None
        if ctx.kv_quant and pos0 != 0 and unq is None:
            # a later piece without the prompt's temporary unquantised buffers (prefill(chunk=...) keeps them): the reference
            # falls back to de-quantising the cache there ("WARNING: de-quantize prompt kv cache!", attention.cpp:511-516)
            raise ops.ZLError("chunked prefill into the INT8 KV cache goes through LLaMA.prefill(chunk=...)")
        tokens = prompt.to(device=dev, dtype=torch.int32).contiguous()
        pos = torch.arange(pos0, pos0 + s, dtype=torch.int32, device=dev)
        hidden = ops.embedding(tokens, self.token_embedding, c.scale_emb)
        cos, sin = self._rope_tables(pos)
        placement = pos.view(1, s)
        buf_lens = ctx.buf_lens[task:task + 1]
        scale = 1.0 / math.sqrt(c.dim_head)
        for li, layer in enumerate(self.layers):
            ka, va = ctx.k_addrs[li][task:task + 1], ctx.v_addrs[li][task:task + 1]
            qkv = layer.project_qkv(hidden, c.eps)
            self.apply_qk_norm(layer, qkv)
            q, k, v = ops.rope_qk_cache(cos, sin, qkv, c.num_heads, c.num_kv_heads, c.dim_head, True)
            if ctx.kv_quant:
                # attn_encode_group with a quantised buffer (attention.cpp:494-510): the prompt attends to its own
                # UNquantised K/V rows while their codes go to the cache
                k3, v3 = k.view(s, c.num_kv_heads, c.dim_head), v.view(s, c.num_kv_heads, c.dim_head)
                ops.quant_copy_to_rag_buffer(pos, buf_lens, k3, v3, ka, va, ctx.ks_addrs[li][task:task + 1],
                                             ctx.vs_addrs[li][task:task + 1], len_q=s)
                if unq is not None:
                    # chunked: this piece's rows join the prompt's unquantised buffers, the piece attends to all of them
                    tk, tv = unq[li, 0], unq[li, 1]
                    tl = torch.tensor([tk.shape[0]], dtype=torch.int32, device=dev)
                    ops.copy_to_rag_buffer2(placement, tl, k3.view(1, s, c.num_kv_heads, c.dim_head), v3.view(1, s, c.num_kv_heads, c.dim_head),
                                            ops.make_ptr_table([tk]), ops.make_ptr_table([tv]))
                    if c.dim_head == 128:
                        att = ops.prefill_attention(q.view(s, c.num_heads, c.dim_head), tk, tv, pos0, c.num_kv_heads, scale)
                    else:
                        mask, ws = self._prefill_mask(s, tk.shape[0], pos0)
                        att = ops.multi_query_attention_rag_buffer(q.view(1, s, c.num_heads, c.dim_head), tl, ops.make_ptr_table([tk]),
                                                                   ops.make_ptr_table([tv]), mask, scale, tk.shape[0], c.num_kv_heads,
                                                                   workspace=ws)
                elif c.dim_head == 128:
                    att = ops.prefill_attention(q.view(s, c.num_heads, c.dim_head), k3, v3, 0, c.num_kv_heads, scale)
                else:
                    mask, ws = self._prefill_mask(s, s, 0)
                    att = ops.multi_query_attention_rag_buffer(
                        q.view(1, s, c.num_heads, c.dim_head), torch.tensor([s], dtype=torch.int32, device=dev),
                        ops.make_ptr_table([k3]), ops.make_ptr_table([v3]), mask, scale, s, c.num_kv_heads, workspace=ws)
                layer.attn_out_add(att.view(s, -1), hidden)
                layer.ff_add(hidden, c.eps)
                continue
            ops.copy_to_rag_buffer2(placement, buf_lens, k.view(1, s, c.num_kv_heads, c.dim_head),
                                    v.view(1, s, c.num_kv_heads, c.dim_head), ka, va)
            if c.dim_head == 128:
                att = ops.prefill_attention(q.view(s, c.num_heads, c.dim_head), ctx.kv[task][li, 0], ctx.kv[task][li, 1], pos0,
                                            c.num_kv_heads, scale)
            else:
                mask, ws = self._prefill_mask(s, ctx.max_len_buf, pos0)
                att = ops.multi_query_attention_rag_buffer(q.view(1, s, c.num_heads, c.dim_head), buf_lens, ka, va, mask,
                                                           scale, ctx.max_len_buf, c.num_kv_heads, workspace=ws)
            layer.attn_out_add(att.view(s, -1), hidden)
            layer.ff_add(hidden, c.eps)
        logits = self._prompt_logits_and_pick(ctx, task, hidden[s - 1:s])
        ctx.positions[task] = pos0 + s
        ctx.placement[task] = pos0 + s
        ctx.valid_lens[task] = pos0 + s + 1
        ctx.steps_left = min(ctx.steps_left, ctx.max_len_buf - (pos0 + s))
        return logits

    def _prompt_logits_and_pick(self, ctx, task, last_hidden):
        """logits of a prompt's last row and the first generated token into ctx.tokens[task]: the pick rides the lm_head launch
        (zl_gemm_nt_small_m_argmax leaves one candidate per wavefront, zl_greedy_advance reduces them -- first index on ties,
        as torch.argmax) instead of three torch launches over the 128 k logits; TP picks on the gathered row (zl_argmax_advance)"""
        if self.tp:
            logits = self._logits(last_hidden)
            ops.argmax_advance(logits[:1], tokens=ctx.tokens[task:task + 1])
            return logits
        key = ("argmax", 1)
        if key not in self._bufs:
            self._bufs[key] = (ops.argmax_workspace(1, self.cfg.vocab_size, self.device),
                               torch.empty(1, dtype=torch.int64, device=self.device))
        ws = self._bufs[key][0]
        logits = self._logits(last_hidden, argmax_ws=ws)
        ops.greedy_advance(ws, 1, self.cfg.vocab_size, tokens=ctx.tokens[task:task + 1])
        return logits

    def step_greedy(self, ctx: DynBatchContext, skip_gemv=False):
        """One greedy decode step entirely on the device (graph-capturable): encode, pick the arg-max token
        inside the lm_head launch + one small reduction, and advance the batch state.  Returns
        (logits, next_tokens int64)."""
        b = ctx.tokens.numel()
        if b > 4 or self.tp:   # the in-launch pick rides the small-M GEMV; bigger batches / TP: one pick + advance launch over the logits
            logits = self.encode(ctx)
            key = ("next", b)
            if key not in self._bufs:
                self._bufs[key] = torch.empty(b, dtype=torch.int64, device=self.device)
            nxt = self._bufs[key]
            ops.argmax_advance(logits, tokens=ctx.tokens, positions=ctx.positions, placement=ctx.placement, valid_lens=ctx.valid_lens,
                               next_tokens=nxt)
            ctx.steps_left -= 1
            return logits, nxt
        key = ("argmax", b)
        if key not in self._bufs:
            self._bufs[key] = (ops.argmax_workspace(b, self.cfg.vocab_size, self.device),
                               torch.empty(b, dtype=torch.int64, device=self.device))
        ws, nxt = self._bufs[key]
        logits = self.encode(ctx, argmax_ws=ws, skip_gemv=skip_gemv)
        ops.greedy_advance(ws, b, self.cfg.vocab_size, ctx.tokens, ctx.positions, ctx.placement, ctx.valid_lens, nxt)
        ctx.steps_left -= 1
        return logits, nxt

    def advance(self, ctx: DynBatchContext, next_tokens: torch.Tensor):
        """Device-side bookkeeping between steps (what fill_search_tokens does on the host in the
        reference, src/generator/batch_generator.cpp:1226-1335): no host sync, graph-capturable."""
        ctx.tokens.copy_(next_tokens.to(torch.int32))
        ctx.positions.add_(1)
        ctx.placement.add_(1)
        ctx.valid_lens.add_(1)
        ctx.steps_left -= 1

    def weight_bytes(self):
        return sum(l.weight_bytes() for l in self.layers)

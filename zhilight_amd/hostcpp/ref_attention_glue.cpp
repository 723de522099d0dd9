// ref_attention_glue.cpp -- what is linked next to the REFERENCE's own src/nn/attention/attention.cpp (compiled unmodified from
// /root/reference, zhilight_amd/build.py: build_refcompile) so that its decode path -- Attention::dyn_rag_forward ->
// NormalImpl::dynamic_batch_forward (attention.cpp:846-964) -> attn_search_rag (:636-741) -- EXECUTES on the MI355X boundary
// (VERDICT r03 item 7).  The reference unit calls the boundary's operators by name (nn::Linear from the reference's linear.cpp,
// nn::rope_qk_cache / rotary_embedding_qk, copy_to_rag_buffer2, get_mqa_workspace, multi_query_attention_rag_buffer: nn_amd.cpp);
// what it needs besides them are the classes AROUND the operators, which live in the reference's .cu / scheduler files:
//   1. kvcache::KVCache / TransformerBuffer (src/kvcache/transformer_buffer.h:11-63; implementation transformer_buffer.cu): the
//      per-task, per-layer K / V buffers model::RagBufferContext (header-only, rag_buffer_context.h) hands out addresses of.
//      Implemented here on core::Context::tensor: resize keeps the rows written so far;
//   2. nn::RotaryEmbedding (src/nn/position/rotary_embedding.h:8-37): constructor, is_normal / is_neox_style, and forward / rotate
//      on top of zl_rope_cos_sin* + zl_rope_qk_cache;
//   3. model::ModelContext's constructor (src/model/model_context.h:121-126): the context the layer dynamic_casts to, with the
//      dyn_batch / rag_buffer slots; no buffers, reducers or engine behind it here;
//   4. nn::FlashDecoding::mha_fwd (the prompt attention of attn_encode_group, :553-562) over zl_prefill_attn and
//      TransformerBuffer::copy over zl_copy_to_rag_buffer2, so that the encode part runs as well; the names that stay off the
//      path (FlashDecoding's varlen / compact entry points, the unfused softmax route, the static-batch copy_to_buffer, the MLA
//      implementation): definitions that throw and say so;
//   5. the pybind11 class RefAttention driving all of it from numpy: load a layer's weights under the reference's parameter
//      names, fill per-task KV histories, run decode steps.  tests/test_gpu_refcompile.py compares with the oracle.
// Test infrastructure: nothing in the product links this file.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime.h>

#include "kvcache/transformer_buffer.h"
#include "model/dyn_batch_context.h"
#include "model/model.h"
#include "model/model_context.h"
#include "model/rag_buffer_context.h"
#include "nn/attention/attention.h"
#include "nn/attention/attention_base.hpp"
#include "nn/attention/attention_kernel.h"
#include "nn/attention/flash_decoding.h"
#include "nn/position/rotary_embedding.h"
#include "zhilight_amd.h"

namespace py = pybind11;
using bmengine::core::Context;
using bmengine::core::DataType;
using bmengine::core::Tensor;

#define ZL_OFF_PATH(what) \
    throw BMEngineException(std::string(what) + " is not on the decode path this module runs (SURVEY.md section 8)", __FILE__, __LINE__, __func__)
#define ZL_CK(call, what)                                                                                        \
    do {                                                                                                         \
        const int st_ = (call);                                                                                  \
        if (st_ != 0) throw BMEngineException(std::string(what) + ": " + zl_status_string(st_), __FILE__, __LINE__, __func__); \
    } while (0)

// ---- 1. the KV buffers ------------------------------------------------------------------------------------------------------
namespace kvcache {

KVCache::KVCache(int batch_size, int num_layers, int num_heads, int dim_head, core::DataType dtype, bool parallel, bool BSHD)
    : batch_size(batch_size), num_layers(num_layers), num_heads(num_heads), dim_head(dim_head), dtype(dtype), parallel(parallel), BSHD(BSHD) {}

TransformerBuffer::TransformerBuffer(int batch_size, int num_layers, int num_heads, int dim_head, core::DataType dtype, bool parallel, bool BSHD)
    : KVCache(batch_size, num_layers, num_heads, dim_head, dtype, parallel, BSHD) {
    buffer.resize(num_layers);
    scales_.resize(num_layers);
}
TransformerBuffer::TransformerBuffer(const KVCacheConfig& c)
    : TransformerBuffer(-1, c.num_layers, c.num_heads, c.dim_head, c.dtype, false, c.BSHD) {
    scale_dtype_ = c.scale_dtype;
    layer_devices = c.layer_devices;
}
TransformerBuffer::~TransformerBuffer() = default;

void TransformerBuffer::check_layer(int i) const {
    BM_ASSERT(i >= 0 && (size_t)i < num_layers, "TransformerBuffer: layer out of range");
}
const core::Tensor& TransformerBuffer::operator[](int i) const { check_layer(i); return buffer[i]; }
core::Tensor& TransformerBuffer::operator[](int i) { check_layer(i); return buffer[i]; }
const core::Tensor& TransformerBuffer::get_scale(int i) const { check_layer(i); return scales_[i]; }

// grow every layer's buffer to new_length rows, keeping what has been written (per-task buffers: batch_size == -1;
// (len, heads, dim) under BSHD, (heads, len, dim) otherwise)
void TransformerBuffer::resize(const core::Context& ctx, size_t new_length) {
    BM_ASSERT(is_dyn_batch(), "TransformerBuffer: only the per-task (ragged) form is provided here");
    hipStream_t st = ctx.current_cuda_stream();
    auto grow = [&](core::Tensor& old, size_t row_elems, core::DataType dt) {
        const size_t esz = core::get_elem_size(dt);
        const size_t old_len = old.numel() ? old.size(BSHD ? 0 : 1) : 0;
        if (old_len >= new_length) return;
        core::Tensor nw = BSHD ? ctx.tensor({new_length, num_heads, row_elems}, dt) : ctx.tensor({num_heads, new_length, row_elems}, dt);
        BM_CUDART_ASSERT(hipMemsetAsync(nw.data(), 0, nw.nbytes(), st));
        if (old_len) {
            if (BSHD) {
                BM_CUDART_ASSERT(hipMemcpyAsync(nw.data(), old.data(), old.nbytes(), hipMemcpyDeviceToDevice, st));
            } else {
                BM_CUDART_ASSERT(hipMemcpy2DAsync(nw.data(), new_length * row_elems * esz, old.data(), old_len * row_elems * esz,
                                                  old_len * row_elems * esz, num_heads, hipMemcpyDeviceToDevice, st));
            }
        }
        BM_CUDART_ASSERT(hipStreamSynchronize(st));      // the old block goes back to the pool below
        old = nw;
    };
    for (size_t i = 0; i < num_layers; ++i) {
        grow(buffer[i], dim_head, dtype);
        if (scale_dtype_) {
            // one scale per (row, head): (len, heads) / (heads, len) -- as a last dimension of 1
            core::Tensor& sc = scales_[i];
            const size_t old_len = sc.numel() ? sc.size(BSHD ? 0 : 1) : 0;
            if (old_len < new_length) {
                core::Tensor nw = BSHD ? ctx.tensor({new_length, num_heads}, *scale_dtype_) : ctx.tensor({num_heads, new_length}, *scale_dtype_);
                BM_CUDART_ASSERT(hipMemsetAsync(nw.data(), 0, nw.nbytes(), st));
                const size_t esz = core::get_elem_size(*scale_dtype_);
                if (old_len) {
                    if (BSHD) BM_CUDART_ASSERT(hipMemcpyAsync(nw.data(), sc.data(), sc.nbytes(), hipMemcpyDeviceToDevice, st));
                    else BM_CUDART_ASSERT(hipMemcpy2DAsync(nw.data(), new_length * esz, sc.data(), old_len * esz, old_len * esz, num_heads, hipMemcpyDeviceToDevice, st));
                }
                BM_CUDART_ASSERT(hipStreamSynchronize(st));
                sc = nw;
            }
        }
    }
}
// scatter the rows of src (n, heads, dim) into layer `layer` at the buffer rows `placement` names and hand the layer's buffer back
// (attn_encode_group, attention.cpp:513-514: the prompt's keys / values enter the task's buffer here): zl_copy_to_rag_buffer2 with
// one task whose "value" operand is the same tensor
core::Tensor TransformerBuffer::copy(const core::Context& ctx, int layer, const core::Tensor& src, const core::Tensor& placement, int start,
                                     bool need_dequant) {
    check_layer(layer);
    core::Tensor& buf = buffer[layer];
    if (scale_dtype_) {
        // the INT8 cache (transformer_buffer.cu:128-152): the chunk's rows become u8 codes + one fp32 scale per (row, head) at rows
        // start .. start + n - 1 (the reference ignores `placement` here as well); the caller attends over `src` itself, or -- a later
        // chunk, need_dequant -- over the already cached rows brought back to T in front of it
        BM_ASSERT(BSHD && src.ndim() == 3 && *scale_dtype_ == core::DataType::kFloat, "quantised buffers: (len, heads, dim) u8 codes with fp32 scales");
        const int64_t n = (int64_t)src.size(0), len_buf = (int64_t)buf.size(0), row = (int64_t)num_heads * dim_head;
        BM_ASSERT(start >= 0 && start + n <= len_buf, "TransformerBuffer::copy: rows past the buffer");
        const int dt = src.dtype() == core::DataType::kHalf ? ZL_F16 : ZL_BF16;
        zl_stream_t st = (zl_stream_t)ctx.current_cuda_stream();
        core::Tensor& sc = scales_[layer];
        ZL_CK(zl_quant_calc_scale_zp(src.data<uint16_t>(), buf.data<uint8_t>() + (size_t)start * row, sc.data<float>() + (size_t)start * num_heads,
                                     n * (int64_t)num_heads, (int64_t)dim_head, 128, dt, st), "quant_calc_scale");
        if (!(need_dequant && start > 0)) return src;
        core::Tensor out = ctx.tensor({(size_t)len_buf, num_heads, dim_head}, src.dtype());
        BM_CUDART_ASSERT(hipMemsetAsync(out.data(), 0, out.nbytes(), ctx.current_cuda_stream()));
        ZL_CK(zl_dequant_group(buf.data(), sc.data<float>(), out.data<uint16_t>(), (int64_t)start * num_heads, (int64_t)dim_head, 128, dt, st), "dequant_group");
        BM_CUDART_ASSERT(hipMemcpyAsync(out.data<char>() + (size_t)start * row * 2, src.data(), src.nbytes(), hipMemcpyDeviceToDevice, ctx.current_cuda_stream()));
        return out;
    }
    const int64_t n = (int64_t)placement.numel();
    BM_ASSERT(src.numel() == (size_t)n * num_heads * dim_head && placement.dtype() == core::DataType::kInt32, "TransformerBuffer::copy: shape mismatch");
    const int len_buf = (int)buf.size(BSHD ? 0 : 1);
    core::Tensor lens = ctx.tensor_of(std::vector<int>{len_buf});
    core::Tensor table = ctx.tensor_of(std::vector<void*>{buf.data()});
    ZL_CK(zl_copy_to_rag_buffer2(placement.data<int32_t>(), lens.data<int32_t>(), src.data<uint16_t>(), src.data<uint16_t>(),
                                 reinterpret_cast<uint16_t* const*>(table.data()), reinterpret_cast<uint16_t* const*>(table.data()), 1, n, (int64_t)num_heads,
                                 (int64_t)dim_head, BSHD ? 1 : 0, (zl_stream_t)ctx.current_cuda_stream()),
          "copy_to_rag_buffer2");
    BM_CUDART_ASSERT(hipStreamSynchronize(ctx.current_cuda_stream()));      // (lens / table go back to the pool)
    return buf;
}
void copy_to_buffer(int, int, int, int, const core::Tensor*, const core::Tensor&, const core::Tensor&, cudaStream_t, bool) {
    ZL_OFF_PATH("kvcache::copy_to_buffer (the static-batch forward; ragged buffers take copy_to_rag_buffer2)");
}

}  // namespace kvcache

// ---- 2. nn::RotaryEmbedding ---------------------------------------------------------------------------------------------------
namespace nn {

class RotaryEmbedding::impl {
public:
    model::ModelConfig cfg;
    explicit impl(const model::ModelConfig& c) : cfg(c) {}
    bool llama3() const { return cfg.rope_cfg.type == "llama3"; }
    bool plain() const { return cfg.rope_cfg.type.empty() || cfg.rope_cfg.type == "default" || cfg.rope_cfg.type == "rope"; }
    // cos / sin (n, dim_head) fp32 of the rows' positions
    void tables(const core::Context& ctx, const core::Tensor& pos, size_t d, core::Tensor* cs, core::Tensor* sn) const {
        const size_t n = pos.numel();
        BM_ASSERT(pos.dtype() == DataType::kInt32, "positions are int32");
        *cs = ctx.tensor({n, d}, DataType::kFloat);
        *sn = ctx.tensor({n, d}, DataType::kFloat);
        zl_stream_t st = (zl_stream_t)ctx.current_cuda_stream();
        const int neox = cfg.rope_cfg.neox_style ? 1 : 0;
        if (llama3())
            ZL_CK(zl_rope_cos_sin_llama3(pos.data<int32_t>(), cs->data<float>(), sn->data<float>(), n, d, cfg.rope_theta, cfg.rope_cfg.factor,
                                         cfg.rope_cfg.low_freq_factor, cfg.rope_cfg.high_freq_factor, (float)cfg.rope_cfg.original_max_position, neox, st),
                  "rope_cos_sin_llama3");
        else if (plain())
            ZL_CK(zl_rope_cos_sin(pos.data<int32_t>(), cs->data<float>(), sn->data<float>(), n, d, cfg.rope_theta, neox, st), "rope_cos_sin");
        else
            ZL_OFF_PATH("RotaryEmbedding with rope type '" + cfg.rope_cfg.type + "'");
    }
    // rotate the heads of x at the rows' positions.  x: (n, heads * d) or (n, heads, d), possibly a last-dimension SLICE of a wider
    // tensor (MLAImpl rotates the 64 rope dimensions inside 192-wide heads, and a 64-wide slice of the fused qkv_a output): the
    // rotation width d is the operand's own head width there (qk_rope_head_dim), strides come from the tensor
    core::Tensor rotate(const core::Context& ctx, const core::Tensor& pos, const core::Tensor& x, core::Tensor* output) const {
        const size_t n = pos.numel();
        BM_ASSERT(x.ndim() == 2 || x.ndim() == 3, "RotaryEmbedding: (n, heads * d) or (n, heads, d)");
        BM_ASSERT_EQ(x.size(0), n, "RotaryEmbedding: rows != positions");
        const size_t d = cfg.qk_rope_head_dim > 0 ? (size_t)cfg.qk_rope_head_dim : (size_t)cfg.dim_head;
        size_t heads, x_sh;
        if (x.ndim() == 3) {
            BM_ASSERT(x.size(2) == d && x.stride(2) == 1, "RotaryEmbedding: head width");
            heads = x.size(1);
            x_sh = x.stride(1);
        } else {
            BM_ASSERT(x.size(1) % d == 0 && x.stride(1) == 1, "RotaryEmbedding: row width");
            heads = x.size(1) / d;
            x_sh = d;
        }
        core::Tensor cs, sn;
        tables(ctx, pos, d, &cs, &sn);
        core::Tensor out = output ? *output : ctx.tensor(x.shape(), x.dtype());
        BM_ASSERT(out.numel() == x.numel() && out.stride(-1) == 1, "RotaryEmbedding: output shape");
        const size_t o_sh = out.ndim() == 3 ? out.stride(1) : d;
        ZL_CK(zl_rope_rotate(cs.data<float>(), sn.data<float>(), x.data<uint16_t>(), out.data<uint16_t>(), n, heads, d, x.stride(0), x_sh, out.stride(0), o_sh,
                             cfg.rope_cfg.neox_style ? 1 : 0, x.dtype() == DataType::kHalf ? ZL_F16 : ZL_BF16, (zl_stream_t)ctx.current_cuda_stream()),
              "rope_rotate");
        return out;
    }
};

RotaryEmbedding::RotaryEmbedding(const core::Context&, model::ModelConfig cfg) : pimpl(new impl(cfg)) {}
RotaryEmbedding::~RotaryEmbedding() = default;
bool RotaryEmbedding::is_normal() const { return pimpl->plain(); }
bool RotaryEmbedding::is_neox_style() const { return pimpl->cfg.rope_cfg.neox_style; }
std::tuple<core::Tensor, core::Tensor> RotaryEmbedding::forward(const core::Context& ctx, const core::Tensor& pos, const core::Tensor& q,
                                                                const core::Tensor& k) {
    return std::make_tuple(pimpl->rotate(ctx, pos, q, nullptr), pimpl->rotate(ctx, pos, k, nullptr));
}
core::Tensor RotaryEmbedding::rotate(const core::Context& ctx, const core::Tensor& pos, const core::Tensor& q, core::Tensor* output) {
    return pimpl->rotate(ctx, pos, q, output);
}
void RotaryEmbedding::rotate_inplace(const core::Context& ctx, const core::Tensor& pos, core::Tensor& q) { pimpl->rotate(ctx, pos, q, &q); }

// ---- 4. off-path names ------------------------------------------------------------------------------------------------------
FlashDecoding::FlashDecoding(const Context&) {}
FlashDecoding::~FlashDecoding() = default;
core::Tensor FlashDecoding::forward(const Context&, Tensor&, const Tensor&, const Tensor&, Tensor*, const Tensor*, const Tensor*, int, int, bool, bool, int, int,
                                    float) {
    ZL_OFF_PATH("nn::FlashDecoding::forward (USE_FA_DECODING; the ragged decode takes multi_query_attention_rag_buffer)");
}
core::Tensor FlashDecoding::compact_kv_fwd(const Context&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                                           const Tensor*, Tensor, float) {
    ZL_OFF_PATH("nn::FlashDecoding::compact_kv_fwd (dynamic batching off)");
}
// the prompt ("encode") attention of one task: q (1, n, H, D) against the first len_kv rows of its buffers (1, len_kv, Hkv, D), which
// already hold the chunk's own rows; causal with the chunk at the END of the keys (flash-attn's bottom-right alignment) =
// zl_prefill_attn with pos0 = len_kv - n (attention.cpp:553-562 is the only call site)
core::Tensor FlashDecoding::mha_fwd(const Context& ctx, Tensor& q, const Tensor& k, const Tensor& v, Tensor* out_, Tensor* alibi_slopes, const float p_dropout,
                                    const float softmax_scale, bool is_causal, int window_size_left, int window_size_right, const float softcap,
                                    const bool return_softmax) {
    BM_ASSERT(q.ndim() == 4 && k.ndim() == 4 && q.size(0) == 1 && k.size(0) == 1, "mha_fwd: (1, len, heads, dim) operands");
    if (!is_causal || alibi_slopes || p_dropout != 0.f || window_size_left >= 0 || window_size_right >= 0 || softcap != 0.f || return_softmax)
        ZL_OFF_PATH("nn::FlashDecoding::mha_fwd with anything but plain causal attention");
    const int64_t n = q.size(1), h = q.size(2), d = q.size(3), len_kv = k.size(1), hkv = k.size(2);
    Tensor out = out_ ? *out_ : ctx.tensor(q.shape(), q.dtype());
    ZL_CK(zl_prefill_attn(q.data<uint16_t>(), k.data<uint16_t>(), v.data<uint16_t>(), out.data<uint16_t>(), n, len_kv - n, h, hkv, d,
                          softmax_scale > 0.f ? softmax_scale : 1.0f / sqrtf((float)d), len_kv, 1, q.dtype() == DataType::kHalf ? ZL_F16 : ZL_BF16,
                          (zl_stream_t)ctx.current_cuda_stream()),
          "prefill_attn");
    return out;
}
void attn_softmax(const core::Context&, float, const core::Tensor&, const core::Tensor&, const core::Tensor&) {
    ZL_OFF_PATH("nn::attn_softmax (the unfused gemm + softmax + gemm route)");
}
void multi_query_self_attention(const core::Context&, const core::Tensor&, const core::Tensor&, const core::Tensor&, const core::Tensor&, float, core::Tensor&, int) {
    ZL_OFF_PATH("nn::multi_query_self_attention (prompt encode without flash attention)");
}

}  // namespace nn

// ---- 3. model::ModelContext ---------------------------------------------------------------------------------------------------
namespace model {
ModelContext::ModelContext(bmengine::core::Context&& ctx, const ModelBase& md, int /*batch_size*/, bool parallel, bool BSHD)
    : bmengine::core::Context(std::move(ctx)), cfg(md.cfg), model_(md), parallel_(parallel) {
    layer_devices.assign(md.num_layers, active_device());
    set_BSHD(BSHD);
    latent_cache_ = cfg.kv_lora_rank > 0 && std::getenv("LATENT_CACHE") && std::atoi(std::getenv("LATENT_CACHE")) == 1;   // (model_context.cpp: the same switch)
}
}  // namespace model

// ---- 5. the test class ------------------------------------------------------------------------------------------------------
namespace {

DataType np_dtype(const py::array& a) {
    const char k = a.dtype().kind();
    const size_t sz = (size_t)a.dtype().itemsize();
    if (k == 'f' && sz == 2) return DataType::kHalf;
    if (k == 'f' && sz == 4) return DataType::kFloat;
    if (k == 'i' && sz == 4) return DataType::kInt32;
    if ((k == 'i' || k == 'u') && sz == 1) return DataType::kInt8;
    if ((k == 'i' || k == 'u') && sz == 2) return DataType::kInt16;
    throw std::runtime_error("unsupported numpy dtype");
}
Tensor host_tensor(const py::array& a, const std::string& name) {
    if (!(a.flags() & py::array::c_style)) throw std::runtime_error(name + ": C-contiguous array expected");
    std::vector<size_t> shape(a.shape(), a.shape() + a.ndim());
    Tensor t = Tensor::from_external(shape, np_dtype(a), const_cast<void*>(a.data()), (size_t)a.nbytes(), -1, false);
    t.set_name(name);
    return t;
}
Tensor to_device(const Context& ctx, const py::array& a, const std::string& name) {
    Tensor h = host_tensor(a, name);
    Tensor d = ctx.tensor(h.shape(), h.dtype());
    d.from_buffer(h.data(), false, ctx.current_cuda_stream());
    return d;
}
py::array to_numpy(const Context& ctx, const Tensor& t) {
    std::vector<py::ssize_t> shape(t.shape().begin(), t.shape().end());
    py::dtype dt = t.dtype() == DataType::kHalf ? py::dtype("float16") : t.dtype() == DataType::kFloat ? py::dtype("float32")
                 : t.dtype() == DataType::kInt32 ? py::dtype("int32") : py::dtype("int8");
    py::array out(dt, shape);
    t.to_buffer(out.mutable_data(), ctx.current_cuda_stream());
    return out;
}

class DummyModel : public model::ModelBase {
public:
    explicit DummyModel(const model::ModelConfig& c) : model::ModelBase(c) {}
    const char* layer_type() const override { return "DummyModel"; }
};

// One reference nn::Attention layer inside a reference ModelContext with reference DynBatchContext / RagBufferContext objects
class RefAttention {
public:
    // mla = (q_lora_rank, kv_lora_rank, qk_nope_head_dim, qk_rope_head_dim, v_head_dim), all zero for ordinary attention.  With
    // kv_lora_rank > 0 the reference builds its MLAImpl (multi_head_latent_attention.cpp) and -- under LATENT_CACHE=1 -- the task
    // buffers hold ONE 576-wide latent row per key and no value buffer (RagBufferContext::has_v, rag_buffer_context.h:88-90)
    RefAttention(int dim_model, int num_heads, int num_kv_heads, int dim_head, float rope_theta, const std::string& model_type, int quant_type,
                 int group_size, int num_layers, bool bshd, int device, const std::vector<int>& mla)
        : cfg_(with_mla(model::ModelConfig(model_type, num_layers, dim_model, num_heads, dim_head, 4 * dim_model, 1024, 1e-5f, num_kv_heads, DataType::kHalf),
                        rope_theta, mla)),
          md_(cfg_),
          ctx_(Context(device), md_, 1, false, bshd),
          num_layers_(num_layers) {
        model::QuantConfig qc(quant_type);
        qc.group_size = group_size;
        attn_.reset(new nn::Attention(ctx_, cfg_, qc, false));
        const bool latent = cfg_.kv_lora_rank > 0 && ctx_.latent_cache();
        // KV_CACHE_DTYPE=int8: u8 codes + fp32 scales, the switch ModelContext::get_kv_cache_config reads (model_context.cpp:61-80)
        const char* kvd = std::getenv("KV_CACHE_DTYPE");
        const bool kv_int8 = kvd && std::string(kvd) == "int8";
        BM_ASSERT(!(kv_int8 && latent), "the latent cache is not quantised");
        kvcache::KVCacheConfig kc{num_layers, latent ? 1 : num_kv_heads, latent ? cfg_.kv_lora_rank + cfg_.qk_rope_head_dim : dim_head,
                                  kv_int8 ? DataType::kInt8 : DataType::kHalf, bshd,
                                  kv_int8 ? std::make_shared<DataType>(DataType::kFloat) : nullptr, std::vector<int>(num_layers, device)};
        kvcache::KVCacheConfig vc = kc;
        if (latent) vc.dim_head = 0;
        rag_ = std::make_shared<model::RagBufferContext>(kc, vc);
        ctx_.set_rag_buffer(rag_);
    }
    static model::ModelConfig with_mla(model::ModelConfig c, float rope_theta, const std::vector<int>& mla) {
        c.rope_theta = rope_theta;
        if (mla.size() == 5 && mla[1] > 0) {
            c.q_lora_rank = mla[0]; c.kv_lora_rank = mla[1]; c.qk_nope_head_dim = mla[2]; c.qk_rope_head_dim = mla[3]; c.v_head_dim = mla[4];
        }
        return c;
    }
    void load(const std::map<std::string, py::array>& arrays, const std::string& prefix) {
        std::map<std::string, const Tensor> sd;
        for (auto& kv : arrays) sd.emplace(kv.first, host_tensor(kv.second, kv.first));
        attn_->load_state_dict(ctx_, sd, prefix, false);
    }
    // task b's buffers hold len_buf rows; k / v (n, hkv, d) fill rows 0 .. n - 1 of `layer`
    void set_history(int b, int layer, int len_buf, const py::array& k, const py::array& v) {
        rag_->resize_task_buf(ctx_, b, (size_t)len_buf);
        fill(rag_->buf_k(b)[layer], k);
        if (rag_->config_v_.dim_head > 0) fill(rag_->buf_v(b)[layer], v);
    }
    py::array get_k(int b, int layer) { return to_numpy(ctx_, rag_->buf_k(b, layer)); }
    py::array get_v(int b, int layer) { return to_numpy(ctx_, rag_->buf_v(b, layer)); }
    // a quantised cache: the fp32 scales next to the codes, and history given as codes + scales
    py::array get_k_scale(int b, int layer) { return to_numpy(ctx_, rag_->buf_k(b).get_scale(layer)); }
    py::array get_v_scale(int b, int layer) { return to_numpy(ctx_, rag_->buf_v(b).get_scale(layer)); }
    void set_history_quant(int b, int layer, int len_buf, const py::array& k, const py::array& v, const py::array& ks, const py::array& vs) {
        BM_ASSERT(rag_->is_cache_quant(), "set_history_quant: KV_CACHE_DTYPE=int8 only");
        rag_->resize_task_buf(ctx_, b, (size_t)len_buf);
        fill(rag_->buf_k(b)[layer], k);
        fill(rag_->buf_v(b)[layer], v);
        fill(const_cast<Tensor&>(rag_->buf_k(b).get_scale(layer)), ks);
        fill(const_cast<Tensor&>(rag_->buf_v(b).get_scale(layer)), vs);
    }
    bool cache_quant() { return rag_->is_cache_quant(); }
    bool latent_cache() { return ctx_.latent_cache(); }
    // one decode step of `layer` for the tasks 0 .. B - 1: hidden (B, dim_model) fp16, positions (B) int32, placement (B) int32 = the
    // buffer row the new key goes to, mask (sum over tasks of len_buf) int8.  with_rope_cache: DynBatchContext::rope_cache filled
    // (RopePreparer's tables), the reference then takes rope_qk_cache instead of rotary_embedding_qk.
    py::array decode_step(int layer, const py::array& hidden, const py::array& positions, const py::array& placement, const py::array& mask,
                          bool with_rope_cache) {
        const size_t B = (size_t)hidden.shape(0);
        auto dyn = std::make_shared<model::DynBatchContext>();
        dyn->s_placement = to_device(ctx_, placement, "s_placement").view({B, 1});
        dyn->s_position = to_device(ctx_, positions, "s_position");
        dyn->s_mask = to_device(ctx_, mask, "s_mask");
        dyn->sv_len_buf.clear();
        for (size_t b = 0; b < B; ++b) dyn->sv_len_buf.push_back((int)rag_->get_buf_len(b));
        dyn->s_len_buf = ctx_.tensor_of(dyn->sv_len_buf);
        if (with_rope_cache) {
            // (the tables RopePreparer leaves in the context: cos / sin of every row's position)
            Tensor cs = ctx_.tensor({B, (size_t)cfg_.dim_head}, DataType::kFloat), sn = ctx_.tensor({B, (size_t)cfg_.dim_head}, DataType::kFloat);
            ZL_CK(zl_rope_cos_sin(dyn->s_position.data<int32_t>(), cs.data<float>(), sn.data<float>(), B, cfg_.dim_head, cfg_.rope_theta, 1,
                                  (zl_stream_t)ctx_.current_cuda_stream()), "rope_cos_sin");
            dyn->rope_cache.cos = cs;
            dyn->rope_cache.sin = sn;
        }
        ctx_.set_dyn_batch(dyn);
        ctx_.set_current_layer(layer);
        rag_->set_buffer_addr(ctx_);
        Tensor x = to_device(ctx_, hidden, "hidden");
        Tensor y = attn_->dyn_rag_forward(ctx_, x, dyn->s_position, nullptr);
        py::array out = to_numpy(ctx_, y);
        ctx_.set_dyn_batch(nullptr);
        return out;
    }

    // the encode part of task b: a prompt chunk hidden (n, dim_model) at positions pos0 .. pos0 + n - 1 of a buffer of len_buf rows
    // (DynBatchContext's e_* fields as the batch generator fills them for one task; no search part)
    py::array encode(int layer, int b, int len_buf, const py::array& hidden, int pos0) {
        const size_t n = (size_t)hidden.shape(0);
        rag_->resize_task_buf(ctx_, b, (size_t)len_buf);
        auto dyn = std::make_shared<model::DynBatchContext>();
        std::vector<int> pos(n);
        for (size_t i = 0; i < n; ++i) pos[i] = pos0 + (int)i;
        dyn->e_placement = ctx_.tensor_of(pos);
        dyn->e_position = ctx_.tensor_of(pos);
        std::vector<int8_t> mask(n * (size_t)len_buf);
        for (size_t i = 0; i < n; ++i)
            for (int j = 0; j < len_buf; ++j) mask[i * len_buf + j] = j <= pos0 + (int)i;
        dyn->e_mask = ctx_.tensor_of(mask);
        dyn->ev_batch = {b};
        dyn->ev_input_len = {(int)n};
        dyn->full_input_len = {pos0 + (int)n};
        dyn->ev_len_buf = {len_buf};
        dyn->s_placement = Tensor();
        ctx_.set_dyn_batch(dyn);
        ctx_.set_current_layer(layer);
        Tensor x = to_device(ctx_, hidden, "hidden");
        Tensor y = attn_->dyn_rag_forward(ctx_, x, dyn->e_position, nullptr);
        py::array out = to_numpy(ctx_, y);
        ctx_.set_dyn_batch(nullptr);
        return out;
    }

private:
    void fill(Tensor& dst, const py::array& src) {
        Tensor h = host_tensor(src, "history");
        BM_ASSERT(h.nbytes() <= dst.nbytes() && ctx_.is_BSHD(), "history rows: (n, hkv, d) into a BSHD buffer");
        BM_CUDART_ASSERT(hipMemcpy(dst.data(), h.data(), h.nbytes(), hipMemcpyHostToDevice));
    }
    model::ModelConfig cfg_;
    DummyModel md_;
    model::ModelContext ctx_;
    int num_layers_;
    std::unique_ptr<nn::Attention> attn_;
    std::shared_ptr<model::RagBufferContext> rag_;
};

}  // namespace

void bind_ref_attention(py::module_& m) {
    py::class_<RefAttention>(m, "RefAttention")
        .def(py::init<int, int, int, int, float, const std::string&, int, int, int, bool, int, const std::vector<int>&>(), py::arg("dim_model"), py::arg("num_heads"),
             py::arg("num_kv_heads"), py::arg("dim_head"), py::arg("rope_theta") = 10000.0f, py::arg("model_type") = "llama", py::arg("quant_type") = 5,
             py::arg("group_size") = 128, py::arg("num_layers") = 1, py::arg("bshd") = true, py::arg("device") = 0, py::arg("mla") = std::vector<int>())
        .def("latent_cache", &RefAttention::latent_cache)
        .def("load", &RefAttention::load)
        .def("set_history", &RefAttention::set_history)
        .def("get_k", &RefAttention::get_k)
        .def("get_k_scale", &RefAttention::get_k_scale)
        .def("get_v_scale", &RefAttention::get_v_scale)
        .def("set_history_quant", &RefAttention::set_history_quant)
        .def("cache_quant", &RefAttention::cache_quant)
        .def("get_v", &RefAttention::get_v)
        .def("encode", &RefAttention::encode, py::arg("layer"), py::arg("b"), py::arg("len_buf"), py::arg("hidden"), py::arg("pos0") = 0)
        .def("decode_step", &RefAttention::decode_step, py::arg("layer"), py::arg("hidden"), py::arg("positions"), py::arg("placement"), py::arg("mask"),
             py::arg("with_rope_cache") = false);
}

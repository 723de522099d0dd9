// ref_attention_glue.cpp -- the pybind11 harness class RefAttention: ONE reference nn::Attention layer (src/nn/attention/attention.cpp,
// compiled unmodified into libzhilight_amd_host.so next to the host library's own classes -- host_kvcache.cpp, host_position.cpp,
// host_attention_ext.cpp, the reference's model_context.cpp) driven from numpy: load the layer's weights under the reference's
// parameter names, fill per-task KV histories, run decode steps and prompt chunks through Attention::dyn_rag_forward ->
// NormalImpl::dynamic_batch_forward (attention.cpp:846-964) / MLAImpl.  tests/test_gpu_refcompile.py compares with the oracle.
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

// ---- the test class ------------------------------------------------------------------------------------------------------
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

// ref_block_glue.cpp -- the pybind11 harness classes RefEncoderLayer and RefFeedForward: a whole transformer layer of the
// reference -- EncoderLayer::forward (block.cpp:86-143): ln_attn -> Attention -> residual add -> ln_ff -> FeedForward -> residual
// add -- and its feed-forward flavours (dense, MOEImpl, GPTQMOE, FP8BlockMOE), compiled unmodified into libzhilight_amd_host.so,
// driven from numpy.  Test infrastructure: nothing in the product links this file.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime.h>

#include "bmengine/functions/all.h"
#include "model/dyn_batch_context.h"
#include "model/model.h"
#include "model/model_context.h"
#include "model/rag_buffer_context.h"
#include "nn/block/block.h"
#include "nn/feedforward/feedforward.h"
#include "nn/layernorm/layernorm.h"
#include "zhilight_amd.h"

namespace py = pybind11;
using bmengine::core::Context;
using bmengine::core::DataType;
using bmengine::core::Tensor;

#define ZL_OFF_PATH(what) \
    throw BMEngineException(std::string(what) + " is not on the dense decode path this module runs (SURVEY.md section 8)", __FILE__, __LINE__, __func__)
#define ZL_CK(call, what)                                                                                        \
    do {                                                                                                         \
        const int st_ = (call);                                                                                  \
        if (st_ != 0) throw BMEngineException(std::string(what) + ": " + zl_status_string(st_), __FILE__, __LINE__, __func__); \
    } while (0)

// ---- the test classes ----------------------------------------------------------------------------------------------------------
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

// One reference nn::EncoderLayer inside a reference ModelContext.
// mla = (q_lora_rank, kv_lora_rank, qk_nope_head_dim, qk_rope_head_dim, v_head_dim): the reference builds its MLAImpl and -- under
// LATENT_CACHE=1 -- the task buffers hold ONE (kv_lora_rank + qk_rope_head_dim)-wide latent row per key and no value buffer;
// moe = (num_experts, top_k, moe_intermediate_size, shared_expert_intermediate_size): FeedForward picks MOEImpl / GPTQMOE / FP8BlockMOE
// from its switches; bf16: the layer's dtype (hidden rows then travel as bf16 bits in int16 / uint16 arrays).  With all three and
// quant_type = FP8_Block this is a DeepSeek-V3-SHAPED layer: MLAImpl over Fp8Block linears followed by FP8BlockMOE.
class RefEncoderLayer {
public:
    RefEncoderLayer(int dim_model, int num_heads, int num_kv_heads, int dim_head, int dim_ff, float rope_theta, float eps, int quant_type, int group_size,
                    int device, const std::string& model_type, const std::vector<int>& mla, const std::vector<int>& moe, bool norm_topk_prob,
                    float routed_scaling_factor, bool bf16)
        : cfg_(make_cfg(model_type, dim_model, num_heads, num_kv_heads, dim_head, dim_ff, rope_theta, eps, mla, moe, norm_topk_prob, routed_scaling_factor, bf16)),
          md_(cfg_),
          ctx_(Context(device), md_, 1, false, true),
          bf16_(bf16) {
        model::QuantConfig qc(quant_type);
        qc.group_size = group_size;
        ctx_.set_current_layer(0);
        layer_.reset(new nn::EncoderLayer(ctx_, cfg_, qc, false));
        const bool latent = cfg_.kv_lora_rank > 0 && ctx_.latent_cache();
        BM_ASSERT(cfg_.kv_lora_rank == 0 || latent, "RefEncoderLayer: the MLA layer runs over the compressed cache here (LATENT_CACHE=1)");
        kvcache::KVCacheConfig kc{1, latent ? 1 : num_kv_heads, latent ? cfg_.kv_lora_rank + cfg_.qk_rope_head_dim : dim_head, cfg_.dtype, true, nullptr,
                                  std::vector<int>(1, device)};
        kvcache::KVCacheConfig vc = kc;
        if (latent) vc.dim_head = 0;
        rag_ = std::make_shared<model::RagBufferContext>(kc, vc);
        ctx_.set_rag_buffer(rag_);
    }
    static model::ModelConfig make_cfg(const std::string& model_type, int dim_model, int num_heads, int num_kv_heads, int dim_head, int dim_ff, float rope_theta,
                                       float eps, const std::vector<int>& mla, const std::vector<int>& moe, bool norm_topk_prob, float routed_scaling_factor,
                                       bool bf16) {
        model::ModelConfig c(model_type, 1, dim_model, num_heads, dim_head, dim_ff, 1024, eps, num_kv_heads, bf16 ? DataType::kBFloat16 : DataType::kHalf);
        c.rope_theta = rope_theta;
        if (mla.size() == 5 && mla[1] > 0) {
            c.q_lora_rank = mla[0]; c.kv_lora_rank = mla[1]; c.qk_nope_head_dim = mla[2]; c.qk_rope_head_dim = mla[3]; c.v_head_dim = mla[4];
        }
        if (moe.size() == 4 && moe[0] > 0) {
            c.moe_num_experts = moe[0]; c.moe_top_k = moe[1]; c.moe_intermediate_size = moe[2]; c.shared_expert_intermediate_size = moe[3];
            c.norm_topk_prob = norm_topk_prob; c.routed_scaling_factor = routed_scaling_factor;
        }
        return c;
    }
    void load(const std::map<std::string, py::array>& arrays, const std::string& prefix) {
        std::map<std::string, const Tensor> sd;
        for (auto& kv : arrays) sd.emplace(kv.first, host_tensor(kv.second, kv.first));
        layer_->load_state_dict(ctx_, sd, prefix, false);
    }
    void set_history(int b, int len_buf, const py::array& k, const py::array& v) {
        rag_->resize_task_buf(ctx_, b, (size_t)len_buf);
        fill(rag_->buf_k(b)[0], k);
        if (rag_->config_v_.dim_head > 0) fill(rag_->buf_v(b)[0], v);
    }
    py::array get_k(int b) { return out_array(rag_->buf_k(b, 0)); }
    py::array get_v(int b) { return out_array(rag_->buf_v(b, 0)); }
    bool latent_cache() { return ctx_.latent_cache(); }
    // one decode step: hidden (B, dim_model) fp16 (bf16 layer: the bits as int16 / uint16) -> the layer's output (B, dim_model), likewise
    py::array decode_step(const py::array& hidden, const py::array& positions, const py::array& placement, const py::array& mask) {
        const size_t B = (size_t)hidden.shape(0);
        auto dyn = std::make_shared<model::DynBatchContext>();
        dyn->s_placement = to_device(ctx_, placement, "s_placement").view({B, 1});
        dyn->s_position = to_device(ctx_, positions, "s_position");
        dyn->s_mask = to_device(ctx_, mask, "s_mask");
        for (size_t b = 0; b < B; ++b) dyn->sv_len_buf.push_back((int)rag_->get_buf_len(b));
        dyn->s_len_buf = ctx_.tensor_of(dyn->sv_len_buf);
        ctx_.set_dyn_batch(dyn);
        ctx_.set_current_layer(0);
        rag_->set_buffer_addr(ctx_);
        Tensor x = to_device(ctx_, hidden, "hidden");
        if (bf16_) {
            BM_ASSERT(x.dtype() == DataType::kInt16, "bf16 layer: hidden rows as int16 / uint16 bits");
            x = x.view_type(x.shape(), DataType::kBFloat16);
        }
        Tensor none;
        Tensor y = layer_->forward(ctx_, x, none, dyn->s_position, none, none, nullptr, nullptr, nullptr, nullptr);
        py::array out = out_array(y);
        ctx_.set_dyn_batch(nullptr);
        return out;
    }

private:
    py::array out_array(const Tensor& t) {
        if (t.dtype() == DataType::kBFloat16) {
            std::vector<py::ssize_t> shape(t.shape().begin(), t.shape().end());
            py::array out(py::dtype("uint16"), shape);
            t.to_buffer(out.mutable_data(), ctx_.current_cuda_stream());
            return out;
        }
        return to_numpy(ctx_, t);
    }
    void fill(Tensor& dst, const py::array& src) {
        if (src.nbytes() == 0) return;
        Tensor h = host_tensor(src, "history");
        BM_ASSERT(h.nbytes() <= dst.nbytes(), "history rows: (n, hkv, d) into a BSHD buffer");
        BM_CUDART_ASSERT(hipMemcpy(dst.data(), h.data(), h.nbytes(), hipMemcpyHostToDevice));
    }
    model::ModelConfig cfg_;
    DummyModel md_;
    model::ModelContext ctx_;
    bool bf16_;
    std::unique_ptr<nn::EncoderLayer> layer_;
    std::shared_ptr<model::RagBufferContext> rag_;
};

// One reference nn::FeedForward (dense, or the MoE implementations feedforward.cpp picks from its switches at construction:
// MOEImpl -- host routing, or its device dispatch route under MOE_GPU_DISPATCH_THRES -- and GPTQMOE under FUSE_GPTQ_MOE)
class RefFeedForward {
public:
    // moe = (num_experts, top_k, moe_intermediate_size, shared_expert_intermediate_size), zeros for the dense layer
    RefFeedForward(int dim_model, int dim_ff, const std::vector<int>& moe, bool norm_topk_prob, float routed_scaling_factor, int quant_type, int group_size,
                   int device, bool bf16)
        : cfg_("llama", 1, dim_model, 8, dim_model / 8, dim_ff, 1024, 1e-5f, 8, bf16 ? DataType::kBFloat16 : DataType::kHalf), ctx_(device), bf16_(bf16) {
        if (moe.size() == 4 && moe[0] > 0) {
            cfg_.moe_num_experts = moe[0]; cfg_.moe_top_k = moe[1]; cfg_.moe_intermediate_size = moe[2]; cfg_.shared_expert_intermediate_size = moe[3];
            cfg_.norm_topk_prob = norm_topk_prob; cfg_.routed_scaling_factor = routed_scaling_factor;
        }
        model::QuantConfig qc(quant_type);
        qc.group_size = group_size;
        ctx_.set_current_layer(0);
        ff_.reset(new nn::FeedForward(ctx_, cfg_, qc, false));
    }
    void load(const std::map<std::string, py::array>& arrays, const std::string& prefix) {
        std::map<std::string, const Tensor> sd;
        for (auto& kv : arrays) sd.emplace(kv.first, host_tensor(kv.second, kv.first));
        ff_->load_state_dict(ctx_, sd, prefix, false);
    }
    // x: float16, or -- bf16 layer -- the bf16 bits as int16 / uint16 (numpy has no bfloat16); the result likewise
    py::array forward(const py::array& x) {
        Tensor dx = to_device(ctx_, x, "x");
        if (bf16_ && dx.dtype() == DataType::kInt16) dx = dx.view_type(dx.shape(), DataType::kBFloat16);
        Tensor y = ff_->forward(ctx_, dx);
        if (y.dtype() == DataType::kBFloat16) {
            std::vector<py::ssize_t> shape(y.shape().begin(), y.shape().end());
            py::array out(py::dtype("uint16"), shape);
            y.to_buffer(out.mutable_data(), ctx_.current_cuda_stream());
            return out;
        }
        return to_numpy(ctx_, y);
    }

private:
    model::ModelConfig cfg_;
    Context ctx_;
    bool bf16_;
    std::unique_ptr<nn::FeedForward> ff_;
};

}  // namespace

void bind_ref_block(py::module_& m) {
    py::class_<RefFeedForward>(m, "RefFeedForward")
        .def(py::init<int, int, const std::vector<int>&, bool, float, int, int, int, bool>(), py::arg("dim_model"), py::arg("dim_ff"),
             py::arg("moe") = std::vector<int>(), py::arg("norm_topk_prob") = true, py::arg("routed_scaling_factor") = 1.0f, py::arg("quant_type") = 0,
             py::arg("group_size") = 128, py::arg("device") = 0, py::arg("bf16") = false)
        .def("load", &RefFeedForward::load)
        .def("forward", &RefFeedForward::forward);
    py::class_<RefEncoderLayer>(m, "RefEncoderLayer")
        .def(py::init<int, int, int, int, int, float, float, int, int, int, const std::string&, const std::vector<int>&, const std::vector<int>&, bool, float, bool>(),
             py::arg("dim_model"), py::arg("num_heads"), py::arg("num_kv_heads"),
             py::arg("dim_head"), py::arg("dim_ff"), py::arg("rope_theta") = 10000.0f, py::arg("eps") = 1e-5f, py::arg("quant_type") = 5,
             py::arg("group_size") = 128, py::arg("device") = 0, py::arg("model_type") = "llama", py::arg("mla") = std::vector<int>(),
             py::arg("moe") = std::vector<int>(), py::arg("norm_topk_prob") = true, py::arg("routed_scaling_factor") = 1.0f, py::arg("bf16") = false)
        .def("latent_cache", &RefEncoderLayer::latent_cache)
        .def("load", &RefEncoderLayer::load)
        .def("set_history", &RefEncoderLayer::set_history)
        .def("get_k", &RefEncoderLayer::get_k)
        .def("get_v", &RefEncoderLayer::get_v)
        .def("decode_step", &RefEncoderLayer::decode_step);
}

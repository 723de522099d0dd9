// ref_block_glue.cpp -- what is linked next to the REFERENCE's own src/nn/block/block.cpp and src/nn/feedforward/feedforward.cpp
// (compiled unmodified, zhilight_amd/build.py: build_refcompile) so that a whole transformer layer of the reference --
// EncoderLayer::forward (block.cpp:86-143): ln_attn -> Attention (attention.cpp, see ref_attention_glue.cpp) -> residual add ->
// ln_ff -> FeedForward (feedforward.cpp) -> residual add -- EXECUTES a decode step on the MI355X boundary.
//   1. nn::LayerNorm with the REFERENCE's class layout (src/nn/layernorm/layernorm.h:7-34: a core::Layer with a pimpl; the
//      boundary's own nn::LayerNorm in nn_amd.h is a different class under the same name, so nn_amd.cpp leaves its definitions out
//      of this module: -DZL_REF_LAYERNORM_EXTERNAL) over zl_rmsnorm;
//   2. the four ModelContext members block.cpp calls (model_context.cpp:133-136, 221-242, 328-341) for ONE rank;
//   3. bmengine::functions helpers only the smooth-quant calibration uses (pow, clamp): declared by the shim, not on this path --
//      definitions that throw (the MoE dispatch route's arange / sort_pair_1d / divide / scatter_update_dim0 are real since
//      round 4: bm_functions.cpp over zl_arange_i32 / zl_sort_pairs_i32 / zl_divide_i32 / zl_scatter_update_dim0);
//   4. the pybind11 class RefEncoderLayer: load a layer under the reference's parameter names, fill KV histories, run decode steps.
// Test infrastructure: nothing in the product links this file.
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

// ---- 1. nn::LayerNorm, the reference's class ----------------------------------------------------------------------------------
namespace nn {

class LayerNorm::impl {
public:
    int dim_model;
    float eps, scale;
    core::DataType dtype;
    int num_head;
    bool rms = true;
    core::Tensor weight;
    impl(int dim_model, float eps, float scale, core::DataType dtype, int num_head)
        : dim_model(dim_model), eps(eps), scale(scale), dtype(dtype), num_head(num_head) {}
    int zdt(const core::Tensor& x) const { return x.dtype() == DataType::kHalf ? ZL_F16 : ZL_BF16; }
    void check(const core::Tensor& x) const {
        BM_ASSERT(rms && num_head == 1, "LayerNorm: the RMS form over the whole row is what this path uses");
        BM_ASSERT_EQ((int)x.size(-1), dim_model, "LayerNorm: dim mismatch");
        BM_ASSERT(weight.numel() == (size_t)dim_model, "LayerNorm: weight not loaded");
    }
    // rows of a possibly STRIDED (rows, dim) operand into a possibly strided output: zl_head_norm with one "head" per row
    void rows_into(const core::Context& ctx, const core::Tensor& x, core::Tensor& out) const {
        check(x);
        BM_ASSERT(x.ndim() == 2 && out.ndim() == 2 && x.stride(1) == 1 && out.stride(1) == 1 && out.size(0) == x.size(0) && out.size(1) == x.size(1),
                  "LayerNorm::forward_2: (rows, dim) operands, dense last dimension");
        BM_ASSERT(scale == 1.0f && dim_model <= 1024, "LayerNorm::forward_2: scale 1, dim <= 1024");
        ZL_CK(zl_head_norm(x.data<uint16_t>(), weight.data<uint16_t>(), out.data<uint16_t>(), x.size(0), 1, dim_model, x.stride(0), out.stride(0), eps, 0, zdt(x),
                           (zl_stream_t)ctx.current_cuda_stream()), "head_norm(rows)");
    }
};

LayerNorm::LayerNorm(const core::Context&, int dim_model, bool quant, float eps, float scale, core::DataType dtype, int num_head)
    : pimpl(new impl(dim_model, eps, scale, dtype, num_head)) {
    BM_ASSERT(!quant, "LayerNorm with fused int8 quantisation: Int8Linear's route, not this module's");
    add_parameter("weight", pimpl->weight);
}
LayerNorm::~LayerNorm() = default;
void LayerNorm::set_rms(bool b) { pimpl->rms = b; }
void LayerNorm::load_state_dict(const core::Context& ctx, const std::map<std::string, const core::Tensor>& state_dict, const std::string& prefix,
                                bool allow_missing) {
    auto it = state_dict.find(prefix + ".weight");
    if (it == state_dict.end()) {
        BM_ASSERT(allow_missing, "missing parameter " + prefix + ".weight");
        return;
    }
    BM_ASSERT_EQ((int)it->second.numel(), pimpl->dim_model, "layernorm weight size mismatch");
    pimpl->weight = ctx.cuda(it->second);
}
core::Tensor LayerNorm::forward(const core::Context& ctx, const core::Tensor& x) {
    pimpl->check(x);
    core::Tensor out = ctx.tensor(x.shape(), x.dtype());
    ZL_CK(zl_rmsnorm(x.data<uint16_t>(), pimpl->weight.data<uint16_t>(), out.data<uint16_t>(), x.numel() / x.size(-1), pimpl->dim_model, pimpl->eps,
                     pimpl->scale, nullptr, nullptr, pimpl->zdt(x), (zl_stream_t)ctx.current_cuda_stream()), "rmsnorm");
    return out;
}
core::Tensor LayerNorm::fuse_add(const core::Context& ctx, const core::Tensor& a, const core::Tensor& b, core::Tensor& c) {
    pimpl->check(a);
    BM_ASSERT_EQ(a.numel(), b.numel(), "shape mismatch");
    if (c.numel() == 0) c = ctx.tensor(a.shape(), a.dtype());
    core::Tensor out = ctx.tensor(a.shape(), a.dtype());
    ZL_CK(zl_rmsnorm(a.data<uint16_t>(), pimpl->weight.data<uint16_t>(), out.data<uint16_t>(), a.numel() / a.size(-1), pimpl->dim_model, pimpl->eps,
                     pimpl->scale, b.data<uint16_t>(), c.data<uint16_t>(), pimpl->zdt(a), (zl_stream_t)ctx.current_cuda_stream()), "rmsnorm(fuse_add)");
    return out;
}
void LayerNorm::inplace(const core::Context& ctx, core::Tensor& x) {
    pimpl->check(x);
    ZL_CK(zl_rmsnorm(x.data<uint16_t>(), pimpl->weight.data<uint16_t>(), x.data<uint16_t>(), x.numel() / x.size(-1), pimpl->dim_model, pimpl->eps,
                     pimpl->scale, nullptr, nullptr, pimpl->zdt(x), (zl_stream_t)ctx.current_cuda_stream()), "rmsnorm(inplace)");
}
// two norms in one call (MLAImpl's q_a / kv_a norms, multi_head_latent_attention.cpp:526): inputs AND outputs may be last-dimension
// slices of wider tensors (the fused qkv_a projection; the compressed_kv row under construction) and the outputs are written in
// place when the caller hands them allocated -- rows through zl_head_norm (one "head" per row, explicit row strides)
void LayerNorm::forward_2(const core::Context& ctx, core::Tensor& x, core::Tensor& y, core::Tensor& x_out, core::Tensor& y_out, LayerNorm* la, LayerNorm* lb) {
    if (x_out.numel() == 0) x_out = ctx.tensor(x.shape(), x.dtype());
    if (y_out.numel() == 0) y_out = ctx.tensor(y.shape(), y.dtype());
    la->pimpl->rows_into(ctx, x, x_out);
    lb->pimpl->rows_into(ctx, y, y_out);
}

}  // namespace nn

// ---- 2. ModelContext members block.cpp calls ----------------------------------------------------------------------------------
namespace model {
void ModelContext::copy2(const Tensor& src, Tensor* dst) {
    BM_CUDART_ASSERT(hipMemcpyAsync(dst->data(), src.data(), src.nbytes(), hipMemcpyDeviceToDevice, current_cuda_stream()));
}
void ModelContext::reduce_sum2(const Tensor& data, Tensor* out, DataType, bool) const {
    BM_ASSERT(world_size() == 1, "ModelContext::reduce_sum2: one rank here (the exchange step is libzhilight_amd_comm.so's)");
    if (out->numel() == 0) *out = tensor(data.shape(), data.dtype());
    BM_CUDART_ASSERT(hipMemcpyAsync(out->data(), data.data(), data.nbytes(), hipMemcpyDeviceToDevice, current_cuda_stream()));
}
void ModelContext::update_act_scale(const std::string&, const Tensor&) { ZL_OFF_PATH("ModelContext::update_act_scale (smooth-quant calibration)"); }
void ModelContext::check_numeric(const core::Tensor& t) const {
    if (checking_numerics()) bmengine::functions::check_numeric(*this, t);
}
}  // namespace model

// ---- 3. functions helpers off this path ---------------------------------------------------------------------------------------
namespace bmengine {
namespace functions {
core::Tensor pow(const core::Context&, const core::Tensor&, float) { ZL_OFF_PATH("functions::pow (smooth-quant calibration)"); }
core::Tensor clamp(const core::Context&, const core::Tensor&, float, float) { ZL_OFF_PATH("functions::clamp (smooth-quant calibration)"); }
}  // namespace functions
}  // namespace bmengine

// ---- 4. the test class ----------------------------------------------------------------------------------------------------------
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

// One reference nn::EncoderLayer inside a reference ModelContext
class RefEncoderLayer {
public:
    RefEncoderLayer(int dim_model, int num_heads, int num_kv_heads, int dim_head, int dim_ff, float rope_theta, float eps, int quant_type, int group_size,
                    int device)
        : cfg_("llama", 1, dim_model, num_heads, dim_head, dim_ff, 1024, eps, num_kv_heads, DataType::kHalf),
          md_((cfg_.rope_theta = rope_theta, cfg_)),
          ctx_(Context(device), md_, 1, false, true) {
        model::QuantConfig qc(quant_type);
        qc.group_size = group_size;
        ctx_.set_current_layer(0);
        layer_.reset(new nn::EncoderLayer(ctx_, cfg_, qc, false));
        kvcache::KVCacheConfig kc{1, num_kv_heads, dim_head, DataType::kHalf, true, nullptr, std::vector<int>(1, device)};
        rag_ = std::make_shared<model::RagBufferContext>(kc, kc);
        ctx_.set_rag_buffer(rag_);
    }
    void load(const std::map<std::string, py::array>& arrays, const std::string& prefix) {
        std::map<std::string, const Tensor> sd;
        for (auto& kv : arrays) sd.emplace(kv.first, host_tensor(kv.second, kv.first));
        layer_->load_state_dict(ctx_, sd, prefix, false);
    }
    void set_history(int b, int len_buf, const py::array& k, const py::array& v) {
        rag_->resize_task_buf(ctx_, b, (size_t)len_buf);
        fill(rag_->buf_k(b)[0], k);
        fill(rag_->buf_v(b)[0], v);
    }
    py::array get_k(int b) { return to_numpy(ctx_, rag_->buf_k(b, 0)); }
    py::array get_v(int b) { return to_numpy(ctx_, rag_->buf_v(b, 0)); }
    // one decode step: hidden (B, dim_model) fp16 -> the layer's output (B, dim_model)
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
        Tensor none;
        Tensor y = layer_->forward(ctx_, x, none, dyn->s_position, none, none, nullptr, nullptr, nullptr, nullptr);
        py::array out = to_numpy(ctx_, y);
        ctx_.set_dyn_batch(nullptr);
        return out;
    }

private:
    void fill(Tensor& dst, const py::array& src) {
        Tensor h = host_tensor(src, "history");
        BM_ASSERT(h.nbytes() <= dst.nbytes(), "history rows: (n, hkv, d) into a BSHD buffer");
        BM_CUDART_ASSERT(hipMemcpy(dst.data(), h.data(), h.nbytes(), hipMemcpyHostToDevice));
    }
    model::ModelConfig cfg_;
    DummyModel md_;
    model::ModelContext ctx_;
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
        .def(py::init<int, int, int, int, int, float, float, int, int, int>(), py::arg("dim_model"), py::arg("num_heads"), py::arg("num_kv_heads"),
             py::arg("dim_head"), py::arg("dim_ff"), py::arg("rope_theta") = 10000.0f, py::arg("eps") = 1e-5f, py::arg("quant_type") = 5,
             py::arg("group_size") = 128, py::arg("device") = 0)
        .def("load", &RefEncoderLayer::load)
        .def("set_history", &RefEncoderLayer::set_history)
        .def("get_k", &RefEncoderLayer::get_k)
        .def("get_v", &RefEncoderLayer::get_v)
        .def("decode_step", &RefEncoderLayer::decode_step);
}

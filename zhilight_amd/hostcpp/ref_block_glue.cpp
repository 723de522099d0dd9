// ref_block_glue.cpp -- the pybind11 harness classes RefEncoderLayer and RefFeedForward: a whole transformer layer of the
// reference -- EncoderLayer::forward (block.cpp:86-143): ln_attn -> Attention -> residual add -> ln_ff -> FeedForward -> residual
// add -- and its feed-forward flavours (dense, MOEImpl, GPTQMOE, FP8BlockMOE), compiled unmodified into libzhilight_amd_host.so,
// driven from numpy.  Test infrastructure: nothing in the product links this file.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime.h>

#include <cstring>

#include "bm_engine.h"
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

// The same layer TENSOR-PARALLEL on a core::Engine (one thread per rank; `devices` may repeat one device): the reference's
// ModelContext::create on every rank's thread, EncoderLayer(parallel = true).  With the DeepSeek configuration this is config 5's
// sharding on one box: ATTN_DATA_PARALLEL=1 -- MLAImpl::forward_compressed_dp_v1 (multi_head_latent_attention.cpp:1097-1232): the
// compressed cache replicated, the decode tasks dealt to the ranks, full-width q_b / absorbed k / v / o projections on the rank's
// tasks, the layer's reduce_sum assembling the rows -- or head-parallel MLA; MOE_EXP_PARALLEL=1 -- experts e % world == rank, the
// shared expert tensor-parallel, FeedForward::route's broadcasts -- with every exchange on the engine's transports.
class RefEngineEncoderLayer {
public:
    RefEngineEncoderLayer(int dim_model, int num_heads, int num_kv_heads, int dim_head, int dim_ff, float rope_theta, float eps, int quant_type, int group_size,
                          const std::string& model_type, const std::vector<int>& mla, const std::vector<int>& moe, bool norm_topk_prob,
                          float routed_scaling_factor, bool bf16, const std::vector<int>& devices)
        : cfg_(RefEncoderLayer::make_cfg(model_type, dim_model, num_heads, num_kv_heads, dim_head, dim_ff, rope_theta, eps, mla, moe, norm_topk_prob,
                                         routed_scaling_factor, bf16)),
          md_(cfg_),
          bf16_(bf16) {
        std::vector<bmengine::core::DeviceConfiguration> dc;
        for (int d : devices) dc.emplace_back(d, (size_t)0);
        engine_.reset(new bmengine::core::Engine(dc));
        ranks_.resize(devices.size());
        model::QuantConfig qc(quant_type);
        qc.group_size = group_size;
        model::DynBatchConfig bc;
        bc.rag_buffer = true;
        bc.flash_attention = true;
        engine_->device_foreach([&](int r) {
            Rank& R = ranks_[r];
            R.ctx.reset(new model::ModelContext(model::ModelContext::create(*engine_, md_, bc, r, true)));
            R.ctx->set_current_layer(0);
            R.layer.reset(new nn::EncoderLayer(*R.ctx, cfg_, qc, true));
        });
    }
    ~RefEngineEncoderLayer() {
        try {
            engine_->device_foreach([&](int r) {
                (void)hipDeviceSynchronize();
                ranks_[r].layer.reset();
                ranks_[r].ctx.reset();
            });
        } catch (...) {
        }
    }
    int world_size() const { return engine_->world_size(); }
    std::vector<int> exchange_errors() {
        std::vector<int> e(ranks_.size());
        engine_->device_foreach([&](int r) { e[r] = engine_->exchange_errors(r); });
        return e;
    }
    void load(const std::map<std::string, py::array>& arrays, const std::string& prefix) {
        std::map<std::string, const Tensor> sd;
        for (auto& kv : arrays) sd.emplace(kv.first, host_tensor(kv.second, kv.first));
        engine_->device_foreach([&](int r) { ranks_[r].layer->load_state_dict(*ranks_[r].ctx, sd, prefix, false); });
    }
    // the compressed cache is REPLICATED: every rank gets task b's rows k (n, 1, kv_lora_rank + rope) in a buffer of len_buf rows
    void set_history(int b, int len_buf, const py::array& k) {
        Tensor hk = host_tensor(k, "k");
        engine_->device_foreach([&](int r) {
            Rank& R = ranks_[r];
            auto rag = R.ctx->rag_buffer();
            BM_ASSERT(rag->config_v_.dim_head == 0, "RefEngineEncoderLayer: the compressed (latent) cache only (LATENT_CACHE=1)");
            rag->resize_task_buf(*R.ctx, b, (size_t)len_buf);
            Tensor& dst = rag->buf_k(b)[0];
            BM_ASSERT(hk.nbytes() <= dst.nbytes(), "history rows into the task buffer");
            if (hk.nbytes()) BM_CUDART_ASSERT(hipMemcpy(dst.data(), hk.data(), hk.nbytes(), hipMemcpyHostToDevice));
        });
    }
    // one decode step on every rank: hidden (B, dim_model) 16-bit rows -> the ranks' outputs (world, B, dim_model) as uint16 bits
    py::array decode_step(const py::array& hidden, const py::array& positions, const py::array& placement, const py::array& mask) {
        const size_t B = (size_t)hidden.shape(0), world = ranks_.size(), dm = (size_t)cfg_.dim_model;
        Tensor hx = host_tensor(hidden, "hidden"), hp = host_tensor(positions, "s_position"), hpl = host_tensor(placement, "s_placement"),
               hm = host_tensor(mask, "s_mask");
        BM_ASSERT(hx.numel() == B * dm && bmengine::core::get_elem_size(hx.dtype()) == 2 && hp.dtype() == DataType::kInt32, "hidden (B, dim_model) 16-bit, int32 positions");
        std::vector<int> pos_host((const int*)hp.data(), (const int*)hp.data() + B);
        std::vector<std::vector<uint16_t>> out(world, std::vector<uint16_t>(B * dm));
        engine_->device_foreach([&](int r) {
            Rank& R = ranks_[r];
            model::ModelContext& ctx = *R.ctx;
            auto rag = ctx.rag_buffer();
            auto dyn = std::make_shared<model::DynBatchContext>();
            dyn->s_placement = upload(ctx, hpl).view({B, 1});
            dyn->s_position = upload(ctx, hp);
            dyn->sv_position = pos_host;                                  // (MLAImpl::attn_by_flash_mla reads the host copy)
            dyn->s_mask = upload(ctx, hm);
            for (size_t b = 0; b < B; ++b) dyn->sv_len_buf.push_back((int)rag->get_buf_len(b));
            dyn->s_len_buf = ctx.tensor_of(dyn->sv_len_buf);
            ctx.set_dyn_batch(dyn);
            ctx.set_current_layer(0);
            rag->set_buffer_addr(ctx);
            Tensor x = upload(ctx, hx);
            x = x.view_type({B, dm}, bf16_ ? DataType::kBFloat16 : DataType::kHalf);
            Tensor none;
            Tensor y = R.layer->forward(ctx, x, none, dyn->s_position, none, none, nullptr, nullptr, nullptr, nullptr);
            BM_ASSERT_EQ(y.numel(), B * dm, "layer output (B, dim_model)");
            y.to_buffer(out[r].data(), ctx.current_cuda_stream());
            ctx.set_dyn_batch(nullptr);
        });
        py::array res(py::dtype("uint16"), std::vector<py::ssize_t>{(py::ssize_t)world, (py::ssize_t)B, (py::ssize_t)dm});
        for (size_t r = 0; r < world; ++r) std::memcpy((char*)res.mutable_data() + r * B * dm * 2, out[r].data(), B * dm * 2);
        return res;
    }
    // rank r's compressed-cache rows of task b: (len_buf, 1, kv_lora_rank + rope) as uint16 bits
    py::array get_k(int r, int b) {
        std::vector<uint16_t> host;
        std::vector<size_t> shape;
        engine_->run(r, [&] {
            const Tensor& t = ranks_[r].ctx->rag_buffer()->buf_k(b, 0);
            shape = t.shape();
            host.resize(t.numel());
            t.to_buffer(host.data(), ranks_[r].ctx->current_cuda_stream());
        });
        py::array out(py::dtype("uint16"), std::vector<py::ssize_t>(shape.begin(), shape.end()));
        std::memcpy(out.mutable_data(), host.data(), host.size() * 2);
        return out;
    }

private:
    struct Rank {
        std::unique_ptr<model::ModelContext> ctx;
        std::unique_ptr<nn::EncoderLayer> layer;
    };
    static Tensor upload(const Context& ctx, const Tensor& host) {
        Tensor d = ctx.tensor(host.shape(), host.dtype());
        d.from_buffer(host.data(), false, ctx.current_cuda_stream());
        return d;
    }
    model::ModelConfig cfg_;
    DummyModel md_;
    bool bf16_;
    std::unique_ptr<bmengine::core::Engine> engine_;
    std::vector<Rank> ranks_;
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
    py::class_<RefEngineEncoderLayer>(m, "RefEngineEncoderLayer")
        .def(py::init<int, int, int, int, int, float, float, int, int, const std::string&, const std::vector<int>&, const std::vector<int>&, bool, float, bool,
                      const std::vector<int>&>(),
             py::arg("dim_model"), py::arg("num_heads"), py::arg("num_kv_heads"), py::arg("dim_head"), py::arg("dim_ff"), py::arg("rope_theta") = 10000.0f,
             py::arg("eps") = 1e-5f, py::arg("quant_type") = 5, py::arg("group_size") = 128, py::arg("model_type") = "llama",
             py::arg("mla") = std::vector<int>(), py::arg("moe") = std::vector<int>(), py::arg("norm_topk_prob") = true,
             py::arg("routed_scaling_factor") = 1.0f, py::arg("bf16") = false, py::arg("devices") = std::vector<int>{0, 0})
        .def("world_size", &RefEngineEncoderLayer::world_size)
        .def("exchange_errors", &RefEngineEncoderLayer::exchange_errors)
        .def("load", &RefEngineEncoderLayer::load)
        .def("set_history", &RefEngineEncoderLayer::set_history)
        .def("get_k", &RefEngineEncoderLayer::get_k)
        .def("decode_step", &RefEngineEncoderLayer::decode_step);
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

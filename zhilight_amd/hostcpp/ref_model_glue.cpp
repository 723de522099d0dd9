// ref_model_glue.cpp -- what is linked next to the REFERENCE's own src/model/llama.cpp (compiled unmodified, zhilight_amd/build.py:
// build_refcompile) so that the path BASELINE.json's north star names from its TOP -- LLaMA::encode (src/model/llama.cpp:75-151)
// -> EncoderLayer::forward (block.cpp) -> Attention (attention.cpp) / FeedForward (feedforward.cpp) -> Linear (linear.cpp), then
// get_logits (llama.cpp:159-165) -- EXECUTES decode steps on the MI355X boundary, five reference units deep.
//   1. nn::RawEmbedding (src/nn/embedding/embedding.h:24-47; embedding.cu:260-289): token lookup = zl_embedding, projection = the
//      lm_head product (zl_gemm_nt_small_m / zl_gemm_nt);
//   2. nn::RopePreparer (src/nn/position/rope_preparer.h): the cos / sin tables of ROPE_CACHE=1 = zl_rope_cos_sin*;
//   3. names of llama.cpp's loss / scoring helpers that are not on the decode path: definitions that throw;
//   4. the pybind11 class RefLLaMA: load a whole model under the reference's parameter names, fill KV histories, run decode steps
//      and read the logits.  tests/test_gpu_refcompile.py compares with the SAME CPU oracle the repository's own LLaMA is held to.
// Test infrastructure: nothing in the product links this file.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime.h>

#include "model/dyn_batch_context.h"
#include "model/llama.h"
#include "model/model_context.h"
#include "model/model_util.h"
#include "model/rag_buffer_context.h"
#include "nn/embedding/embedding.h"
#include "nn/position/rope_preparer.h"
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

namespace nn {

// ---- 1. RawEmbedding ----------------------------------------------------------------------------------------------------------
class RawEmbedding::impl {
public:
    int dim_model, vocab_size;
    core::DataType dtype;
    float scale = 1.0f, logit_scale = 1.0f;
    core::Tensor weight;
    int zdt() const { return dtype == DataType::kHalf ? ZL_F16 : ZL_BF16; }
};
RawEmbedding::RawEmbedding(const core::Context& ctx, int dim_model, int vocab_size, bool scale_weights, core::DataType dtype, bool parallel)
    : pimpl(new impl) {
    BM_ASSERT(!parallel || ctx.world_size() == 1, "RawEmbedding: one rank here");
    pimpl->dim_model = dim_model;
    pimpl->vocab_size = vocab_size;
    pimpl->dtype = dtype;
    if (scale_weights) pimpl->scale = 1.0f / sqrtf((float)dim_model);
    pimpl->weight = ctx.parameter({(size_t)vocab_size, (size_t)dim_model}, dtype);
    add_parameter("weight", pimpl->weight);
}
RawEmbedding::~RawEmbedding() = default;
void RawEmbedding::set_scale_weights(bool b) { pimpl->scale = b ? 1.0f / sqrtf((float)pimpl->dim_model) : 1.0f; }
void RawEmbedding::set_scale_factor(float b) { pimpl->scale = b; }
void RawEmbedding::set_logit_scale(float b) { pimpl->logit_scale = b; }
void RawEmbedding::load_state_dict(const core::Context& ctx, const std::map<std::string, const core::Tensor>& state_dict, const std::string& prefix,
                                   bool allow_missing) {
    core::Layer::load_state_dict(ctx, state_dict, prefix, allow_missing);
}
core::Tensor RawEmbedding::forward(const core::Context& ctx, const core::Tensor& ids) {
    BM_ASSERT(ids.dtype() == DataType::kInt32, "token ids are int32");
    const size_t n = ids.numel();
    std::vector<size_t> shape = ids.shape();
    shape.push_back((size_t)pimpl->dim_model);
    core::Tensor out = ctx.tensor(shape, pimpl->dtype);
    ZL_CK(zl_embedding(ids.data<int32_t>(), pimpl->weight.data<uint16_t>(), out.data<uint16_t>(), n, pimpl->dim_model, 0, pimpl->vocab_size, pimpl->scale,
                       pimpl->zdt(), (zl_stream_t)ctx.current_cuda_stream()), "embedding");
    return out;
}
core::Tensor RawEmbedding::projection(const core::Context& ctx, const core::Tensor& input) {
    const int64_t m = input.numel() / input.size(-1), k = input.size(-1), n = pimpl->vocab_size;
    BM_ASSERT_EQ(k, (int64_t)pimpl->dim_model, "RawEmbedding::projection: dim mismatch");
    std::vector<size_t> shape = input.shape();
    shape.back() = (size_t)n;
    core::Tensor out = ctx.tensor(shape, pimpl->dtype);
    const float alpha = pimpl->logit_scale;
    if (m <= 4)
        ZL_CK(zl_gemm_nt_small_m(input.data<uint16_t>(), k, pimpl->weight.data<uint16_t>(), nullptr, out.data<uint16_t>(), m, n, k, alpha, pimpl->zdt(), nullptr, 0.f,
                                 (zl_stream_t)ctx.current_cuda_stream()), "lm_head (row-streaming)");
    else
        ZL_CK(zl_gemm_nt(input.data<uint16_t>(), k, pimpl->weight.data<uint16_t>(), nullptr, out.data<uint16_t>(), m, n, k, alpha, pimpl->zdt(),
                         (zl_stream_t)ctx.current_cuda_stream()), "lm_head");
    return out;
}

// ---- 2. RopePreparer ----------------------------------------------------------------------------------------------------------
class RopePreparer::impl {
public:
    model::ModelConfig cfg;
    explicit impl(const model::ModelConfig& c) : cfg(c) {}
};
RopePreparer::RopePreparer(const core::Context&, model::ModelConfig cfg) : pimpl(new impl(cfg)) {}
RopePreparer::~RopePreparer() = default;
std::tuple<core::Tensor, core::Tensor> RopePreparer::forward(const core::Context& ctx, const core::Tensor&, const core::Tensor& pos) {
    const model::ModelConfig& c = pimpl->cfg;
    const size_t n = pos.numel(), d = c.dim_head;
    core::Tensor cs = ctx.tensor({n, d}, DataType::kFloat), sn = ctx.tensor({n, d}, DataType::kFloat);
    zl_stream_t st = (zl_stream_t)ctx.current_cuda_stream();
    if (c.rope_cfg.type == "llama3")
        ZL_CK(zl_rope_cos_sin_llama3(pos.data<int32_t>(), cs.data<float>(), sn.data<float>(), n, d, c.rope_theta, c.rope_cfg.factor, c.rope_cfg.low_freq_factor,
                                     c.rope_cfg.high_freq_factor, (float)c.rope_cfg.original_max_position, c.rope_cfg.neox_style ? 1 : 0, st), "rope_cos_sin_llama3");
    else
        ZL_CK(zl_rope_cos_sin(pos.data<int32_t>(), cs.data<float>(), sn.data<float>(), n, d, c.rope_theta, c.rope_cfg.neox_style ? 1 : 0, st), "rope_cos_sin");
    return std::make_tuple(cs, sn);
}

// ---- 3. off-path names ----------------------------------------------------------------------------------------------------------
std::tuple<float, core::Tensor> log_prob_raw(const core::Context&, const core::Tensor&, const core::Tensor&, int32_t) { ZL_OFF_PATH("nn::log_prob_raw (scoring)"); }
int greedy_match_raw(const core::Context&, const core::Tensor&, const core::Tensor&, int32_t) { ZL_OFF_PATH("nn::greedy_match_raw (scoring)"); }
std::tuple<float, core::Tensor> cross_entropy_raw(const core::Context&, const core::Tensor&, const core::Tensor&, int32_t, float) {
    ZL_OFF_PATH("nn::cross_entropy_raw (loss)");
}

}  // namespace nn

namespace model {
core::Tensor convert_fp32(const core::Context& ctx, const core::Tensor& logits) { return bmengine::functions::typecast(ctx, logits, DataType::kFloat); }
}  // namespace model

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

// The reference's model::LLaMA inside the reference's ModelContext
class RefLLaMA {
public:
    RefLLaMA(int num_layers, int dim_model, int num_heads, int num_kv_heads, int dim_head, int dim_ff, int vocab_size, float eps, float rope_theta,
             int quant_type, int group_size, int device)
        : cfg_(make_cfg(num_layers, dim_model, num_heads, num_kv_heads, dim_head, dim_ff, vocab_size, eps, rope_theta)),
          md_(cfg_),
          ctx_(Context(device), md_, 1, false, true) {
        model::QuantConfig qc(quant_type);
        qc.group_size = group_size;
        model_.reset(new model::LLaMA(ctx_, cfg_, qc, false));
        kvcache::KVCacheConfig kc{num_layers, num_kv_heads, dim_head, DataType::kHalf, true, nullptr, std::vector<int>(num_layers, device)};
        rag_ = std::make_shared<model::RagBufferContext>(kc, kc);
        ctx_.set_rag_buffer(rag_);
    }
    static model::ModelConfig make_cfg(int num_layers, int dim_model, int num_heads, int num_kv_heads, int dim_head, int dim_ff, int vocab_size, float eps,
                                       float rope_theta) {
        model::ModelConfig c("llama", num_layers, dim_model, num_heads, dim_head, dim_ff, vocab_size, eps, num_kv_heads, DataType::kHalf);
        c.rope_theta = rope_theta;
        return c;
    }
    void load(const std::map<std::string, py::array>& arrays, const std::string& prefix) {
        std::map<std::string, const Tensor> sd;
        for (auto& kv : arrays) sd.emplace(kv.first, host_tensor(kv.second, kv.first));
        model_->load_state_dict(ctx_, sd, prefix, false);
    }
    // task b, every layer: k / v (num_layers, n, hkv, d) fill rows 0 .. n - 1 of buffers of len_buf rows
    void set_history(int b, int len_buf, const py::array& k, const py::array& v) {
        rag_->resize_task_buf(ctx_, b, (size_t)len_buf);
        const size_t layers = (size_t)k.shape(0), per = (size_t)k.nbytes() / layers;
        for (size_t l = 0; l < layers; ++l) {
            BM_CUDART_ASSERT(hipMemcpy(rag_->buf_k(b)[(int)l].data(), (const char*)k.data() + l * per, per, hipMemcpyHostToDevice));
            BM_CUDART_ASSERT(hipMemcpy(rag_->buf_v(b)[(int)l].data(), (const char*)v.data() + l * per, per, hipMemcpyHostToDevice));
        }
    }
    // one decode step of the whole model: tokens (B) int32 at positions (B) -> logits (B, vocab)
    py::array decode_step(const py::array& tokens, const py::array& positions, const py::array& mask) {
        const size_t B = (size_t)tokens.shape(0);
        auto dyn = std::make_shared<model::DynBatchContext>();
        dyn->s_token = to_device(ctx_, tokens, "s_token");
        dyn->s_position = to_device(ctx_, positions, "s_position");
        dyn->s_placement = to_device(ctx_, positions, "s_placement").view({B, 1});
        dyn->s_mask = to_device(ctx_, mask, "s_mask");
        for (size_t b = 0; b < B; ++b) dyn->sv_len_buf.push_back((int)rag_->get_buf_len(b));
        dyn->s_len_buf = ctx_.tensor_of(dyn->sv_len_buf);
        ctx_.set_dyn_batch(dyn);
        rag_->set_buffer_addr(ctx_);
        Tensor none;
        Tensor hidden = model_->encode(ctx_, dyn->s_token, dyn->s_position, none, none, none, none, none, true);
        Tensor logits = model_->get_logits(ctx_, hidden, false);
        py::array out = to_numpy(ctx_, logits);
        ctx_.set_dyn_batch(nullptr);
        return out;
    }

    // Timing of the path a maintainer binding the boundary gets: `iters` back-to-back decode steps of the
    // reference's LLaMA::encode + get_logits (same tokens / positions: every step rewrites the same KV slot), HIP events on the
    // context's stream; eager (one C-ABI launch + one pooled ctx.tensor per reference op), and -- graph = true -- ONE step
    // captured into a hipGraph and replayed (possible because the pool makes ctx.tensor allocation-free in steady state and the
    // wrappers launch on the context's stream only).  Returns {"eager_ms", "graph_ms" or "graph_error"}.
    py::dict time_decode_steps(const py::array& tokens, const py::array& positions, const py::array& mask, int warmup, int iters, bool graph) {
        const size_t B = (size_t)tokens.shape(0);
        auto dyn = std::make_shared<model::DynBatchContext>();
        dyn->s_token = to_device(ctx_, tokens, "s_token");
        dyn->s_position = to_device(ctx_, positions, "s_position");
        dyn->s_placement = to_device(ctx_, positions, "s_placement").view({B, 1});
        dyn->s_mask = to_device(ctx_, mask, "s_mask");
        for (size_t b = 0; b < B; ++b) dyn->sv_len_buf.push_back((int)rag_->get_buf_len(b));
        dyn->s_len_buf = ctx_.tensor_of(dyn->sv_len_buf);
        ctx_.set_dyn_batch(dyn);
        rag_->set_buffer_addr(ctx_);
        hipStream_t st = ctx_.current_cuda_stream();
        auto one = [&]() {
            Tensor none;
            Tensor hidden = model_->encode(ctx_, dyn->s_token, dyn->s_position, none, none, none, none, none, true);
            Tensor logits = model_->get_logits(ctx_, hidden, false);
        };
        py::dict out;
        hipEvent_t e0, e1;
        BM_CUDART_ASSERT(hipEventCreate(&e0));
        BM_CUDART_ASSERT(hipEventCreate(&e1));
        for (int i = 0; i < warmup; ++i) one();
        BM_CUDART_ASSERT(hipStreamSynchronize(st));
        BM_CUDART_ASSERT(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) one();
        BM_CUDART_ASSERT(hipEventRecord(e1, st));
        BM_CUDART_ASSERT(hipEventSynchronize(e1));
        float ms = 0.f;
        BM_CUDART_ASSERT(hipEventElapsedTime(&ms, e0, e1));
        out["eager_ms"] = ms / iters;
        if (graph) {
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            std::string err;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) err = "hipStreamBeginCapture failed";
            else {
                try {
                    one();
                } catch (const std::exception& e) {
                    err = std::string("encode under capture: ") + e.what();
                }
                const hipError_t ec = hipStreamEndCapture(st, &g);
                if (err.empty() && ec != hipSuccess) err = std::string("hipStreamEndCapture: ") + hipGetErrorString(ec);
            }
            if (err.empty() && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) err = "hipGraphInstantiate failed";
            if (err.empty()) {
                for (int i = 0; i < warmup; ++i) (void)hipGraphLaunch(ge, st);
                BM_CUDART_ASSERT(hipStreamSynchronize(st));
                BM_CUDART_ASSERT(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) (void)hipGraphLaunch(ge, st);
                BM_CUDART_ASSERT(hipEventRecord(e1, st));
                BM_CUDART_ASSERT(hipEventSynchronize(e1));
                BM_CUDART_ASSERT(hipEventElapsedTime(&ms, e0, e1));
                out["graph_ms"] = ms / iters;
            } else {
                (void)hipGetLastError();
                out["graph_error"] = err;
            }
            if (ge) (void)hipGraphExecDestroy(ge);
            if (g) (void)hipGraphDestroy(g);
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        ctx_.set_dyn_batch(nullptr);
        return out;
    }

    // the prompt of task b through the whole model (the encode part of a dynamic batch: DynBatchContext's e_* fields for one task at
    // positions pos0 .. pos0 + n - 1) -> the logits of its last token (1, vocab)
    py::array prefill(int b, int len_buf, const py::array& tokens, int pos0) {
        const size_t n = (size_t)tokens.shape(0);
        rag_->resize_task_buf(ctx_, b, (size_t)len_buf);
        auto dyn = std::make_shared<model::DynBatchContext>();
        std::vector<int> pos(n);
        for (size_t i = 0; i < n; ++i) pos[i] = pos0 + (int)i;
        dyn->e_token = to_device(ctx_, tokens, "e_token");
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
        ctx_.set_dyn_batch(dyn);
        Tensor none;
        Tensor hidden = model_->encode(ctx_, dyn->e_token, dyn->e_position, none, none, none, none, none, true);      // (n, dim_model), final norm applied
        Tensor logits = model_->get_logits(ctx_, hidden.slice_dim0(n - 1, n), false);
        py::array out = to_numpy(ctx_, logits);
        ctx_.set_dyn_batch(nullptr);
        return out;
    }
    py::array get_k(int b, int layer) { return to_numpy(ctx_, rag_->buf_k(b, layer)); }
    py::array get_v(int b, int layer) { return to_numpy(ctx_, rag_->buf_v(b, layer)); }

private:
    model::ModelConfig cfg_;
    DummyModel md_;
    model::ModelContext ctx_;
    std::unique_ptr<model::LLaMA> model_;
    std::shared_ptr<model::RagBufferContext> rag_;
};

}  // namespace

void bind_ref_model(py::module_& m) {
    py::class_<RefLLaMA>(m, "RefLLaMA")
        .def(py::init<int, int, int, int, int, int, int, float, float, int, int, int>(), py::arg("num_layers"), py::arg("dim_model"), py::arg("num_heads"),
             py::arg("num_kv_heads"), py::arg("dim_head"), py::arg("dim_ff"), py::arg("vocab_size"), py::arg("eps") = 1e-5f, py::arg("rope_theta") = 10000.0f,
             py::arg("quant_type") = 5, py::arg("group_size") = 128, py::arg("device") = 0)
        .def("load", &RefLLaMA::load)
        .def("set_history", &RefLLaMA::set_history)
        .def("prefill", &RefLLaMA::prefill, py::arg("b"), py::arg("len_buf"), py::arg("tokens"), py::arg("pos0") = 0)
        .def("get_k", &RefLLaMA::get_k)
        .def("get_v", &RefLLaMA::get_v)
        .def("decode_step", &RefLLaMA::decode_step)
        .def("time_decode_steps", &RefLLaMA::time_decode_steps, py::arg("tokens"), py::arg("positions"), py::arg("mask"), py::arg("warmup") = 3,
             py::arg("iters") = 20, py::arg("graph") = true);
}

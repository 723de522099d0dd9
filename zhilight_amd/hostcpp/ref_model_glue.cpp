// ref_model_glue.cpp -- the pybind11 harness classes around the REFERENCE's model::LLaMA (src/model/llama.cpp, compiled unmodified
// into libzhilight_amd_host.so): the path BASELINE.json's north star names from its TOP -- LLaMA::encode (llama.cpp:75-151) ->
// EncoderLayer::forward (block.cpp) -> Attention (attention.cpp) / FeedForward (feedforward.cpp) -> Linear (linear.cpp), then
// get_logits (llama.cpp:159-165) -- inside the reference's own ModelContext (model_context.cpp, compiled unmodified):
//   RefLLaMA        one rank: load a whole model under the reference's parameter names, fill KV histories, run decode steps, time them;
//   RefEngineLLaMA  tensor-parallel: a core::Engine (hostcpp/bm_engine.cpp) with one thread per rank, ModelContext::create on every
//                   rank's thread, LLaMA(parallel = true) sharding its weights by ctx.rank(), decode steps with every reduce going
//                   ModelContext::reduce_sum -> c10d::NCCLAllReduce -> the engine's transports.
// tests/test_gpu_refcompile.py compares with the SAME CPU oracle the repository's own LLaMA is held to.
// Test infrastructure: nothing in the product links this file.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime.h>

#include <cstring>

#include "bm_c10d.h"
#include "bm_engine.h"
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


// The reference's model::LLaMA, tensor-parallel: a core::Engine with one thread per rank (hostcpp/bm_engine.cpp), the reference's
// ModelContext::create (model_context.cpp:89-122) on every rank's thread, LLaMA(parallel = true) -- column / row sharded linears,
// KV heads dealt to the ranks, vocab-parallel embedding and lm_head -- and every reduce through ModelContext::reduce_sum ->
// reduce_sum2 -> c10d::NCCLAllReduce -> the engine's transports.  `devices` may name one device several times
// (the one-GPU test box: the ranks then exchange over the one-shot transport only).  Python objects are touched on the calling
// thread only: the rank threads see host tensors aliasing the numpy arrays and plain vectors.
class RefEngineLLaMA {
public:
    RefEngineLLaMA(int num_layers, int dim_model, int num_heads, int num_kv_heads, int dim_head, int dim_ff, int vocab_size, float eps, float rope_theta,
                   int quant_type, int group_size, const std::vector<int>& devices)
        : cfg_(RefLLaMA::make_cfg(num_layers, dim_model, num_heads, num_kv_heads, dim_head, dim_ff, vocab_size, eps, rope_theta)), md_(cfg_) {
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
            R.model.reset(new model::LLaMA(*R.ctx, cfg_, qc, true));
        });
    }
    ~RefEngineLLaMA() {
        try {
            engine_->device_foreach([&](int r) {           // models and contexts die on the threads that made them, before the engine
                (void)hipDeviceSynchronize();
                ranks_[r].model.reset();
                ranks_[r].ctx.reset();
            });
        } catch (...) {
        }
    }
    int world_size() const { return engine_->world_size(); }
    bool has_rccl() const { return engine_->has_rccl(); }
    std::vector<int> exchange_errors() {
        std::vector<int> e(ranks_.size());
        engine_->device_foreach([&](int r) { e[r] = engine_->exchange_errors(r); });
        return e;
    }
    void load(const std::map<std::string, py::array>& arrays, const std::string& prefix) {
        std::map<std::string, const Tensor> sd;
        for (auto& kv : arrays) sd.emplace(kv.first, host_tensor(kv.second, kv.first));
        engine_->device_foreach([&](int r) { ranks_[r].model->load_state_dict(*ranks_[r].ctx, sd, prefix, false); });
    }
    // task b, every layer: k / v (num_layers, n, hkv, d) -- ALL kv heads; every rank keeps its hkv / world heads in rows 0 .. n - 1
    void set_history(int b, int len_buf, const py::array& k, const py::array& v) {
        Tensor hk = host_tensor(k, "k"), hv = host_tensor(v, "v");
        BM_ASSERT(hk.ndim() == 4 && hk.shape() == hv.shape() && hk.dtype() == DataType::kHalf, "history: (layers, n, hkv, d) fp16");
        const size_t layers = hk.size(0), n = hk.size(1), hkv = hk.size(2), d = hk.size(3), world = ranks_.size();
        BM_ASSERT(hkv % world == 0, "kv heads must divide by the world size");
        const size_t local = hkv / world;
        engine_->device_foreach([&](int r) {
            Rank& R = ranks_[r];
            auto rag = R.ctx->rag_buffer();
            rag->resize_task_buf(*R.ctx, b, (size_t)len_buf);
            for (size_t l = 0; l < layers && n > 0; ++l) {
                const char* sk = hk.data<char>() + ((l * n * hkv) + (size_t)r * local) * d * 2;
                const char* sv = hv.data<char>() + ((l * n * hkv) + (size_t)r * local) * d * 2;
                BM_CUDART_ASSERT(hipMemcpy2D(rag->buf_k(b)[(int)l].data(), local * d * 2, sk, hkv * d * 2, local * d * 2, n, hipMemcpyHostToDevice));
                BM_CUDART_ASSERT(hipMemcpy2D(rag->buf_v(b)[(int)l].data(), local * d * 2, sv, hkv * d * 2, local * d * 2, n, hipMemcpyHostToDevice));
            }
        });
    }
    // one decode step of the whole model on every rank: tokens (B) int32 at positions (B) -> the ranks' logits (world, B, vocab)
    py::array decode_step(const py::array& tokens, const py::array& positions, const py::array& mask) {
        const size_t B = (size_t)tokens.shape(0), world = ranks_.size(), vocab = (size_t)cfg_.vocab_size;
        Tensor ht = host_tensor(tokens, "s_token"), hp = host_tensor(positions, "s_position"), hm = host_tensor(mask, "s_mask");
        std::vector<std::vector<uint16_t>> out(world, std::vector<uint16_t>(B * vocab));
        engine_->device_foreach([&](int r) {
            Rank& R = ranks_[r];
            model::ModelContext& ctx = *R.ctx;
            auto rag = ctx.rag_buffer();
            auto dyn = std::make_shared<model::DynBatchContext>();
            dyn->s_token = upload(ctx, ht);
            dyn->s_position = upload(ctx, hp);
            dyn->s_placement = upload(ctx, hp).view({B, 1});
            dyn->s_mask = upload(ctx, hm);
            for (size_t b = 0; b < B; ++b) dyn->sv_len_buf.push_back((int)rag->get_buf_len(b));
            dyn->s_len_buf = ctx.tensor_of(dyn->sv_len_buf);
            ctx.set_dyn_batch(dyn);
            rag->set_buffer_addr(ctx);
            Tensor none;
            Tensor hidden = R.model->encode(ctx, dyn->s_token, dyn->s_position, none, none, none, none, none, true);
            Tensor logits = R.model->get_logits(ctx, hidden, false);
            BM_ASSERT_EQ(logits.numel(), B * vocab, "logits (B, vocab)");
            logits.to_buffer(out[r].data(), ctx.current_cuda_stream());
            ctx.set_dyn_batch(nullptr);
        });
        return stack(out, B, vocab);
    }
    // the prompt of task b through the whole model on every rank -> the ranks' logits of its last token (world, 1, vocab)
    py::array prefill(int b, int len_buf, const py::array& tokens, int pos0) {
        const size_t n = (size_t)tokens.shape(0), world = ranks_.size(), vocab = (size_t)cfg_.vocab_size;
        Tensor ht = host_tensor(tokens, "e_token");
        std::vector<int> pos(n);
        for (size_t i = 0; i < n; ++i) pos[i] = pos0 + (int)i;
        std::vector<int8_t> mask(n * (size_t)len_buf);
        for (size_t i = 0; i < n; ++i)
            for (int j = 0; j < len_buf; ++j) mask[i * len_buf + j] = j <= pos0 + (int)i;
        std::vector<std::vector<uint16_t>> out(world, std::vector<uint16_t>(vocab));
        engine_->device_foreach([&](int r) {
            Rank& R = ranks_[r];
            model::ModelContext& ctx = *R.ctx;
            ctx.rag_buffer()->resize_task_buf(ctx, b, (size_t)len_buf);
            auto dyn = std::make_shared<model::DynBatchContext>();
            dyn->e_token = upload(ctx, ht);
            dyn->e_placement = ctx.tensor_of(pos);
            dyn->e_position = ctx.tensor_of(pos);
            dyn->e_mask = ctx.tensor_of(mask);
            dyn->ev_batch = {b};
            dyn->ev_input_len = {(int)n};
            dyn->full_input_len = {pos0 + (int)n};
            dyn->ev_len_buf = {len_buf};
            ctx.set_dyn_batch(dyn);
            Tensor none;
            Tensor hidden = R.model->encode(ctx, dyn->e_token, dyn->e_position, none, none, none, none, none, true);
            Tensor logits = R.model->get_logits(ctx, hidden.slice_dim0(n - 1, n), false);
            BM_ASSERT_EQ(logits.numel(), vocab, "logits (1, vocab)");
            logits.to_buffer(out[r].data(), ctx.current_cuda_stream());
            ctx.set_dyn_batch(nullptr);
        });
        return stack(out, 1, vocab);
    }
    // rank r's K rows of task b in `layer`: (len_buf, hkv / world, d)
    py::array get_k(int r, int b, int layer) {
        std::vector<uint16_t> host;
        std::vector<size_t> shape;
        engine_->run(r, [&] {
            const Tensor& t = ranks_[r].ctx->rag_buffer()->buf_k(b, layer);
            shape = t.shape();
            host.resize(t.numel());
            t.to_buffer(host.data(), ranks_[r].ctx->current_cuda_stream());
        });
        py::array out(py::dtype("float16"), std::vector<py::ssize_t>(shape.begin(), shape.end()));
        std::memcpy(out.mutable_data(), host.data(), host.size() * 2);
        return out;
    }

private:
    struct Rank {
        std::unique_ptr<model::ModelContext> ctx;
        std::unique_ptr<model::LLaMA> model;
    };
    static Tensor upload(const Context& ctx, const Tensor& host) {
        Tensor d = ctx.tensor(host.shape(), host.dtype());
        d.from_buffer(host.data(), false, ctx.current_cuda_stream());
        return d;
    }
    static py::array stack(const std::vector<std::vector<uint16_t>>& out, size_t rows, size_t vocab) {
        py::array res(py::dtype("float16"), std::vector<py::ssize_t>{(py::ssize_t)out.size(), (py::ssize_t)rows, (py::ssize_t)vocab});
        for (size_t r = 0; r < out.size(); ++r) std::memcpy((char*)res.mutable_data() + r * rows * vocab * 2, out[r].data(), rows * vocab * 2);
        return res;
    }
    model::ModelConfig cfg_;
    DummyModel md_;
    std::unique_ptr<bmengine::core::Engine> engine_;
    std::vector<Rank> ranks_;
};


// The engine's collectives on their own: one Context per rank thread, every rank calling the c10d entry point the reference's layer
// code calls (bm_c10d.h) on ITS array; returns what every rank holds afterwards.  Arrays of any element size: ranks that share a
// device move non-16-bit-float payloads byte by byte (bm_engine.cpp), which must be bit-exact.
class RefEngine {
public:
    explicit RefEngine(const std::vector<int>& devices) {
        std::vector<bmengine::core::DeviceConfiguration> dc;
        for (int d : devices) dc.emplace_back(d, (size_t)0);
        engine_.reset(new bmengine::core::Engine(dc));
        ctx_.resize(devices.size());
        engine_->device_foreach([&](int r) { ctx_[r].reset(new Context(engine_->create_context_rank(r))); });
    }
    ~RefEngine() {
        try {
            engine_->device_foreach([&](int r) {
                (void)hipDeviceSynchronize();
                ctx_[r].reset();
            });
        } catch (...) {
        }
    }
    std::vector<int> exchange_errors() {
        std::vector<int> e(ctx_.size());
        engine_->device_foreach([&](int r) { e[r] = engine_->exchange_errors(r); });
        return e;
    }
    // op: "broadcast" (root), "all_gather", "all_reduce" (fp16 sums), "reduce_scatter"; per_rank: one array per rank, equal shapes
    std::vector<py::array> run(const std::string& op, const std::vector<py::array>& per_rank, int root) {
        const size_t world = ctx_.size();
        BM_ASSERT(per_rank.size() == world, "one array per rank");
        std::vector<Tensor> host;
        for (auto& a : per_rank) host.push_back(host_tensor(a, "x"));
        const size_t nbytes = host[0].nbytes();
        const size_t out_bytes = op == "all_gather" ? nbytes * world : op == "reduce_scatter" ? nbytes / world : nbytes;
        std::vector<std::vector<char>> out(world, std::vector<char>(out_bytes));
        engine_->device_foreach([&](int r) {
            const Context& ctx = *ctx_[r];
            Tensor x = ctx.tensor(host[r].shape(), host[r].dtype());
            x.from_buffer(host[r].data(), false, ctx.current_cuda_stream());
            Tensor y;
            if (op == "broadcast") {
                bmengine::c10d::NCCLBroadcast(ctx, x, x, root);
                y = x;
            } else if (op == "all_gather") {
                y = ctx.all_gather(x);
            } else if (op == "reduce_scatter") {
                y = ctx.reduce_scatter(x);
            } else if (op == "all_reduce") {
                y = ctx.tensor(x.shape(), x.dtype());
                bmengine::c10d::NCCLAllReduce(ctx, x, y, ncclSum);
            } else {
                BM_EXCEPTION("unknown op " + op);
            }
            BM_ASSERT_EQ(y.nbytes(), out_bytes, "result size");
            y.to_buffer(out[r].data(), ctx.current_cuda_stream());
        });
        std::vector<py::array> res;
        for (size_t r = 0; r < world; ++r) {
            py::array a(per_rank[0].dtype(), std::vector<py::ssize_t>{(py::ssize_t)(out_bytes / per_rank[0].dtype().itemsize())});
            std::memcpy(a.mutable_data(), out[r].data(), out_bytes);
            res.push_back(a);
        }
        return res;
    }

private:
    std::unique_ptr<bmengine::core::Engine> engine_;
    std::vector<std::unique_ptr<Context>> ctx_;
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
    py::class_<RefEngine>(m, "RefEngine")
        .def(py::init<const std::vector<int>&>(), py::arg("devices") = std::vector<int>{0, 0})
        .def("exchange_errors", &RefEngine::exchange_errors)
        .def("run", &RefEngine::run, py::arg("op"), py::arg("per_rank"), py::arg("root") = 0);
    py::class_<RefEngineLLaMA>(m, "RefEngineLLaMA")
        .def(py::init<int, int, int, int, int, int, int, float, float, int, int, const std::vector<int>&>(), py::arg("num_layers"), py::arg("dim_model"),
             py::arg("num_heads"), py::arg("num_kv_heads"), py::arg("dim_head"), py::arg("dim_ff"), py::arg("vocab_size"), py::arg("eps") = 1e-5f,
             py::arg("rope_theta") = 10000.0f, py::arg("quant_type") = 5, py::arg("group_size") = 128, py::arg("devices") = std::vector<int>{0, 0})
        .def("world_size", &RefEngineLLaMA::world_size)
        .def("has_rccl", &RefEngineLLaMA::has_rccl)
        .def("exchange_errors", &RefEngineLLaMA::exchange_errors)
        .def("load", &RefEngineLLaMA::load)
        .def("set_history", &RefEngineLLaMA::set_history)
        .def("prefill", &RefEngineLLaMA::prefill, py::arg("b"), py::arg("len_buf"), py::arg("tokens"), py::arg("pos0") = 0)
        .def("get_k", &RefEngineLLaMA::get_k)
        .def("decode_step", &RefEngineLLaMA::decode_step);
}

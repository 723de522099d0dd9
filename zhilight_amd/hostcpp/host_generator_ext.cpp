// host_generator_ext.cpp -- what src/generator/batch_generator.cpp (the reference's dynamic-batch scheduler, compiled UNMODIFIED into the
// zhilight.C binding: zhilight_amd/build.py build_binding) calls next to the model: the logit post-processing the reference keeps in
// src/generator/beam_util.cu / random_util.cu and bmengine's functions/{softmax,topk}.cu, the slice copies of the prefix cache
// (src/kvcache/transformer_buffer.cu:420-540) and the handful of cuRAND entry points its samplers are written against.
//   on the device (sampling_ops.hip over the C ABI): log_softmax_bias, softmax, TopK, gather_logits, scatter_update, the repetition /
//       presence penalties;
//   on the host, restated from the behaviour: calc_repetition_ngram (the token that followed an earlier occurrence of the current n-token suffix
//       is penalised by penalty ^ (n + 1)),
//       the per-hypothesis loops around the penalty kernel, random_sampler_gpu (sort descending, inclusive sum, u ~ U(0, top_p'),
//       first index whose cumulative mass reaches u -- sampling is off the hot path and runs on the host here);
//   curand*: a counter-based generator (seed, offset) -> uniforms in (0, 1]: the samplers need A reproducible stream, not cuRAND's.
// Sampling results therefore differ from the reference's draw by draw (another generator); greedy / beam search do not sample.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <unordered_map>

#include "host_common.h"

#include "bmengine/functions/all.h"
#include "generator/beam_buffer_manager.hpp"
#include "generator/beam_util.h"
#include "generator/random_util.h"
#include "kvcache/transformer_buffer.h"
#include "model/model_context.h"

// ---- the generator behind curandGenerator_t -----------------------------------------------------------------------------------
struct zl_rand_generator_st {
    unsigned long long seed = 0, offset = 0;
    hipStream_t stream = nullptr;
};
namespace {
inline unsigned long long mix64(unsigned long long z) {       // splitmix64's finaliser
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
inline float uniform01(unsigned long long seed, unsigned long long counter) {   // (0, 1], 24 bits
    const unsigned long long h = mix64(mix64(seed + 0x9e3779b97f4a7c15ULL) ^ (counter * 0xd1342543de82ef95ULL + 1));
    return (float)((h >> 40) + 1) * (1.0f / 16777216.0f);
}
int tcode(DataType t) {
    switch (t) {
    case DataType::kHalf: return ZL_T_F16;
    case DataType::kBFloat16: return ZL_T_BF16;
    case DataType::kFloat: return ZL_T_F32;
    default: throw BMEngineException(std::string("logits of type ") + bmengine::core::get_data_type_name(t) + ": half / bfloat16 / float expected", __FILE__, __LINE__, __func__);
    }
}
zl_stream_t st_of(const Context& ctx) { return (zl_stream_t)ctx.current_cuda_stream(); }
}  // namespace

extern "C" {
curandStatus_t curandCreateGenerator(curandGenerator_t* generator, curandRngType_t) {
    *generator = new zl_rand_generator_st();
    return CURAND_STATUS_SUCCESS;
}
curandStatus_t curandDestroyGenerator(curandGenerator_t generator) {
    delete generator;
    return CURAND_STATUS_SUCCESS;
}
curandStatus_t curandSetStream(curandGenerator_t generator, hipStream_t stream) {
    generator->stream = stream;
    return CURAND_STATUS_SUCCESS;
}
curandStatus_t curandSetPseudoRandomGeneratorSeed(curandGenerator_t generator, unsigned long long seed) {
    generator->seed = seed;
    return CURAND_STATUS_SUCCESS;
}
curandStatus_t curandSetGeneratorOrdering(curandGenerator_t, curandOrdering_t) { return CURAND_STATUS_SUCCESS; }
curandStatus_t curandSetGeneratorOffset(curandGenerator_t generator, unsigned long long offset) {
    generator->offset = offset;
    return CURAND_STATUS_SUCCESS;
}
// n uniforms into DEVICE memory, ordered on the generator's stream (synchronous: the host block is a local)
curandStatus_t curandGenerateUniform(curandGenerator_t generator, float* out, size_t n) {
    std::vector<float> host(n);
    for (size_t i = 0; i < n; ++i) host[i] = uniform01(generator->seed, generator->offset + i);
    generator->offset += n;
    if (hipMemcpyAsync(out, host.data(), n * sizeof(float), hipMemcpyHostToDevice, generator->stream) != hipSuccess) return (curandStatus_t)1;
    if (hipStreamSynchronize(generator->stream) != hipSuccess) return (curandStatus_t)1;
    return CURAND_STATUS_SUCCESS;
}
}

// ---- bmengine::functions: softmax, TopK -----------------------------------------------------------------------------------------
namespace bmengine {
namespace functions {

void softmax(const core::Context& ctx, const core::Tensor& logits, const core::Tensor& output, float temperature) {
    BM_ASSERT(logits.ndim() >= 1 && logits.shape() == output.shape() && logits.dtype() == output.dtype(), "softmax: logits and output of one shape and type");
    const size_t n = logits.size(-1);
    ZL_CK(zl_softmax_rows(logits.data(), output.data(), logits.numel() / n, n, temperature, tcode(logits.dtype()), st_of(ctx)), "softmax");
}
void bitonic_topk(const core::Context& ctx, const core::Tensor& x, const core::Tensor& out, const core::Tensor& pos) {
    BM_ASSERT(x.ndim() == 2 && out.ndim() == 2 && pos.ndim() == 2 && out.size(0) == x.size(0) && pos.shape() == out.shape(), "topk: (batch, n) -> (batch, top) x 2");
    BM_ASSERT(out.dtype() == x.dtype() && pos.dtype() == DataType::kInt32, "topk: values in the input type, int32 positions");
    ZL_CK(zl_topk_rows(x.data(), out.data(), pos.data<int32_t>(), x.size(0), x.size(1), (int)out.size(1), tcode(x.dtype()), st_of(ctx)), "topk");
}
class TopK::impl {};
TopK::TopK(const core::Context&) : pimpl(new impl) {}
TopK::~TopK() = default;
std::pair<core::Tensor, core::Tensor> TopK::forward(const core::Context& ctx, const core::Tensor& inp, int top) {
    BM_ASSERT(inp.ndim() == 2, "inp must be 2d");
    BM_ASSERT(top > 0, "top must be > 0");
    auto ret = std::make_pair(ctx.tensor({inp.size(0), (size_t)top}, inp.dtype()), ctx.tensor({inp.size(0), (size_t)top}, DataType::kInt32));
    bitonic_topk(ctx, inp, ret.first, ret.second);
    return ret;
}

}  // namespace functions
}  // namespace bmengine

// ---- beam_utility -----------------------------------------------------------------------------------------------------------------
namespace beam_utility {

void log_softmax_bias(const core::Context& ctx, const core::Tensor& logits, const core::Tensor& bias, float temperature, core::Tensor* out) {
    BM_ASSERT(logits.ndim() >= 2, "logits must be 2 or 3 dimensional");
    BM_ASSERT(out && logits.shape() == out->shape(), "logits and out has different shape");
    const size_t n = logits.size(-1), rows = logits.numel() / n;
    BM_ASSERT(bias.dtype() == DataType::kFloat && bias.numel() >= rows, "bias: one float per row");
    ZL_CK(zl_log_softmax_bias(logits.data(), bias.data<float>(), out->data(), rows, n, temperature, tcode(logits.dtype()), st_of(ctx)), "log_softmax_bias");
}
core::Tensor log_softmax_bias(const core::Context& ctx, const core::Tensor& logits, const core::Tensor& bias) {
    core::Tensor out = ctx.tensor(logits.shape(), logits.dtype());
    log_softmax_bias(ctx, logits, bias, 0.f, &out);         // (temperature 0 = the form without the division, beam_util.cu:44-66)
    return out;
}
core::Tensor log_softmax_bias(const core::Context& ctx, const core::Tensor& logits, const core::Tensor& bias, float temperature) {
    core::Tensor out = ctx.tensor(logits.shape(), logits.dtype());
    log_softmax_bias(ctx, logits, bias, temperature, &out);
    return out;
}
core::Tensor gather_logits(const core::Context& ctx, const core::Tensor& indexes, const core::Tensor& logits) {
    BM_ASSERT(indexes.dtype() == DataType::kInt32, "indexes are int32");
    core::Tensor out = ctx.tensor(indexes.shape(), DataType::kFloat);
    ZL_CK(zl_gather_logits(indexes.data<int32_t>(), logits.data(), out.data<float>(), indexes.numel(), tcode(logits.dtype()), st_of(ctx)), "gather_logits");
    return out;
}
core::Tensor apply_gumbel_softmax(const core::Context& ctx, curandGenerator_t& gen, const core::Tensor& logits) {
    // out = T(x - log(-log(u))), u ~ U(0, 1] (beam_util.cu:159-190): the noise is drawn on the host (see the header comment)
    const size_t n = logits.numel();
    core::Tensor eps = ctx.tensor({n}, DataType::kFloat), f32 = bmengine::functions::typecast(ctx, logits, DataType::kFloat);
    CURAND_CHECK(curandGenerateUniform(gen, eps.data<float>(), n));
    std::vector<float> u = eps.to_vector<float>(ctx.current_cuda_stream()), x = f32.to_vector<float>(ctx.current_cuda_stream());
    for (size_t i = 0; i < n; ++i) x[i] = x[i] - logf(-logf(u[i]));
    core::Tensor noisy = ctx.tensor(logits.shape(), DataType::kFloat);
    noisy.from_buffer(x.data(), false, ctx.current_cuda_stream());
    return bmengine::functions::typecast(ctx, noisy, logits.dtype());
}
void beam_repetition_penalty(const core::Context& ctx, const std::vector<float>& penalty_factor, const std::vector<int32_t>& tokens,
                             const std::vector<int32_t>& batch_id, core::Tensor& logits, const std::vector<float>& presence_penalty) {
    BM_ASSERT(tokens.size() == batch_id.size() && penalty_factor.size() == tokens.size(), "tokens, batch_id and the factors must have the same size");
    BM_ASSERT(tokens.size() > 0, "tokens and batch_id must have at least one element");
    BM_ASSERT(presence_penalty.empty() || presence_penalty.size() == tokens.size(), "one presence penalty per token");
    core::Tensor f = ctx.tensor_of(penalty_factor), t = ctx.tensor_of(tokens), b = ctx.tensor_of(batch_id);
    core::Tensor p = presence_penalty.empty() ? core::Tensor() : ctx.tensor_of(presence_penalty);
    ZL_CK(zl_repetition_penalty(f.data<float>(), p.numel() ? p.data<float>() : nullptr, t.data<int32_t>(), b.data<int32_t>(), logits.data(), tokens.size(),
                                logits.size(-1), tcode(logits.dtype()), st_of(ctx)), "beam_repetition_penalty");
    BM_CUDART_ASSERT(hipStreamSynchronize(ctx.current_cuda_stream()));      // (the index tensors go back to the pool)
}
void random_repetition_penalty(const core::Context& ctx, const std::vector<float>& penalty_factor, const std::vector<int32_t>& tokens,
                               const std::vector<int32_t>& batch_id, core::Tensor& logits) {
    beam_repetition_penalty(ctx, penalty_factor, tokens, batch_id, logits, {});
}
void scatter_update(const core::Context& ctx, const std::vector<float>& values, const std::vector<int32_t>& token_ids, const std::vector<int32_t>& batch_ids,
                    core::Tensor& logits, bool add) {
    BM_ASSERT(token_ids.size() == batch_ids.size() && values.size() == batch_ids.size(), "values, tokens and batch_id must have the same size");
    BM_ASSERT(batch_ids.size() > 0, "tokens and batch_id must have at least one element");
    core::Tensor v = ctx.tensor_of(values), t = ctx.tensor_of(token_ids), b = ctx.tensor_of(batch_ids);
    ZL_CK(zl_scatter_logits(v.data<float>(), t.data<int32_t>(), b.data<int32_t>(), logits.data(), batch_ids.size(), logits.size(-1), add ? 1 : 0,
                            tcode(logits.dtype()), st_of(ctx)), "scatter_update");
    BM_CUDART_ASSERT(hipStreamSynchronize(ctx.current_cuda_stream()));
}
// The callers hand the hypothesis NEWEST TOKEN FIRST.  border[i] (prefix function) = the longest run of most-recent tokens that re-occurs
// ending at position i; the element just in front of that re-occurrence (index i - len) is the token that FOLLOWED the earlier occurrence of
// the current len-token suffix -- generating it again would extend the repeat to len + 1 tokens, so it is penalised by
// ngram_penalty ^ (len + 1) (every token at least by ngram_penalty ^ 1), and a token keeps its largest penalty.
std::unordered_map<int, float> calc_repetition_ngram(const std::vector<int>& token_ids, float ngram_penalty) {
    std::unordered_map<int, float> ret;
    const int n = (int)token_ids.size();
    if (n == 0) return ret;
    std::vector<int> border(n, -1);                          // border[i]: last index of the longest proper prefix that is a suffix of [0 .. i]
    for (int i = 1; i < n; ++i) {
        int p = border[i - 1];
        while (p >= 0 && token_ids[p + 1] != token_ids[i]) p = border[p];
        border[i] = token_ids[p + 1] == token_ids[i] ? p + 1 : -1;
    }
    std::vector<int> longest(n, 0);
    for (int i = 0; i < n; ++i) {
        const int len = border[i] + 1;
        longest[i - len] = std::max(longest[i - len], len);
    }
    for (int i = 0; i < n; ++i) {
        const float v = powf(ngram_penalty, (float)(longest[i] + 1));
        auto it = ret.find(token_ids[i]);
        if (it == ret.end()) ret.emplace(token_ids[i], std::max(0.f, v));
        else it->second = std::max(it->second, v);
    }
    return ret;
}
void apply_beam_repetition_penalty(model::ModelContext& ctx, const BeamBufferManager<int>& bm, const std::vector<int>& hypotheses_last_pos, float ngram_penalty,
                                   float repetition_penalty, core::Tensor* logits_all) {
    std::vector<float> factors;
    std::vector<int32_t> tokens, rows, reversed;
    for (size_t h = 0; h < hypotheses_last_pos.size(); ++h) {
        reversed.clear();
        bm.get_hypothesis_tokens(hypotheses_last_pos[h], &reversed, true);
        for (const auto& kv : calc_repetition_ngram(reversed, ngram_penalty)) {
            tokens.push_back(kv.first);
            rows.push_back((int32_t)h);
            factors.push_back(kv.second * repetition_penalty);
        }
    }
    if (!tokens.empty()) beam_repetition_penalty(ctx, factors, tokens, rows, *logits_all);
}
void batch_apply_repetition_penalty(model::ModelContext& ctx, const std::vector<std::vector<std::vector<int>>>& output_sequences, float ngram_penalty,
                                    float repetition_penalty, core::Tensor& logits_all) {
    std::vector<core::Tensor> per_task = logits_all.chunk();
    for (size_t b = 0; b < output_sequences.size(); ++b) {
        std::vector<float> factors;
        std::vector<int32_t> tokens, rows;
        for (size_t h = 0; h < output_sequences[b].size(); ++h)
            for (const auto& kv : calc_repetition_ngram(output_sequences[b][h], ngram_penalty)) {
                tokens.push_back(kv.first);
                rows.push_back((int32_t)h);
                factors.push_back(kv.second * repetition_penalty);
            }
        if (!tokens.empty()) beam_repetition_penalty(ctx, factors, tokens, rows, per_task[b]);
    }
}
void init_curand_gen(const core::Context& ctx, curandGenerator_t& gen, int seed) {
    CURAND_CHECK(curandSetStream(gen, ctx.current_cuda_stream()));
    CURAND_CHECK(curandSetGeneratorOffset(gen, 0));
    CURAND_CHECK(curandSetGeneratorOrdering(gen, CURAND_ORDERING_PSEUDO_BEST));
    CURAND_CHECK(curandSetPseudoRandomGeneratorSeed(gen, (unsigned long long)seed));
}
// probs (..., n_classes) -> select (rows * num_samples): per row sort descending (stable), inclusive sum, threshold u * min(top_p, cum[top_k - 1]) * cum[last],
// first position whose cumulative mass reaches it (random_util.cu:85-199).  On the host; like the reference it leaves the cumulative sums in probs.
void random_sampler_gpu(const core::Context& ctx, curandGenerator_t& gen, core::Tensor& probs, core::Tensor& select, float top_p, int top_k, int num_samples) {
    const size_t n = probs.size(-1), rows = probs.numel() / n;
    BM_ASSERT(top_p <= 1.0f && top_p >= 0.0f, "top_p must be in [0, 1]");
    BM_ASSERT(top_k >= 0 && (size_t)top_k < n, "invalid top k");
    BM_ASSERT_EQ(select.size(0), rows * (size_t)num_samples, "invalid select size");
    const size_t per_row = select.size(0) / rows;
    core::Tensor u_d = ctx.tensor({rows * per_row}, DataType::kFloat);
    CURAND_CHECK(curandGenerateUniform(gen, u_d.data<float>(), rows * per_row));
    std::vector<float> u = u_d.to_vector<float>(ctx.current_cuda_stream());
    std::vector<float> p = bmengine::functions::typecast(ctx, probs, DataType::kFloat).to_vector<float>(ctx.current_cuda_stream());
    std::vector<int32_t> picks(rows * per_row), order(n);
    std::vector<float> cum(rows * n);
    for (size_t r = 0; r < rows; ++r) {
        const float* row = p.data() + r * n;
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [row](int a, int b) { return row[a] > row[b]; });
        float acc = 0.f;
        float* c = cum.data() + r * n;
        for (size_t i = 0; i < n; ++i) c[i] = (acc += row[order[i]]);
        const float cap = top_k > 0 ? std::min(c[top_k - 1], top_p) : top_p;
        for (size_t j = 0; j < per_row; ++j) {
            const float v = u[r * per_row + j] * cap * c[n - 1];
            const size_t at = std::lower_bound(c, c + n - 1, v) - c;      // (the last class is the fallback, as in the reference's search)
            picks[r * per_row + j] = order[at];
        }
    }
    core::Tensor cum_d = ctx.tensor(probs.shape(), DataType::kFloat);
    cum_d.from_buffer(cum.data(), false, ctx.current_cuda_stream());
    core::Tensor back = bmengine::functions::typecast(ctx, cum_d, probs.dtype());
    BM_CUDART_ASSERT(hipMemcpyAsync(probs.data(), back.data(), probs.nbytes(), hipMemcpyDeviceToDevice, ctx.current_cuda_stream()));
    BM_ASSERT(select.dtype() == DataType::kInt32, "select is int32");
    select.from_buffer(picks.data(), false, ctx.current_cuda_stream());
}

}  // namespace beam_utility

// ---- the prefix cache's slice copies ----------------------------------------------------------------------------------------------
namespace kvcache {

// rows [start, start + len) of every layer's buffer -> (layers, len, heads, dim) under BSHD, (layers, heads, len, dim) otherwise
core::Tensor TransformerBuffer::dump_slice(core::Context& ctx, size_t start, size_t len, core::Tensor* out) {
    BM_ASSERT(is_dyn_batch(), "Not a dynamic batch buffer");
    BM_ASSERT(buffer[0].numel(), "buffer no data");
    const size_t len_buf = buffer[0].size(BSHD ? 0 : 1), esz = core::get_elem_size(dtype), row = (size_t)dim_head * esz;
    BM_ASSERT_LE(start + len, len_buf, "out of range");
    const std::vector<size_t> shape = BSHD ? std::vector<size_t>{num_layers, len, (size_t)num_heads, (size_t)dim_head}
                                           : std::vector<size_t>{num_layers, (size_t)num_heads, len, (size_t)dim_head};
    core::Tensor ret = out ? out->view(shape) : ctx.tensor(shape, dtype);
    hipStream_t st = ctx.current_cuda_stream();
    const size_t per_layer = len * num_heads * row;
    for (size_t l = 0; l < num_layers; ++l) {
        char* dst = ret.data<char>() + l * per_layer;
        if (BSHD) BM_CUDART_ASSERT(hipMemcpyAsync(dst, buffer[l].data<char>() + start * num_heads * row, per_layer, hipMemcpyDeviceToDevice, st));
        else BM_CUDART_ASSERT(hipMemcpy2DAsync(dst, len * row, buffer[l].data<char>() + start * row, len_buf * row, len * row, num_heads, hipMemcpyDeviceToDevice, st));
    }
    return ret;
}
void TransformerBuffer::load_slice(core::Context& ctx, size_t start, size_t len, const core::Tensor& input) {
    BM_ASSERT(is_dyn_batch(), "Not a dynamic batch buffer");
    BM_ASSERT(buffer[0].numel(), "buffer must be resized first");
    const size_t len_buf = buffer[0].size(BSHD ? 0 : 1), esz = core::get_elem_size(dtype), row = (size_t)dim_head * esz;
    BM_ASSERT_LE(start + len, len_buf, "out of range");
    BM_ASSERT_EQ(input.size(0), num_layers, "Wrong num_layers");
    BM_ASSERT_EQ(input.numel(), num_layers * len * num_heads * dim_head, "Wrong slice size");
    hipStream_t st = ctx.current_cuda_stream();
    const size_t per_layer = len * num_heads * row;
    for (size_t l = 0; l < num_layers; ++l) {
        const char* src = input.data<char>() + l * per_layer;
        if (BSHD) BM_CUDART_ASSERT(hipMemcpyAsync(buffer[l].data<char>() + start * num_heads * row, src, per_layer, hipMemcpyDeviceToDevice, st));
        else BM_CUDART_ASSERT(hipMemcpy2DAsync(buffer[l].data<char>() + start * row, len_buf * row, src, len * row, len * row, num_heads, hipMemcpyDeviceToDevice, st));
    }
}

}  // namespace kvcache

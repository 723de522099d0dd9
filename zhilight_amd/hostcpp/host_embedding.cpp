// host_embedding.cpp -- nn::RawEmbedding (src/nn/embedding/embedding.h:24-47; embedding.cu:230-289 NormalImpl, :291-390
// RowParallelImpl): token lookup = zl_embedding, projection = the lm_head product (zl_gemm_nt_small_m / zl_gemm_nt).
// parallel (tensor-parallel ranks): the vocabulary rows are dealt in equal parts of round_up(vocab, 128) / world, rows past
// the vocabulary zero (embedding.cu:303-321); forward looks the rank's rows up (zero for the others') and sums over the ranks
// (ctx.reduce_sum, :350); projection multiplies by the rank's rows and gathers the parts (c10d::NCCLAllGather, :382) -- here the
// part logits are (rows, part) and the gathered parts are placed side by side with strided copies (the reference computes them
// transposed and transposes back, :366-384: same values).
#include <cmath>

#include "bm_c10d.h"
#include "bm_functions.h"
#include "host_common.h"
#include "nn/embedding/embedding.h"

namespace nn {

class RawEmbedding::impl {
public:
    int dim_model, vocab_size;
    core::DataType dtype;
    float scale = 1.0f, logit_scale = 1.0f;
    core::Tensor weight;
    // ZLD16M copy of `weight` (zl_dense_pack_m: a wavefront's MFMA fragment load is 1 KiB contiguous), made when a projection of 5..32 rows
    // first needs it -- what the Python driver keeps for the lm_head of a decode batch (176-186 us against 210-680 us row-major)
    core::Tensor packed;
    const void* packed_of = nullptr;
    bool parallel = false;
    int begin = 0, end = 0;               // the vocabulary rows this rank holds
    int zdt() const { return dtype == DataType::kHalf ? ZL_F16 : ZL_BF16; }
    size_t part() const { return (size_t)(end - begin); }
};
RawEmbedding::RawEmbedding(const core::Context& ctx, int dim_model, int vocab_size, bool scale_weights, core::DataType dtype, bool parallel)
    : pimpl(new impl) {
    BM_ASSERT(dtype == DataType::kHalf || dtype == DataType::kBFloat16, "RawEmbedding: fp16 / bf16");
    pimpl->dim_model = dim_model;
    pimpl->vocab_size = vocab_size;
    pimpl->dtype = dtype;
    pimpl->parallel = parallel && ctx.world_size() > 1;
    pimpl->end = vocab_size;
    if (scale_weights) pimpl->scale = 1.0f / sqrtf((float)dim_model);
    if (!pimpl->parallel) {
        pimpl->weight = ctx.parameter({(size_t)vocab_size, (size_t)dim_model}, dtype);
        add_parameter("weight", pimpl->weight);
    }
}
RawEmbedding::~RawEmbedding() = default;
void RawEmbedding::set_scale_weights(bool b) { pimpl->scale = b ? 1.0f / sqrtf((float)pimpl->dim_model) : 1.0f; }
void RawEmbedding::set_scale_factor(float b) { pimpl->scale = b; }
void RawEmbedding::set_logit_scale(float b) { pimpl->logit_scale = b; }
void RawEmbedding::load_state_dict(const core::Context& ctx, const std::map<std::string, const core::Tensor>& state_dict, const std::string& prefix,
                                   bool allow_missing) {
    pimpl->packed_of = nullptr;           // a packed copy of the previous contents is stale
    pimpl->packed = core::Tensor();
    if (!pimpl->parallel) {
        core::Layer::load_state_dict(ctx, state_dict, prefix, allow_missing);
        return;
    }
    const size_t vocab = (size_t)pimpl->vocab_size, dim = (size_t)pimpl->dim_model;
    const size_t round_size = (vocab + 127) / 128 * 128, part = round_size / (size_t)ctx.world_size();
    BM_ASSERT(part * (size_t)ctx.world_size() == round_size, "RawEmbedding: round_up(vocab, 128) must divide by the world size");
    pimpl->begin = (int)((size_t)ctx.rank() * part);
    pimpl->end = pimpl->begin + (int)part;
    auto it = state_dict.find(prefix + ".weight");
    BM_ASSERT(it != state_dict.end(), "Weight not found: " + prefix + ".weight");
    const core::Tensor& src = it->second;
    BM_ASSERT(src.ndim() == 2 && src.size(0) == vocab && src.size(1) == dim, "RawEmbedding: (vocab, dim_model) weight");
    BM_ASSERT(src.dtype() == DataType::kHalf || src.dtype() == DataType::kBFloat16 || src.dtype() == DataType::kFloat,
              "RawEmbedding: half, bfloat16 or float weight");
    pimpl->weight = ctx.tensor({part, dim}, pimpl->dtype);
    hipStream_t st = ctx.current_cuda_stream();
    BM_CUDART_ASSERT(hipMemsetAsync(pimpl->weight.data(), 0, pimpl->weight.nbytes(), st));
    const size_t first = (size_t)pimpl->begin, last = std::min((size_t)pimpl->end, vocab);
    if (first < last) {
        const core::Tensor rows = src.slice_dim0(first, last);
        const hipMemcpyKind kind = src.device() >= 0 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        if (src.dtype() == pimpl->dtype) {
            BM_CUDART_ASSERT(hipMemcpyAsync(pimpl->weight.data(), rows.data(), rows.nbytes(), kind, st));
        } else {
            // another element type than the model's (a bf16 checkpoint into an fp16 model, fp32 sources: the reference's assign_or_copy +
            // typecast, embedding.cu load_state_dict / EMBEDDING_AUTO_CAST): stage the rank's rows and convert, never reinterpret
            core::Tensor staged = ctx.tensor({last - first, dim}, src.dtype());
            BM_CUDART_ASSERT(hipMemcpyAsync(staged.data(), rows.data(), rows.nbytes(), kind, st));
            core::Tensor conv = bmengine::functions::typecast(ctx, staged, pimpl->dtype);
            BM_CUDART_ASSERT(hipMemcpyAsync(pimpl->weight.data(), conv.data(), conv.nbytes(), hipMemcpyDeviceToDevice, st));
        }
    }
    BM_CUDART_ASSERT(hipStreamSynchronize(st));          // (the source may be a host array the caller frees)
}
core::Tensor RawEmbedding::forward(const core::Context& ctx, const core::Tensor& ids) {
    BM_ASSERT(ids.dtype() == DataType::kInt32, "token ids are int32");
    BM_ASSERT(ids.ndim() == 1 || ids.ndim() == 2, "ids must be 1d or 2d");
    const size_t n = ids.numel();
    std::vector<size_t> shape = ids.shape();
    shape.push_back((size_t)pimpl->dim_model);
    core::Tensor out = ctx.tensor(shape, pimpl->dtype);
    ZL_CK(zl_embedding(ids.data<int32_t>(), pimpl->weight.data<uint16_t>(), out.data<uint16_t>(), n, pimpl->dim_model, pimpl->begin, pimpl->end, pimpl->scale,
                       pimpl->zdt(), (zl_stream_t)ctx.current_cuda_stream()), "embedding");
    if (pimpl->parallel) return ctx.reduce_sum(out, pimpl->dtype);     // rows of the other ranks' tokens are zero here
    return out;
}
core::Tensor RawEmbedding::projection(const core::Context& ctx, const core::Tensor& input) {
    const int64_t m = input.numel() / input.size(-1), k = input.size(-1);
    BM_ASSERT_EQ(k, (int64_t)pimpl->dim_model, "RawEmbedding::projection: dim mismatch");
    const int64_t n = pimpl->parallel ? (int64_t)pimpl->part() : (int64_t)pimpl->vocab_size;
    const float alpha = pimpl->scale * pimpl->logit_scale;
    zl_stream_t st = (zl_stream_t)ctx.current_cuda_stream();
    std::vector<size_t> shape = input.shape();
    shape.back() = (size_t)pimpl->vocab_size;
    core::Tensor local = pimpl->parallel ? ctx.tensor({(size_t)m, (size_t)n}, pimpl->dtype) : ctx.tensor(shape, pimpl->dtype);
    if (m <= 4)
        ZL_CK(zl_gemm_nt_small_m(input.data<uint16_t>(), k, pimpl->weight.data<uint16_t>(), nullptr, local.data<uint16_t>(), m, n, k, alpha, pimpl->zdt(), nullptr, 0.f, st),
              "lm_head (row-streaming)");
    else if (m <= 32 && k % 128 == 0 && zl_dense_m_bytes(n, k) > 0) {
        // a decode batch: the same product (zl_gemm_nt's bits) from the packed copy
        if (pimpl->packed_of != pimpl->weight.data()) {
            pimpl->packed = ctx.tensor({(size_t)zl_dense_m_bytes(n, k)}, DataType::kInt8, "lm_head_zld16m");
            ZL_CK(zl_dense_pack_m(pimpl->weight.data<uint16_t>(), (uint16_t*)pimpl->packed.data(), n, k, st), "lm_head (ZLD16M pack)");
            pimpl->packed_of = pimpl->weight.data();
        }
        ZL_CK(zl_gemm_nt_packed(input.data<uint16_t>(), k, (const uint16_t*)pimpl->packed.data(), nullptr, local.data<uint16_t>(), m, n, k, alpha, pimpl->zdt(), st),
              "lm_head (packed)");
    } else
        ZL_CK(zl_gemm_nt(input.data<uint16_t>(), k, pimpl->weight.data<uint16_t>(), nullptr, local.data<uint16_t>(), m, n, k, alpha, pimpl->zdt(), st), "lm_head");
    if (!pimpl->parallel) return local;
    // (world, m, part) -> (m, vocab): rank r's part becomes columns r * part .. of every row (the padding past the vocabulary is dropped)
    const size_t world = (size_t)ctx.world_size(), part = (size_t)n, vocab = (size_t)pimpl->vocab_size;
    core::Tensor all = ctx.tensor({world, (size_t)m, part}, pimpl->dtype);
    bmengine::c10d::NCCLAllGather(ctx, local, all);
    core::Tensor out = ctx.tensor(shape, pimpl->dtype);
    for (size_t r = 0; r < world && r * part < vocab; ++r) {
        const size_t width = std::min(part, vocab - r * part);
        BM_CUDART_ASSERT(hipMemcpy2DAsync(out.data<char>() + r * part * 2, vocab * 2, all.data<char>() + r * (size_t)m * part * 2, part * 2, width * 2, (size_t)m,
                                          hipMemcpyDeviceToDevice, ctx.current_cuda_stream()));
    }
    return out;
}

}  // namespace nn

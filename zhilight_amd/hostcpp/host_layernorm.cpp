// host_layernorm.cpp -- nn::LayerNorm with the REFERENCE's class layout (src/nn/layernorm/layernorm.h:7-34: a core::Layer with a
// pimpl; layernorm.cu:408-485) over zl_rmsnorm / zl_head_norm.  The boundary's own nn::LayerNorm in nn_amd.h is a different class
// under the same name, so nn_amd.cpp leaves its definitions out of this library (-DZL_REF_LAYERNORM_EXTERNAL).
#include "host_common.h"
#include "nn/layernorm/layernorm.h"

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
        BM_ASSERT(scale == 1.0f, "LayerNorm::forward_2: scale 1");
        if (dim_model > 1024) {
            // wider than the per-head kernel serves (DeepSeek-V3's q_lora_rank = 1536): the rows are gathered dense, normed by the row
            // kernel and -- a strided output -- scattered back
            const size_t rows = x.size(0), wb = (size_t)dim_model * 2;
            zl_stream_t st = (zl_stream_t)ctx.current_cuda_stream();
            core::Tensor dense = ctx.tensor({rows, (size_t)dim_model}, x.dtype());
            ZL_CK(zl_copy_2d(x.data(), x.stride(0) * 2, dense.data(), wb, wb, rows, st), "copy_2d(gather rows)");
            const bool direct = out.stride(0) == (size_t)dim_model;
            core::Tensor normed = direct ? out : ctx.tensor({rows, (size_t)dim_model}, x.dtype());
            ZL_CK(zl_rmsnorm(dense.data<uint16_t>(), weight.data<uint16_t>(), normed.data<uint16_t>(), rows, dim_model, eps, 1.0f, nullptr, nullptr, zdt(x), st),
                  "rmsnorm(rows)");
            if (!direct) ZL_CK(zl_copy_2d(normed.data(), wb, out.data(), out.stride(0) * 2, wb, rows, st), "copy_2d(scatter rows)");
            return;
        }
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
    const int64_t rows = (int64_t)(x.numel() / x.size(-1));
    if (core::boundary_fusion_enabled() && x.dtype() == DataType::kHalf && pimpl->scale == 1.0f && rows <= 8 && pimpl->dim_model <= 4096 &&
        pimpl->dim_model % 128 == 0 && x.is_continuous()) {
        // a few decode rows: the W4 GEMVs carry this norm in their prologue (bit-identical to this launch + the GEMV, include/
        // zhilight_amd.h "M = 1..8 with norm_weight").  Hand the result tensor back unlaunched; nn::gptq::gptq_gemm_k_major /
        // gemm_fuse_gate_in recognise it, anybody else who touches it (or the input) gets the ordinary launch first (bm_hip.h DeferredOp).
        core::DeferredOp d;
        d.kind = 1;
        d.y = out.nullable_data(); d.y_bytes = out.nbytes();
        d.x = x.nullable_data(); d.x_bytes = x.nbytes();
        d.y_alive = out.storage_token();
        d.stream = ctx.current_cuda_stream();
        d.norm_w = pimpl->weight.data<uint16_t>(); d.eps = pimpl->eps; d.rows = rows; d.dim = pimpl->dim_model;
        const core::Tensor xin = x, w = pimpl->weight;
        void* yp = out.nullable_data();
        const float eps = pimpl->eps;
        const int dim = pimpl->dim_model, dt = pimpl->zdt(x);
        const hipStream_t st = d.stream;
        d.launch = [xin, w, yp, rows, dim, eps, dt, st]() {
            ZL_CK(zl_rmsnorm((const uint16_t*)xin.nullable_data(), (const uint16_t*)w.nullable_data(), (uint16_t*)yp, rows, dim, eps, 1.0f, nullptr, nullptr, dt,
                             (zl_stream_t)st), "rmsnorm (deferred)");
        };
        core::defer_op(std::move(d));
        return out;
    }
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


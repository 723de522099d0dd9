// host_position.cpp -- nn::RotaryEmbedding (src/nn/position/rotary_embedding.h:8-37; rotary_embedding.cu:673-700) and
// nn::RopePreparer (src/nn/position/rope_preparer.h; rope_preparer.cu:212-233) over zl_rope_cos_sin* + zl_rope_rotate: the cos / sin
// tables of the rows' positions in fp32 (plain and llama3 frequency rules), and the rotation of (possibly strided) head slices.
#include "host_common.h"
#include "model/model_config.hpp"
#include "nn/position/rope_preparer.h"
#include "nn/position/rotary_embedding.h"

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


}  // namespace nn

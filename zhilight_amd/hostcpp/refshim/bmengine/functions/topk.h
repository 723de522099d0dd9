// refshim: bmengine/functions/topk.h -- the top-k layer the batch generator samples with (batch_generator.cpp:184-196): declared so that the
// unit compiles for the report-only link check; the kernel behind it belongs to the scheduler's sampling, off this boundary.
#pragma once
#include "bm_functions.h"
namespace bmengine {
namespace functions {
void bitonic_topk(const core::Context& ctx, const core::Tensor& x, const core::Tensor& out, const core::Tensor& pos);
class TopK : public core::Layer {
    BM_LAYER_DEF(TopK)
    TopK(const core::Context& ctx);
    std::pair<core::Tensor, core::Tensor> forward(const core::Context& ctx, const core::Tensor& inp, int top);
};
}  // namespace functions
}  // namespace bmengine

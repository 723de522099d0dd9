// refshim: bmengine/functions/sort.h.  sort_pair_1d / sort_with_indices_1d (CUB radix sorts in the reference) are DECLARED here so
// that FeedForward's dispatch route compiles; the boundary does not define them yet (build_refcheck lists them as pending).
#pragma once
#include <tuple>
#include "bm_functions.h"
namespace bmengine {
namespace functions {
std::pair<core::Tensor, core::Tensor> sort_pair_1d(const core::Context& ctx, const core::Tensor& keys, const core::Tensor& values, int max_key = 0);
std::pair<core::Tensor, core::Tensor> sort_with_indices_1d(const core::Context& ctx, const core::Tensor& keys, int max_key = 0);
}  // namespace functions
}  // namespace bmengine

// refshim: bmengine/functions/scatter.h.  scatter_update_dim0 (dst[dst_index[j]] = src[src_index[j]]) is DECLARED here so that
// FeedForward's dispatch route compiles; the boundary does not define it yet (build_refcheck lists it as pending).
#pragma once
#include "bm_functions.h"
namespace bmengine {
namespace functions {
void scatter_update_dim0(const core::Context& ctx, core::Tensor& dst, const core::Tensor& dst_index, const core::Tensor& src,
                         const core::Tensor& src_index);
}  // namespace functions
}  // namespace bmengine

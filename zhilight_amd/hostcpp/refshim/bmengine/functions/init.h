// refshim: bmengine/functions/init.h.  arange is DECLARED here so that FeedForward's dispatch route compiles; the boundary does
// not define it yet (build_refcheck lists it as pending).
#pragma once
#include "bm_functions.h"
namespace bmengine {
namespace functions {
core::Tensor arange(const core::Context& ctx, int start, int end, int step = 1);
}  // namespace functions
}  // namespace bmengine

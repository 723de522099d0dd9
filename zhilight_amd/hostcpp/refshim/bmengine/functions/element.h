// refshim: bmengine/functions/element.h.  divide (integer tensor by a float divisor) is DECLARED here so that FeedForward's dispatch
// route compiles; the boundary does not define it yet (build_refcheck lists it as pending).
#pragma once
#include "bm_functions.h"
namespace bmengine {
namespace functions {
core::Tensor divide(const core::Context& ctx, const core::Tensor& a, float divisor);
// (SmoothQuant scale arithmetic of block.cpp:600-670, load time only; pending like divide)
core::Tensor pow(const core::Context& ctx, const core::Tensor& a, float exp);
core::Tensor clamp(const core::Context& ctx, const core::Tensor& a, float min, float max);
}  // namespace functions
}  // namespace bmengine

// refshim: bmengine/functions/softmax.h -- the row softmax the batch generator applies to logits before sampling
// (batch_generator.cpp:1861): declared for the report-only link check; sampling is the scheduler's, off this boundary.
#pragma once
#include "bm_functions.h"
namespace bmengine {
namespace functions {
void softmax(const core::Context& ctx, const core::Tensor& logits, const core::Tensor& output, float temperature = 1.0f);
}  // namespace functions
}  // namespace bmengine

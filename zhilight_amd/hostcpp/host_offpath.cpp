// host_offpath.cpp -- names the reference's host translation units reference that are OFF the MI355X hot-path boundary (Marlin,
// reconstruct_exllama, smooth-quant calibration helpers, the loss / scoring helpers of llama.cpp): definitions that throw,
// so that the library links and a call says what is missing -- plus the two that are ON it under another party's name:
//   deep_gemm_fp8_block_h20_group (3rd/deep_gemm/deep_gemm_api.h): the closed DeepGEMM entry point Fp8Block::forward / grouped_gemm
//       call (linear.cpp:1697-1950; the reference unit is compiled with -DENABLE_DS_DEEP_GEMM) = this boundary's block-scaled FP8 GEMM;
//   model::convert_fp32 (src/model/model_util.h) = functions::typecast.
#include "host_common.h"

#include "bmengine/functions/all.h"
#include "model/model_util.h"
#include "nn/embedding/embedding.h"
#include "nn/linear/linear.h"
#include "nn/quant/gptq/gptq.h"
#include "nn/quant/marlin/marlin.h"
#include "3rd/deep_gemm/deep_gemm_api.h"

#define ZL_OFF_BOUNDARY(what) \
    throw BMEngineException(std::string(what) + " is not on the MI355X hot-path boundary (SURVEY.md section 8: out of scope)", __FILE__, __LINE__, __func__)

core::Tensor gptq_marlin_repack(const core::Context&, core::Tensor&, core::Tensor&, size_t, size_t, int64_t) { ZL_OFF_BOUNDARY("gptq_marlin_repack"); }
core::Tensor gptq_marlin_gemm(const core::Context&, const core::Tensor&, core::Tensor&, core::Tensor&, core::Tensor&, core::Tensor&,
                              core::Tensor&, core::Tensor&, size_t, size_t, size_t, bool, bool, bool) {
    ZL_OFF_BOUNDARY("gptq_marlin_gemm");
}
namespace nn {
namespace gptq {
// (nn::gptq::gptq_gemm and reconstruct_gptq -- the GPTQ_KERNEL_ALGO=0 route zhilight/quant.py:73-76 selects for desc_act
//  checkpoints -- are ON the path: nn_amd.cpp)
void reconstruct_exllama(const uint32_t*, const uint32_t*, const half*, const int*, half*, int, int, int, const cudaStream_t, int, int) {
    ZL_OFF_BOUNDARY("nn::gptq::reconstruct_exllama (Int4GPTQ::get_dequant_weight without the k-major layout; use dequant_k_major)");
}
}  // namespace gptq

std::tuple<float, core::Tensor> log_prob_raw(const core::Context&, const core::Tensor&, const core::Tensor&, int32_t) { ZL_OFF_PATH("nn::log_prob_raw (scoring)"); }
int greedy_match_raw(const core::Context&, const core::Tensor&, const core::Tensor&, int32_t) { ZL_OFF_PATH("nn::greedy_match_raw (scoring)"); }
std::tuple<float, core::Tensor> cross_entropy_raw(const core::Context&, const core::Tensor&, const core::Tensor&, int32_t, float) {
    ZL_OFF_PATH("nn::cross_entropy_raw (loss)");
}
}  // namespace nn

// The C signature carries neither aligned_m nor the output type: aligned_m = round_up(m, 4) (per_token_cast_to_fp8), output bf16
// (DeepGEMM's only one).
extern "C" int deep_gemm_fp8_block_h20_group(void* lhs, void* lhs_scales, void* rhs, void* rhs_scales, void* out, void* grouped_layout,
                                             void* stream, int m, int n, int k, int /*block_m*/, int num_groups) {
    const int st = zl_fp8_block_gemm_group((const uint8_t*)lhs, (const float*)lhs_scales, (m + 3) / 4 * 4, (const uint8_t*)rhs,
                                           (const float*)rhs_scales, (const int32_t*)grouped_layout, (uint16_t*)out, m, n, k, num_groups,
                                           ZL_BF16, (zl_stream_t)stream);
    return st == 0 ? 0 : -1;
}

namespace model {
core::Tensor convert_fp32(const core::Context& ctx, const core::Tensor& logits) { return bmengine::functions::typecast(ctx, logits, DataType::kFloat); }
}  // namespace model

// bmengine::functions helpers only the smooth-quant calibration uses: declared by the shim, not on this path
namespace bmengine {
namespace functions {
core::Tensor pow(const core::Context&, const core::Tensor&, float) { ZL_OFF_PATH("functions::pow (smooth-quant calibration)"); }
core::Tensor clamp(const core::Context&, const core::Tensor&, float, float) { ZL_OFF_PATH("functions::clamp (smooth-quant calibration)"); }
}  // namespace functions
}  // namespace bmengine

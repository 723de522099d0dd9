// nn_amd.h -- the free functions / layers ZhiLight's hot-path host code calls, with the reference's signatures, on top
// of the MI355X C ABI (include/zhilight_amd.h).  A maintainer swaps the CUDA translation units behind these names for
// nn_amd.cpp; the callers (linear.cpp, attention.cpp, feedforward.cpp, block.cpp, layernorm users) stay as they are.
// Every wrapper: checks shapes / dtypes with BM_ASSERT (-> BMEngineException, as the reference), allocates its outputs
// with ctx.tensor(...), launches on ctx.current_stream(), turns a non-zero C-ABI status into BMEngineException.
//
// reference header                                   | names provided here
//   src/nn/quant/gptq/gptq.h:10-184                  | nn::gptq::{gptq_gemm_k_major, gemm_fuse_gate_in, dequant_k_major, gptq_shuffle,
//                                                    |            increase_zero, q4_to_q8, un_shuffle, shuffle_awq, gemm_moe_up, gemm_moe_down}
//                                                    |            + amd_pack_k_major, amd_pack_moe
//   src/nn/quant/int8/quant_kernel.h:15-128          | int8_op::{quant_calc_scale x2, set_quant_scale, quant_scale_back, quant_scale_back3,
//                                                    |           layernorm_quant, quant_back_element_add_scale, quant_back_transpose,
//                                                    |           quant_back_act_mul, quant_back_copy_to_buffer} + int8_gemm_nt
//   src/nn/attention/attention_kernel.h:55-80        | nn::{AttentionWorkspace, get_mqa_workspace, multi_query_attention_rag_buffer}
//   src/nn/position/rotary_embedding.h:34-61         | nn::{rotary_embedding_qk, rope_qk_cache}
//   src/kvcache/ragged_buffer_kernel.h:27-36         | nn::copy_to_rag_buffer2
//   src/nn/block/block_kernel.h                      | nn::{element_add_scale, element_add_scale_out}
//   src/nn/linear/activation_kernel.h:8-16           | nn::gate_mul_inplace
//   src/nn/layernorm/layernorm.h:7-37                | nn::LayerNorm {forward, fuse_add, inplace}
//   src/nn/quant/gptq/gptq.h:150-160                  | nn::gptq::{int32_to_int16, reverse_perm, permute_input}
//   src/nn/quant/fp8/fp8.h:7-33                       | nn::fp8::{cvt_half_to_fp8, calc_scale, dynamic_scaled_quant, per_token_cast_to_fp8,
//                                                    |           dequant_fp8_block_weight} + fp8_block_gemm (deep_gemm_fp8_block_h20_group)
//   src/nn/feedforward/ff_kernel.h:16-40              | nn::{top_k_softmax, group_topk_softmax}
//   src/nn/linear/activation_kernel.h:8-10, nn/functions/functions.h:60 | nn::{gelu_inplace, silu_inplace, multiply}
#pragma once
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include <hip/hip_fp16.h>   // __half: the reference spells reconstruct_gptq's operands `half`

#include "bm_hip.h"

namespace nn {
using namespace bmengine;

namespace gptq {

// ---- load-time transforms (Int4GPTQ::load_state_dict, linear.cpp:1162-1244) -----------------------------------------
void gptq_shuffle(const core::Context& ctx, core::Tensor& q_weight, core::Tensor q_perm);   // (K/8, N) int32; q_perm (K) int32 = argsort(g_idx) or empty
void un_shuffle(const core::Context& ctx, core::Tensor& input);                             // AWQ nibble order -> natural, in place
void increase_zero(const core::Context& ctx, core::Tensor& input);                          // every nibble + 1 (15 wraps to 0), in place
core::Tensor shuffle_awq(const core::Context& ctx, core::Tensor& input, bool use_exllama);  // (K, N/8) -> (K/8, N)
core::Tensor q4_to_q8(const core::Context& ctx, const core::Tensor& input);                 // int32 words -> one byte per nibble
// act-order (desc_act): g_idx -> argsort on the host (Int4GPTQ::argsort_cpu) -> the 16-bit permutation and its inverse,
// both returned as kInt32 tensors with the last dimension halved (two uint16 per word), as the reference stores them
core::Tensor int32_to_int16(const core::Context& ctx, const core::Tensor& input);
core::Tensor reverse_perm(const core::Context& ctx, const core::Tensor& input);             // out[input[i]] = i
core::Tensor permute_input(const core::Context& ctx, const core::Tensor& input, const core::Tensor& q_perm);   // out[m, k] = in[m, q_perm[k]]

// ---- the legacy route, GPTQ_KERNEL_ALGO=0 (src/nn/quant/gptq/gptq.h:10-22, 62-72; q_gemm.cu:874-918, 641-700) ------------------
// what Int4GPTQ::forward calls without the k-major layout (linear.cpp:1000) -- zhilight/quant.py:73-76 selects it for every
// desc_act checkpoint.  b_q_weight (K/8, N) as gptq_shuffle left it (use_exllama) or in checkpoint order (otherwise), qzeros
// (K/G, N/8) + 1, scales (K/G, N), b_g_idx = argsort(g_idx) (use_exllama, or empty) / the raw g_idx.  Computed on the k-major
// kernels (see nn_amd.cpp); reconstruct_gptq is bit-exact.
core::Tensor gptq_gemm(const core::Context& ctx, core::Tensor a, core::Tensor b_q_weight, core::Tensor b_gptq_qzeros,
                       core::Tensor b_gptq_scales, core::Tensor b_g_idx, bool use_exllama, int group_size, int size_n1, int size_n2);
void reconstruct_gptq(const uint32_t* b_q_weight, const uint32_t* b_gptq_qzeros, const __half* b_gptq_scales, const int* b_g_idx,
                      __half* out, int height, int width, int num_group, const hipStream_t stream);

// ---- the k-major GEMM family ---------------------------------------------------------------------------------------
// q_weight (N, K/8) int32 exllama-shuffled words, qzeros (N, K/G) int8 (already +1), scales (N, K/G) half: the operands
// Int4GPTQ keeps after preprocess_weight + transpose_weight.  The MI355X kernels stream a re-tiled copy of them; it is
// built by amd_pack_k_major ONCE per weight (the one line a maintainer adds at the end of Int4GPTQ::load_state_dict),
// which returns the packed tensors behind the same three names: q_weight' (N, K/8) int32 = ZLW4M nibble tiles, qzeros'
// EMPTY, scales' (N, K/128) int32 = ZLW4M meta words.  gptq_gemm_k_major recognises the packed form by scales.dtype() ==
// kInt32; handed the raw k-major operands (the path a completely unmodified caller takes -- the reference's own
// linear.cpp, compiled as it is, runs this way: hostcpp/refshim, tests/test_gpu_refcompile.py) it re-tiles them on first
// sight into the weight-identity cache below.  q_perm / rev_perm (act-order): the 16-bit permutations Int4GPTQ keeps
// (int32_to_int16 / reverse_perm of argsort(g_idx)); the activations are gathered through q_perm in front of the GEMM
// (q_gemm_k_major.cu:1094,1104-1106), the rows were regrouped by gptq_shuffle at load.
struct PackedW4 {
    core::Tensor q_weight, qzeros, scales;
};
PackedW4 amd_pack_k_major(const core::Context& ctx, const core::Tensor& q_weight, const core::Tensor& qzeros,
                          const core::Tensor& scales, bool row_interleave = false);

// A raw k-major weight handed to gptq_gemm_k_major / gemm_fuse_gate_in / gemm_moe_* (the path a completely unmodified
// caller takes) is re-tiled ONCE: the packed form is kept in a process-wide cache keyed by the identity of the operand
// tensors (data pointers + shape); the entry holds a reference to the raw tensors, so an address cannot be recycled
// while its entry lives.  Cost: the raw copy stays resident next to the packed one -- callers that can spare one line
// call amd_pack_k_major at load and drop the raw operands instead.  amd_weight_cache_clear() drops every entry (call it
// when a model is unloaded); amd_weight_cache_size() counts them.
void amd_weight_cache_clear();
size_t amd_weight_cache_size();

// out_type 0: W16 (N, K) half.  out_type 2 (W4_FP8_ALGO): E4M3 codes of W16 with the (1,) scale calc_w4a8_scale / the caller
// left on q_weight.quant_scale.  out_type 1 (W4A8, q_gemm_k_major.cu:907-952 with KERNEL_dequant<int8_t,1>): int8 codes
// (N, K) = nearbyintf(W16 / scale[n]); like the reference it needs q_weight.quant_scale, set by calc_w4a8_scale below
// (Int4GPTQ::calc_w4a8_scale, linear.cpp:1101-1112: scale[n] = max_k |W16[n,k]| / 127, float).
core::Tensor dequant_k_major(const core::Context& ctx, const core::Tensor& q_weight, const core::Tensor& qzeros,
                             const core::Tensor& scales, int out_type = 0);
void calc_w4a8_scale(const core::Context& ctx, core::Tensor& q_weight, const core::Tensor& qzeros, const core::Tensor& scales);

core::Tensor gptq_gemm_k_major(const core::Context& ctx,
                               const core::Tensor& a,          // (M, K)
                               const core::Tensor& q_weight,   // (N, K / 8)
                               const core::Tensor& qzeros,     // (N, K / group_size)
                               const core::Tensor& scales,     // (N, K / group_size)
                               const core::Tensor& q_perm,     // (K)
                               const core::Tensor& rev_perm,   // (K)
                               const core::Tensor* bias, bool sym, bool cache_only = false, core::Tensor* output = nullptr,
                               const core::Tensor* precomputed_w8 = nullptr);

core::Tensor gemm_fuse_gate_in(const core::Context& ctx, const core::Tensor& a, const core::Tensor& q_weight1,
                               const core::Tensor& qzeros1, const core::Tensor& scales1, const core::Tensor& rev_perm1,
                               const core::Tensor& q_weight2, const core::Tensor& qzeros2, const core::Tensor& scales2,
                               const core::Tensor& rev_perm2, bool sym);

// FUSE_GPTQ_MOE (src/nn/quant/gptq/gptq.h: gemm_moe_up / gemm_moe_down, q_gemm_k_major.cu:392-520).  The operands are the
// reference's (EXP, N, K/8) int32 / (EXP, N, K/G) int8 / (EXP, N, K/G) half stacks; this repository's kernels want every
// expert packed (zl_w4_pack; gate / up row-interleaved) and stacked: amd_pack_moe does that once at load
// (FeedForward::load_state_dict), the *_packed entry points take its result.  The entry points with the reference's
// signatures pack temporaries on every call (correct, slow: what an unmodified caller gets).
struct PackedMoE {
    core::Tensor q_weight, scales, zeros;      // (EXP, bytes of one packed expert) each
    int64_t experts = 0, n = 0, k = 0, group_size = 0;
    bool row_interleave = false;
};
PackedMoE amd_pack_moe(const core::Context& ctx, const core::Tensor& q_weight, const core::Tensor& qzeros, const core::Tensor& scales,
                       const core::Tensor* q_weight2 = nullptr, const core::Tensor* qzeros2 = nullptr,
                       const core::Tensor* scales2 = nullptr);      // with the second set: [gate; up], row-interleaved
core::Tensor gemm_moe_up_packed(const core::Context& ctx, const core::Tensor& a, const PackedMoE& w, const core::Tensor& expert_ids,
                                int n_shared_expert, bool exp_parallel);
core::Tensor gemm_moe_down_packed(const core::Context& ctx, const core::Tensor& a, const PackedMoE& w, const core::Tensor& expert_ids,
                                  const core::Tensor& expert_weights, int n_shared_expert, bool exp_parallel,
                                  core::Tensor* output = nullptr);     // output given: ADD_C (accumulates into it)
core::Tensor gemm_moe_up(const core::Context& ctx, const core::Tensor& a, const core::Tensor& q_weight1, const core::Tensor& qzeros1,
                         const core::Tensor& scales1, const core::Tensor& rev_perm1, const core::Tensor& q_weight2,
                         const core::Tensor& qzeros2, const core::Tensor& scales2, const core::Tensor& rev_perm2, bool sym,
                         const core::Tensor& expert_ids, int n_shared_expert, bool exp_parallel);
core::Tensor gemm_moe_down(const core::Context& ctx, const core::Tensor& a, const core::Tensor& q_weight, const core::Tensor& qzeros,
                           const core::Tensor& scales, const core::Tensor& expert_ids, const core::Tensor& expert_weights, bool sym,
                           int n_shared_expert, bool exp_parallel, core::Tensor* output = nullptr);
}  // namespace gptq

namespace awq {
// AWQ tensors as stored (src/nn/quant/awq/awq.h:10-25): _kernel (K, N/8) int32, _scaling_factors (K/G, N) half,
// _zeros (K/G, N/8) int32.  thx / thy are launch shapes of the CUDA kernel and ignored here.
core::Tensor awq_dequantize(const core::Context& ctx, core::Tensor _kernel, core::Tensor _scaling_factors, core::Tensor _zeros,
                            int split_k_iters, int thx, int thy);
core::Tensor awq_gemm(const core::Context& ctx, core::Tensor _in_feats, core::Tensor _kernel, core::Tensor _scaling_factors,
                      core::Tensor _zeros, size_t split_k_iters);
}  // namespace awq

namespace fp8 {
// W4A8 with FP8 activations (W4_FP8_ALGO=1: gptq_gemm_k_major's first branch, q_gemm_k_major.cu:1003-1035).  Scales are (1,)
// fp32 DEVICE tensors = amax / MAX_E4M3; codes are OCP E4M3FN bytes of T(x) * T(1 / scale), round-to-nearest-even, saturating
// (fp8_util.cu).  dynamic_scaled_quant returns the codes with the scale attached as quant_scale.
core::Tensor calc_scale(const core::Context& ctx, const core::Tensor& input, float MAX_E4M3 = 448);
core::Tensor dynamic_scaled_quant(const core::Context& ctx, const core::Tensor& input, float MAX_E4M3 = 448);
core::Tensor cvt_half_to_fp8(const core::Context& ctx, const core::Tensor& input, const core::Tensor& scale);
// FP8 128x128-block linear of config 5 (fp8.h:21-33, fp8_util.cu:229-385): per-token 1x128 activation scales (the codes carry
// them as quant_scale: (n/128, aligned_m) column-major or (aligned_m, n/128)), block dequantisation of a weight
core::Tensor per_token_cast_to_fp8(const core::Context& ctx, const core::Tensor& input, bool scale_col_major = true, float MAX_E4M3 = 448);
core::Tensor dequant_fp8_block_weight(const core::Context& ctx, const core::Tensor& input, const core::Tensor& scale, core::DataType out_type);
// the product Fp8Block::forward / grouped_gemm hand to deep_gemm_fp8_block_h20_group (linear.cpp:1863-1945); weight (N, K) or
// (G, N, K) codes (kInt8 / kFP8_E4M3 storage), weight_scale (ceil(N/128), K/128) fp32 [x G], m_indices (M) int32 for G > 1
core::Tensor fp8_block_gemm(const core::Context& ctx, const core::Tensor& a_quant, const core::Tensor& weight, const core::Tensor& weight_scale,
                            core::DataType out_type, const core::Tensor* m_indices = nullptr, core::Tensor* output = nullptr);
}  // namespace fp8

// ---- MoE router (src/nn/feedforward/ff_kernel.h:16-40) -------------------------------------------------------------------
std::tuple<core::Tensor, core::Tensor> top_k_softmax(const core::Context& ctx, const core::Tensor& input, const core::Tensor& worker_load,
                                                     const core::Tensor& expert_load, int k, int k_ext, bool norm_topk_prob, float weight_scale,
                                                     const std::string& scoring_func);
std::tuple<core::Tensor, core::Tensor> group_topk_softmax(const core::Context& ctx, const core::Tensor& input,
                                                          const core::Tensor& score_correction_bias, const core::Tensor& worker_load,
                                                          const core::Tensor& expert_load, int num_group, int topk_group, int top_k, int top_k_ext,
                                                          bool norm_topk_prob, float weight_scale, const std::string& scoring_func);
// dispatch / combine of the prompt-side MoE path (ff_kernel.h:42-96); all_loads = HOST copy of [expert loads | rank loads]
core::Tensor sum_experts(const core::Context& ctx, const core::Tensor& input, const core::Tensor& index, const core::Tensor& weights);
core::Tensor sum_experts(const core::Context& ctx, std::vector<core::Tensor> inputs, const core::Tensor& concat_inputs, const core::Tensor& experts,
                         const core::Tensor& index, const core::Tensor& weights, bool exp_parallel, int world_size = 0, int local_rank = 0);
void route_shared_lb(const core::Context& ctx, core::Tensor& exp_ids, core::Tensor& exp_weights, core::Tensor& worker_load,
                     core::Tensor& expert_load, int top_k, int num_local_experts);
core::Tensor plus_for_sort(const core::Context& ctx, core::Tensor& exp_ids, int num_experts);
core::Tensor calc_reverse_idx(const core::Context& ctx, core::Tensor& exp_ids, core::Tensor& idx, const std::vector<int>& all_loads, int num_experts,
                              bool sorted_by_rank);
std::tuple<core::Tensor, core::Tensor, int> fill_m_indices_padded_indices(const core::Context& ctx, const std::vector<int>& all_loads, int block_m,
                                                                          int num_experts, bool exp_parallel);

void gelu_inplace(const core::Tensor& inp, hipStream_t stream);
void silu_inplace(const core::Tensor& inp, hipStream_t stream);
void multiply(const core::Context& ctx, const core::Tensor& a, float b, core::Tensor* c);   // c = a * T(b)

// ---- attention -----------------------------------------------------------------------------------------------------
struct AttentionWorkspace {
    core::Tensor cache;
    core::Tensor local_max;
    core::Tensor local_sum_exp;
};
AttentionWorkspace get_mqa_workspace(const core::Context& ctx, const core::Tensor& batch_q, int max_len_buf, bool is_quantized);

// attention_kernel.h:39-50 (attention_kernel.cu:1150-1213): every query head has its own kv head -- the same launcher with one
// query head per kv head.  position_bias (ALiBi-style models) and the 192 / 128 MLA head shape are not on this path: refused.
void attention_qkv_rag_buffer(const core::Context& ctx, const core::Tensor& batch_q, const core::Tensor& buf_lens,
                              const core::Tensor& key_buf_addrs, const core::Tensor& val_buf_addrs, const core::Tensor& mask,
                              const core::Tensor& position_bias, float scale, int max_len_buf, core::Tensor& output);
void multi_query_attention_rag_buffer(const core::Context& ctx,
                                      const core::Tensor& batch_q,        // (batch, len_q, num_kv_heads * m_query, dim_head)
                                      const core::Tensor& buf_lens,       // (batch)
                                      const core::Tensor& key_buf_addrs,  // (batch) => (num_kv_heads, len_buf, dim_head)
                                      const core::Tensor& val_buf_addrs,
                                      const core::Tensor& mask,           // (batch) => (len_q, len_buf)
                                      const float scale, const int max_len_buf,
                                      core::Tensor& output,               // like batch_q
                                      const int m_query = 8, int algo_id = -1, const AttentionWorkspace& ws = {},
                                      const core::Tensor& scale_key_addrs = core::Tensor(),
                                      const core::Tensor& scale_val_addrs = core::Tensor(),
                                      core::DataType dequant_dtype = core::DataType::kHalf);

// ---- rotary, KV scatter, element-wise ------------------------------------------------------------------------------
void rotary_embedding_qk(const core::Context& ctx, const core::Tensor& pos, const core::Tensor& in, core::Tensor& out_q,
                         core::Tensor& out_k, core::Tensor& out_v, size_t num_heads, size_t num_kv_heads, size_t dim_head,
                         float rope_theta, core::DataType dtype);
void rope_qk_cache(const core::Context& ctx, const core::Tensor& cos, const core::Tensor& sin, const core::Tensor& in,
                   core::Tensor& out_q, core::Tensor& out_k, core::Tensor& out_v, size_t num_heads, size_t num_kv_heads,
                   size_t dim_head, core::DataType dtype, bool neox_style);
void copy_to_rag_buffer2(const core::Context& ctx, const core::Tensor& placement, const core::Tensor& buf_lens,
                         const core::Tensor& k_src, const core::Tensor& v_src, core::Tensor* buf_k_addr, core::Tensor* buf_v_addr,
                         bool is_scale = false);
core::Tensor element_add_scale(const core::Context& ctx, const core::Tensor& a, const core::Tensor& b, float scale,
                               bool scale_residual = true);
void element_add_scale_out(const core::Context& ctx, const core::Tensor& a, const core::Tensor& b, core::Tensor& c, float scale,
                           bool scale_residual = true);
void gate_mul_inplace(const core::Context& ctx, core::Tensor& inp, const core::Tensor& in2, const std::string& gate_type);
// ff_kernel.h:10-14 (ff_kernel.cu:33-78): input (..., 2 * dim_ff) = [in | gated] of a fused projection -> act(in) * gated, the arithmetic
// of gate_mul_inplace (act in fp32, one rounding to T).  Two strided copies split the halves, then the same launch.
core::Tensor gate_fuse(const core::Context& ctx, const core::Tensor& input, const std::string& act_fn_type);

// ---- RMSNorm layer -------------------------------------------------------------------------------------------------
class LayerNorm {
public:
    LayerNorm(const core::Context& ctx, int dim_model, bool quant = false, float eps = 1e-6, float scale = 1.0,
              core::DataType dtype = core::DataType::kHalf, int num_head = 1);
    core::Tensor forward(const core::Context& ctx, const core::Tensor& x);
    core::Tensor fuse_add(const core::Context& ctx, const core::Tensor& a, const core::Tensor& b, core::Tensor& c);   // c = a + b
    void inplace(const core::Context& ctx, core::Tensor& x);
    void load_state_dict(const core::Context& ctx, const std::map<std::string, const core::Tensor>& state_dict,
                         const std::string& prefix, bool allow_missing = false);
    const core::Tensor& weight() const { return weight_; }

private:
    int dim_model_;
    float eps_, scale_;
    core::DataType dtype_;
    core::Tensor weight_;
};
}  // namespace nn

// ---- FlashMLA binding (src/nn/attention/ds_flash_mla_api.h:6-32): the same two functions, same arguments, over zl_mla_decode_attn_paged.
// The two tensors get_mla_metadata returns are opaque to the caller (multi_head_latent_attention.cpp:902-915 only hands them back);
// here the split plan is a host-side function of (B, H, max_len), so they carry nothing the kernel reads.
namespace ds {
using namespace bmengine;
std::tuple<core::Tensor, core::Tensor> get_mla_metadata(const core::Context& ctx, const core::Tensor& seqlens_k, const size_t num_heads_per_head_k,
                                                        const size_t num_heads_k = 1);
// q (B, len_q, H, 576); kcache (num_blocks, 64, 1, 576); seqlens_k (B) int32; block_table (B, max_blocks) int32; out (B, len_q, H, 512).
// Returns (out viewed (B, len_q * H, 1, 512), softmax_lse (B, 1, len_q * H)) like the reference's folded views (.cpp:121-136).
std::tuple<core::Tensor, core::Tensor> mha_fwd_kvcache_mla(const core::Context& ctx, core::Tensor& q, const core::Tensor& kcache,
                                                           const size_t head_size_v, const core::Tensor& seqlens_k, const core::Tensor& block_table,
                                                           const float softmax_scale, bool is_causal, const core::Tensor& tile_scheduler_metadata,
                                                           const core::Tensor& num_splits, core::Tensor out);
}  // namespace ds

namespace int8_op {
using namespace bmengine;

void quant_calc_scale(const core::Context& ctx, const core::Tensor& input, core::Tensor* output, core::Tensor* output_scale,
                      int q_max = 127, int q_zero = 0);
core::Tensor quant_calc_scale(const core::Context& ctx, const core::Tensor& input, int q_max = 127, int q_zero = 0);
void set_quant_scale(core::Tensor& tensor, const core::Tensor& scale);
std::tuple<core::Tensor, core::Tensor> quant_group_32(const core::Context& ctx, const core::Tensor& input);
void dequant_group_32(const core::Context& ctx, const core::Tensor& q, const core::Tensor& scale, core::Tensor* out);
void dequant_sum_quant_g32(const core::Context& ctx, const core::Tensor& my, const core::Tensor& q_others, const core::Tensor& scale_others,
                           core::Tensor* q_sum, core::Tensor* scale_sum);
core::Tensor quant_scale_back(const core::Context& ctx, const core::Tensor& input, const core::Tensor* scale_x,
                              const core::Tensor* scale_y, core::DataType out_type = core::DataType::kDouble,
                              core::Tensor* output = nullptr);
void quant_scale_back3(const core::Context& ctx, const core::Tensor& input, const core::Tensor* scale_x, const core::Tensor* scale_y,
                       int dim_q, int dim_kv, core::Tensor* q, core::Tensor* k, core::Tensor* v);
void layernorm_quant(const core::Context& ctx, const core::Tensor& input, const core::Tensor& weight, core::Tensor* output,
                     core::Tensor* output_int8, core::Tensor* scale_output, float eps, float scale);
core::Tensor quant_back_element_add_scale(const core::Context& ctx, const core::Tensor& input, const core::Tensor* scale_x,
                                          const core::Tensor* scale_y, const core::Tensor& input_b, float scale);
core::Tensor quant_back_transpose(const core::Context& ctx, const core::Tensor& input, const core::Tensor* scale_x,
                                  const core::Tensor* scale_y);
core::Tensor quant_back_act_mul(const core::Context& ctx, const core::Tensor& A, const core::Tensor* a_scale_x,
                                const core::Tensor* a_scale_y, const core::Tensor& B, const core::Tensor* b_scale_x,
                                const core::Tensor* b_scale_y, const std::string& act_type);
void quant_back_copy_to_buffer(const core::Context& ctx, int num_heads, int len_kv, int len_buf, int dim_head,
                               const core::Tensor* placement, const core::Tensor& src, const core::Tensor* scale_x,
                               const core::Tensor* scale_y, const core::Tensor& dst);
// the IMMA product of Int8Linear::forward (linear.cpp:600-616: functions::Gemm int8 x int8^T -> int32)
core::Tensor int8_gemm_nt(const core::Context& ctx, const core::Tensor& a, const core::Tensor& b);
}  // namespace int8_op

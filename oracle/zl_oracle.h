/*
 * zl_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NEVER SHIPPED OR MEASURED AS THE PRODUCT).
 *
 * A plain-C restatement of the arithmetic of ZhiLight's quantized-GEMM + fused-attention decode
 * hot path, used ONLY by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / CPU baseline.  The product path (zhilight_amd/)
 * never links, imports or falls back to anything in this directory.
 *
 * PARITY STATUS: "parity unpinned" for the quantized kernels (GPTQ/AWQ/INT8), the ragged-KV decode
 * attention, fused qkv+RoPE and fused add+RMSNorm: the reference is CUDA-only (cannot be built or
 * run here) and its own tests hold no golden vectors / never exercise those kernels
 * (SURVEY.md 8c).  Each function below cites the reference file:line it restates, including the
 * reference's rounding points.  The parts the reference tests DO pin through their in-test PyTorch
 * models (neox RoPE, softmax attention, gated feed-forward, linear, embedding) are pinned by
 * tests/golden/ fixtures generated from those PyTorch models (tests/golden/gen_from_reference.py).
 *
 * Two flavours where floating point is involved:
 *   R ("reference-faithful"): reproduces the CUDA kernel's rounding points and reduction order
 *     (fp16 hfma2 partial dots, fp32 fma, 32-lane shuffle trees, ...), so that it is bit-for-bit
 *     what the CUDA kernel computes under IEEE arithmetic.
 *   E ("exact"): the same mathematical function accumulated in fp64.
 *
 * dtype codes: 0 = fp16 (IEEE binary16), 1 = bf16.   All tensors are dense row-major.
 * All citations are relative to /root/reference.
 */
#ifndef ZL_ORACLE_H
#define ZL_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar conversions (soft float, no CPU fp16 support assumed) ---- */
uint16_t zlo_f32_to_f16(float f);
uint16_t zlo_f64_to_f16(double d);
float    zlo_f16_to_f32(uint16_t h);
uint16_t zlo_f32_to_bf16(float f);
float    zlo_bf16_to_f32(uint16_t h);
void zlo_f32_to_f16_array(const float* in, uint16_t* out, int64_t n);
void zlo_f16_to_f32_array(const uint16_t* in, float* out, int64_t n);

/* ---- a4: GPTQ / AWQ load-time layout transforms (bit-exact integer work) ---- */
void zlo_gptq_shuffle(uint32_t* qweight, int64_t k8, int64_t n);
void zlo_gptq_increase_zero(uint32_t* qzeros, int64_t nwords);
void zlo_gptq_q4_to_q8(const uint32_t* in, uint8_t* out, int64_t nwords);
void zlo_transpose_u32(const uint32_t* in, uint32_t* out, int64_t rows, int64_t cols);
void zlo_transpose_u16(const uint16_t* in, uint16_t* out, int64_t rows, int64_t cols);
void zlo_transpose_u8(const uint8_t* in, uint8_t* out, int64_t rows, int64_t cols);
void zlo_awq_un_shuffle(uint32_t* q, int64_t dim0, int64_t n);
void zlo_awq_shuffle(const uint32_t* in, uint32_t* out, int64_t k, int64_t n, int use_exllama);
void zlo_gptq_prepare_k_major(const uint32_t* qweight_hf, const uint32_t* qzeros_hf,
                              const uint16_t* scales_hf, int64_t k, int64_t n, int64_t g,
                              uint32_t* qw_km, uint8_t* qz_km, uint16_t* sc_km);
void zlo_gptq_dequant_hf_naive(const uint32_t* qweight_hf, const uint32_t* qzeros_hf,
                               const uint16_t* scales_hf, const int32_t* g_idx,
                               int64_t k, int64_t n, int64_t g, double* w_nk);

/* ---- a2/a3/a5: W4A16 k-major GEMM ---- */
void zlo_gptq_gemm_k_major(const uint16_t* x, const uint32_t* qw, const uint8_t* qz,
                           const uint16_t* sc, const uint16_t* bias, uint16_t* y,
                           int64_t m, int64_t n, int64_t k, int64_t g, int sym, int add_c);
void zlo_gptq_moe_up(const uint16_t* x, const uint32_t* qw1, const uint8_t* qz1, const uint16_t* sc1, const uint32_t* qw2,
                     const uint8_t* qz2, const uint16_t* sc2, const int32_t* ids, uint16_t* out, int64_t m, int64_t n, int64_t k,
                     int64_t g, int sym, int top_k, int n_shared, int shared_base, int exp_parallel, int world, int rank);
void zlo_gptq_moe_down(const uint16_t* a, const uint32_t* qw, const uint8_t* qz, const uint16_t* sc, const int32_t* ids,
                       const float* weights, uint16_t* out, int64_t m, int64_t n, int64_t k, int64_t g, int sym, int top_k,
                       int n_shared, int shared_base, int exp_parallel, int world, int rank, int add_c);
void zlo_gptq_gemm_k_major_exact(const uint16_t* x, const uint32_t* qw, const uint8_t* qz,
                                 const uint16_t* sc, const uint16_t* bias, double* y,
                                 int64_t m, int64_t n, int64_t k, int64_t g, int sym);
void zlo_gptq_dequant_k_major(const uint32_t* qw, const uint8_t* qz, const uint16_t* sc,
                              uint16_t* out, int64_t n, int64_t k, int64_t g);
void zlo_gptq_gemm_fuse_gate_in(const uint16_t* x,
                                const uint32_t* qw1, const uint8_t* qz1, const uint16_t* sc1,
                                const uint32_t* qw2, const uint8_t* qz2, const uint16_t* sc2,
                                uint16_t* y, int64_t m, int64_t n, int64_t k, int64_t g, int sym);

/* ---- a17: RMSNorm (+ fused residual add) ---- */
void zlo_rmsnorm(const uint16_t* x, const uint16_t* w, uint16_t* out, int64_t rows, int64_t dim,
                 float eps, float scale, const uint16_t* x2, uint16_t* out_sum, int dtype);
void zlo_rmsnorm_exact(const uint16_t* x, const uint16_t* w, double* out, int64_t rows, int64_t dim,
                       float eps, float scale, const uint16_t* x2, int dtype);

/* ---- a13: RoPE ---- */
void zlo_rope_cos_sin(const int32_t* pos, float* cosv, float* sinv, int64_t s, int64_t d,
                      float base, int neox);
void zlo_rope_cos_sin_dynamic(const int32_t* pos, const int32_t* seq_len, float* cosv, float* sinv, int64_t s, int64_t d,
                              float base, float factor, float max_pos, int neox);
void zlo_yarn_params(double base, int dim_head, int original_max_position, double factor, int beta_fast, int beta_slow,
                     double attn_factor, int deepseek, double mscale, double mscale_all_dim, float* low, float* high,
                     float* out_mscale);
void zlo_rope_cos_sin_yarn(const int32_t* pos, float* cosv, float* sinv, int64_t s, int64_t d, float base, float factor,
                           float low, float high, float mscale, int neox);
void zlo_head_norm(const uint16_t* x, const uint16_t* w, uint16_t* out, int64_t rows, int64_t heads, int64_t d,
                   int64_t ld_in, int64_t ld_out, float eps, int mode, int dtype);
void zlo_rope_cos_sin_llama3(const int32_t* pos, float* cosv, float* sinv, int64_t s, int64_t d,
                             float base, float factor, float low_freq_factor,
                             float high_freq_factor, float old_context_len, int neox);
void zlo_rotary_embedding_qk(const int32_t* pos, const uint16_t* in, uint16_t* q, uint16_t* k,
                             uint16_t* v, int64_t s, int64_t h, int64_t hkv, int64_t d,
                             float theta, int dtype);
void zlo_rope_qk_cache(const float* cosv, const float* sinv, const uint16_t* in, uint16_t* q,
                       uint16_t* k, uint16_t* v, int64_t s, int64_t h, int64_t hkv, int64_t d,
                       int neox, int dtype);

/* ---- a14: ragged KV scatter ---- */
void zlo_copy_to_rag_buffer2(const int32_t* placement, const int32_t* buf_lens,
                             const uint16_t* k_src, const uint16_t* v_src,
                             uint16_t* const* k_bufs, uint16_t* const* v_bufs,
                             int64_t b, int64_t len_q, int64_t hkv, int64_t d, int bshd);

/* ---- a15: decode attention over ragged KV ---- */
void zlo_mqa_rag_buffer(const uint16_t* q, const int32_t* buf_lens,
                        const uint16_t* const* k_bufs, const uint16_t* const* v_bufs,
                        const int8_t* mask, uint16_t* out, int64_t b, int64_t len_q, int64_t h,
                        int64_t hkv, int64_t d, float scale, int bshd, int dtype);
void zlo_mqa_rag_buffer_split_kv(const uint16_t* q, const int32_t* buf_lens,
                                 const uint16_t* const* k_bufs, const uint16_t* const* v_bufs,
                                 const int8_t* mask, uint16_t* out, int64_t b, int64_t len_q,
                                 int64_t h, int64_t hkv, int64_t d, float scale, int bshd,
                                 int dtype, int num_split);
void zlo_mqa_rag_buffer_exact(const uint16_t* q, const int32_t* buf_lens,
                              const uint16_t* const* k_bufs, const uint16_t* const* v_bufs,
                              const int8_t* mask, double* out, int64_t b, int64_t len_q, int64_t h,
                              int64_t hkv, int64_t d, float scale, int bshd, int dtype);

/* ---- a18: element-wise ---- */
void zlo_element_add_scale(const uint16_t* a, const uint16_t* b, uint16_t* c, int64_t n,
                           float scale, int scale_residual, int dtype);
void zlo_silu_mul(const uint16_t* inp, const uint16_t* in2, uint16_t* out, int64_t n, int dtype);
void zlo_gelu_mul(const uint16_t* inp, const uint16_t* in2, uint16_t* out, int64_t n, int dtype);

/* ---- a22 / a21: embedding, dense NT GEMM (lm_head) ---- */
void zlo_embedding(const int32_t* ids, const uint16_t* weight, uint16_t* out, int64_t s,
                   int64_t dim, int32_t begin, int32_t end, float scale, int dtype);
void zlo_gemm_nt(const uint16_t* x, const uint16_t* w, const uint16_t* bias, uint16_t* y,
                 int64_t m, int64_t n, int64_t k, float alpha, int dtype);
void zlo_gemm_nt_exact(const uint16_t* x, const uint16_t* w, const uint16_t* bias, double* y,
                       int64_t m, int64_t n, int64_t k, float alpha, int dtype);

/* ---- a8..a11: INT8 (W8A8 dynamic per-token) ---- */
void zlo_quant_calc_scale(const uint16_t* x, int8_t* q, float* scale, int64_t m, int64_t k,
                          int dtype);
void zlo_rmsnorm_quant(const uint16_t* x, const uint16_t* w, uint16_t* out, int8_t* q,
                       float* out_scale, int64_t rows, int64_t dim, float eps, float scale,
                       int dtype);
void zlo_int8_gemm_nt(const int8_t* a, const int8_t* b, int32_t* c, int64_t m, int64_t n,
                      int64_t k);
void zlo_quant_scale_back(const int32_t* c, const float* sx, const uint16_t* sy, uint16_t* out,
                          int64_t m, int64_t n, int dtype);
void zlo_quant_back_act_mul(const int32_t* a, const float* asx, const uint16_t* asy,
                            const int32_t* b, const float* bsx, const uint16_t* bsy,
                            uint16_t* out, int64_t m, int64_t n, int act, int dtype);


void zlo_quant_scale_back3(const int32_t* c, const float* sx, const uint16_t* sy, uint16_t* q, uint16_t* k,
                           uint16_t* v, int64_t m, int64_t n, int64_t dim_q, int64_t dim_kv, int dtype);
void zlo_quant_back_element_add_scale(const int32_t* a, const float* sx, const uint16_t* sy, const uint16_t* b,
                                      float scale, uint16_t* out, int64_t m, int64_t n, int dtype);
void zlo_quant_back_transpose(const int32_t* inp, const float* sx, const uint16_t* sy, uint16_t* out,
                              int64_t batch, int64_t len_q, int64_t heads, int64_t d, int dtype);
void zlo_quant_back_copy_to_buffer(const int32_t* src, const float* sx, const uint16_t* sy, const int32_t* placement,
                                   uint16_t* dst, int64_t batch, int64_t len_kv, int64_t heads, int64_t d,
                                   int64_t len_buf, int64_t src_stride, int64_t dst_stride, int64_t place_stride,
                                   int dtype);


void zlo_quant_calc_scale_zp(const uint16_t* x, uint8_t* q, float* scale, int64_t m, int64_t k, int q_zero, int dtype);
void zlo_mqa_rag_buffer_quant_exact(const uint16_t* q, const int32_t* buf_lens, const uint8_t* const* k_bufs,
                                    const uint8_t* const* v_bufs, const float* const* k_scales,
                                    const float* const* v_scales, const int8_t* mask, double* out, int64_t b,
                                    int64_t len_q, int64_t h, int64_t hkv, int64_t d, float scale, int bshd, int dtype);

/* native AWQ (A.9) and W4A8 (q_gemm_k_major.cu:1036-1073) */
void zlo_awq_dequantize(const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales, uint16_t* out,
                        int64_t k, int64_t n, int64_t g);
void zlo_awq_gemm(const uint16_t* x, const uint16_t* w16, uint16_t* y, int64_t m, int64_t n, int64_t k, int64_t split_k_iters);
void zlo_awq_gemm_exact(const uint16_t* x, const uint16_t* w16, double* y, int64_t m, int64_t n, int64_t k);
void zlo_w4a8_weight_to_int8(const uint16_t* w16, int8_t* w8, float* scale, int64_t n, int64_t k);
void zlo_quant_scale_back_f32(const int32_t* c, const float* sx, const float* sy, uint16_t* out, int64_t m, int64_t n);

/* W4A8 with FP8 activations (q_gemm_k_major.cu:1003-1035, fp8_util.cu) */
uint8_t zlo_f32_to_e4m3(float f);
float zlo_e4m3_to_f32(uint8_t c);
void zlo_fp8_calc_scale(const uint16_t* x, int64_t numel, float max_e4m3, float* scale, int dtype);
void zlo_fp8_cvt_half(const uint16_t* x, int64_t numel, float scale, uint8_t* out, int dtype);
void zlo_fp8_gemm_nt(const uint8_t* a, const uint8_t* b, float scale_a, float scale_b, uint16_t* out, int64_t m, int64_t n, int64_t k);

/* INT8-compressed tensor-parallel reduce (quant_reduce_kernel.cu) */
void zlo_quant_group_32(const uint16_t* x, int8_t* q, uint16_t* scale, int64_t groups, int dtype);
void zlo_dequant_sum_quant_g32(const uint16_t* my, const int8_t* q_others, const uint16_t* scale_others, int8_t* out_q,
                               uint16_t* out_scale, int64_t groups, int world, int dtype);
void zlo_dequant_group_32(const int8_t* q, const uint16_t* scale, uint16_t* out, int64_t groups, int dtype);


/* f4 (config 5, first part): FP8 128x128-block linear (fp8_util.cu:229-385, deep_gemm_api.h) and the MoE router (ff_kernel.cu:92-470) */
void zlo_fp8_per_token_cast(const uint16_t* x, int64_t ldx, uint8_t* out, int64_t ld_out, float* scale, int64_t aligned_m,
                            int64_t m, int64_t n, int col_major, float max_e4m3, int dtype);
void zlo_fp8_block_dequant(const uint8_t* w, const float* scale, uint16_t* out, int64_t rows, int64_t cols, int64_t stride_scale, int dtype);
void zlo_fp8_block_gemm(const uint8_t* a, const float* sa, int64_t aligned_m, const uint8_t* w, const float* sw, const int32_t* m_indices,
                        uint16_t* out, int64_t m, int64_t n, int64_t k, int dtype);
void zlo_moe_top_k_softmax(const uint16_t* logits, int64_t tokens, int num_exp, int k, int top_k_ext, int renormalize, float weight_scale,
                           int scoring, int dtype, float* out_v, int32_t* out_idx, int32_t* worker_load, int32_t* expert_load, int num_worker);
void zlo_moe_group_topk(const uint16_t* logits, const float* correction_bias, int64_t tokens, int num_exp, int k, int top_k_ext,
                        int renormalize, float weight_scale, int scoring, int num_group, int topk_group, int dtype, float* out_v,
                        int32_t* out_idx, int32_t* worker_load, int32_t* expert_load, int num_worker);


/* MoE dispatch / combine (ff_kernel.cu:518-1082) */
void zlo_moe_sum_experts(const uint16_t* input, const int32_t* index, const float* weight, uint16_t* out, int64_t seq_len, int k,
                         int64_t dim_model, int dtype);
void zlo_moe_sum_experts_arr(const uint16_t* const* inputs, const int32_t* experts, const int32_t* index, const float* weight, uint16_t* out,
                             int64_t seq_len, int k, int64_t dim_model, int exp_parallel, int world_size, int local_rank, int dtype);
void zlo_moe_route_shared_lb(int32_t* exp_ids, const int32_t* worker_load_base, int32_t* worker_load, int32_t* expert_load, int max_load,
                             int world_size, int64_t seq_len, int top_k, int top_k_ext, int num_local_experts);
void zlo_moe_plus_for_sort(const int32_t* exp_ids, int32_t* out, int multiple, int world_size, int64_t numel);
void zlo_moe_calc_reverse_idx(const int32_t* exp_ids, const int32_t* indices, const int32_t* all_loads, int num_experts, int world_size,
                              int sorted_by_rank, int32_t* expert_offset, int32_t* rev_indices, int64_t numel);
int zlo_moe_fill_m_indices(const int32_t* all_loads, int block_m, int num_experts, int rank, int ws, int32_t* padded_indices, int32_t* m_indices);


/* f4: MLA decode attention over the latent cache (multi_head_latent_attention.cpp:836-872); flavour 0 = E (fp64), 1 = R (open route) */
void zlo_mla_decode_attn(const uint16_t* q_adj, const int32_t* buf_lens, const int32_t* valid_lens, const uint16_t* const* kv_bufs,
                         uint16_t* out, int64_t b, int64_t h, int kv_rank, int rope_dim, float scale, int dtype, int flavour);

#ifdef __cplusplus
}
#endif
#endif

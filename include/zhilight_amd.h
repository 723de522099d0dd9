/*
 * zhilight_amd.h -- C ABI of the MI355X (gfx950) decode hot path.
 *
 * This is the drop-in boundary: one flat `extern "C"` launcher per kernel
 * family of ZhiLight's `src/nn` quantized-GEMM + fused-attention path.  Every launcher
 *   - takes raw DEVICE pointers, sizes and a `hipStream_t` (passed as void*),
 *   - only enqueues work on that stream: it never allocates, never synchronises, reads no environment variable, keeps
 *     no global mutable state and is re-entrant (several host threads, one per GPU, may call concurrently).  Launchers
 *     that split a product over workgroups take their scratch from the CALLER (zl_w4_opts_t::scratch, the attention
 *     workspace); tuning overrides are explicit arguments of the *_ex entry points,
 *   - returns 0 on success, a negative ZL_E* code for an invalid argument, or a positive
 *     hipError_t if the launch itself failed.  Nothing throws.
 * The C++ `nn::` / `gptq::` / `int8_op::` wrappers of the reference (which allocate outputs with
 * `ctx.tensor` and raise BMEngineException) sit directly on top of these; INTEGRATION.md shows the
 * binding.  Each declaration cites the reference interface it replaces (paths relative to the
 * ZhiLight tree).
 *
 * Entry points that come in pairs (round 3 VERDICT item 9):
 *   NAME / NAME_ex   -- NAME_ex is the implementation and takes the launcher options (zl_w4_opts_t, or a tuning struct of its
 *       own) as one more argument; NAME is `return NAME_ex(..., NULL, s)`: the launcher's own choices, no scratch (K-split routes
 *       off).  Both stay: NAME is what a binding written against the reference's operator signature calls
 *       (zl_w4a16_gemm_mfma, zl_w4a16_gemm_tiled, zl_w4a16_qkv_rope_scatter, zl_decode_attn, zl_decode_attn_quant,
 *       zl_w8a8_gemm_phase, zl_mla_decode_attn), NAME_ex what a host that owns scratch memory and knobs calls (ops.py does).
 *   NAME / NAME_h    -- two FORMATS of the same hand-over, not two versions: zl_decode_attn_splits leaves fp32 split partials
 *       (130 floats per record; consumed by zl_w4a16_gemm_attn_merge, the fp16-dequant route, and by zl_decode_attn's own merge),
 *       zl_decode_attn_splits_h half-precision ones (256 B + an 8-byte (max, sum) pair; consumed by
 *       zl_w4a16_gemm_attn_merge_h[_ex] on the integer-plane kernel).  A caller picks the pair that matches its projection
 *       route (ops.attn_merge_plan returns which); mixing them is ZL_ESHAPE / garbage, hence distinct names.
 *
 * dtype codes: ZL_F16 = 0 (IEEE half), ZL_BF16 = 1.  All tensors are dense row-major unless a
 * stride is passed.  "T" below means the activation dtype.
 */
#ifndef ZHILIGHT_AMD_H
#define ZHILIGHT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZL_VERSION 100 /* 0.1.0 */

enum { ZL_F16 = 0, ZL_BF16 = 1, ZL_F32 = 2 /* accepted where a function says so: the MoE routers' logits */ };

enum {
    ZL_OK = 0,
    ZL_EINVAL = -1,   /* null pointer / non-positive size */
    ZL_ESHAPE = -2,   /* shape not supported by the kernel (alignment, divisibility) */
    ZL_EDTYPE = -3,   /* dtype not supported on this path */
    ZL_ELIMIT = -4    /* exceeds a hardware limit (LDS, grid) */
};

typedef void* zl_stream_t; /* hipStream_t */

int zl_version(void);
const char* zl_status_string(int status);
/* number of CUs of the current device (cached per call; used by grid heuristics). >0 or -hipError */
int zl_device_cu_count(void);


/* ------------------------------------------------------------------------------------------------
 * a4  Load-time layout transforms (bit-exact integer work).
 * Replaces nn::gptq::gptq_shuffle / increase_zero / q4_to_q8 / un_shuffle / shuffle_awq
 * (src/nn/quant/gptq/gptq.h:24-49,141-148; kernels utils.cu:25-214, q_gemm.cu:778-791) and
 * functions::Transpose as used by Int4GPTQ::transpose_weight (src/nn/linear/linear.cpp:1085-1099).
 * ---------------------------------------------------------------------------------------------- */
int zl_gptq_shuffle(uint32_t* qweight /* (K/8,N) in place */, int64_t k8, int64_t n, zl_stream_t s);
int zl_gptq_increase_zero(uint32_t* qzeros /* in place */, int64_t nwords, zl_stream_t s);
int zl_gptq_q4_to_q8(const uint32_t* in, uint8_t* out /* 8*nwords bytes */, int64_t nwords, zl_stream_t s);
int zl_transpose_2d(const void* in, void* out, int64_t rows, int64_t cols, int elem_size /*1,2,4*/, zl_stream_t s);
/* nn::gptq::reconstruct_gptq (src/nn/quant/gptq/q_gemm.cu:641-700): the checkpoint-order (K/8, N) weight of the legacy route
 * (GPTQ_KERNEL_ALGO=0 without exllama) dequantised to a dense fp16 (K, N) matrix, out[k][n] = hmul(half(q - zero[g][n]), scale[g][n])
 * with g = g_idx[k] (NULL: k / (K / groups)); qzeros (groups, N/8) nibbles as they sit on the device (after increase_zero). */
int zl_gptq_reconstruct(const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales, const int32_t* g_idx, uint16_t* out,
                        int64_t k, int64_t n, int64_t groups, zl_stream_t s);
int zl_awq_un_shuffle(uint32_t* q /* (dim0,n) in place */, int64_t dim0, int64_t n, zl_stream_t s);
int zl_awq_shuffle(const uint32_t* in /* (K,N/8) */, uint32_t* out /* (K/8,N) */, int64_t k, int64_t n,
                   int use_exllama, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * ZLW4: the gfx950-native packed W4 layout consumed by zl_w4a16_gemm (DESIGN.md "ZLW4 layout").
 * It is produced once at load time from the reference's k-major tensors
 *     qweight (N, K/8) uint32 (exllama-shuffled words), qzeros (N, K/G) uint8, scales (N, K/G) fp16
 * i.e. exactly what Int4GPTQ::preprocess_weight + transpose_weight leave on the device
 * (src/nn/linear/linear.cpp:1139-1160, 1085-1099), so it plays the role of that load-time step.
 *   Kp = K rounded up to 1024, Np = N rounded up to 2, Q = Kp/1024, C = max(1, 256/G)
 *   qw     : uint32 [Np/2][Q][64][4]   word (row 2*pr + lane/32, index (lane%32) + 32*(4q+j))
 *   scales : fp16   [Np/2][Q][2][C][4] scale of group ((1024q + 256j)/G + c) of row 2*pr + h
 *   zeros  : uint16 [Np/2][Q][2][C]    the same 4 groups' zero points, one nibble each (bits 4j)
 * i.e. everything one wave-load (pair pr, load q) needs is contiguous and the three streams advance
 * linearly: 1 KiB of words, 2*C*8 B of scales, 2*C*2 B of zeros per load.
 * Padding rows / words hold q = z = 0, scale = 0 (contribute exactly 0).
 * row_interleave != 0 packs source row (i%2)*(N/2) + i/2 at packed row i, so that a vertically
 * concatenated [gate; up] matrix ends up as (gate_n, up_n) row pairs for the fused silu*mul epilogue.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t n, k, group_size;   /* logical shape */
    int64_t np, kp, q, c;       /* padded shape, loads per row, scale classes */
    int64_t qw_bytes, scales_bytes, zeros_bytes;
} zl_w4_layout_t;

int zl_w4_layout(int64_t n, int64_t k, int64_t group_size, zl_w4_layout_t* out);
int zl_w4_pack(const uint32_t* qweight_km, const uint8_t* qzeros_km, const uint16_t* scales_km,
               int64_t n, int64_t k, int64_t group_size, int row_interleave,
               uint32_t* qw, uint16_t* scales, uint16_t* zeros, zl_stream_t s);
/* inverse of zl_w4_pack's qw/zeros/scales for row n (debug / tests): dequantised fp16 rows,
 * W16[n,k] = rn16(rn16(q - z) * s)  -- nn::gptq::dequant_k_major, out_type 0
 * (src/nn/quant/gptq/q_gemm_k_major.cu:907-952, kernel :843-886) */
int zl_w4_dequant(const uint32_t* qw, const uint16_t* scales, const uint16_t* zeros,
                  int64_t n, int64_t k, int64_t group_size, uint16_t* out /* (N,K) fp16 */, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * f3  Fused MoE GEMVs of the decode path (FUSE_GPTQ_MOE=1): nn::gptq::gemm_moe_up / gemm_moe_down
 * (src/nn/quant/gptq/q_gemm_k_major.cu:392-520, kernels :243-390).  Every expert's matrix is packed by zl_w4_pack on its
 * own (gate / up: the (2 n_ff, K) matrix with row_interleave = 1); expert e starts expert_stride_* BYTES after expert 0 in
 * each of the three arrays (strides >= the zl_w4_layout sizes, multiples of 16 / 8 / 2, scales_stride / 8 == zeros_stride / 2).
 * expert_ids (M, top_k) int32: the token's routed experts; the n_shared shared experts follow at ids shared_base,
 * shared_base + 1, ... (the reference's SHARED_EXP_ID = experts stored - n_shared) with weight 1.  exp_parallel: only experts
 * with id % world_size == rank are local (stored at id / world_size), the others are skipped like in the reference.
 *   up:   out[m, t, n] = half( silu(x_m . gate_e[n]) * (x_m . up_e[n]) ),  out (M, top_k + n_shared, n_ff); skipped experts: 0
 *   down: out[m, n]    = half( sum_t w[m, t] * (a[m, t] . W_e[n]) )  (add_c: + float(out[m, n])),  a (M, top_k + n_shared, K)
 * Arithmetic: DEV_gemm_warp_reduce per (token, expert, row) = the decode GEMV's; down scales the per-lane fp32 partials by
 * the routing weight before the 32-lane tree (one fma per expert), silu in double like the file-local helper -- bit-identical
 * to the CUDA kernels.  1 <= top_k + n_shared <= 32. */
int zl_w4a16_moe_up(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint16_t* scales, const uint16_t* zeros,
                    int64_t expert_stride_qw, int64_t expert_stride_scales, int64_t expert_stride_zeros,
                    const int32_t* expert_ids, uint16_t* out, int64_t m, int64_t n_ff, int64_t k, int64_t group_size,
                    int top_k, int n_shared, int shared_base, int exp_parallel, int world_size, int rank, zl_stream_t s);
int zl_w4a16_moe_down(const uint16_t* a, int64_t lda, const uint32_t* qw, const uint16_t* scales, const uint16_t* zeros,
                      int64_t expert_stride_qw, int64_t expert_stride_scales, int64_t expert_stride_zeros,
                      const int32_t* expert_ids, const float* expert_weights, uint16_t* out, int64_t m, int64_t n,
                      int64_t k, int64_t group_size, int top_k, int n_shared, int shared_base, int exp_parallel,
                      int world_size, int rank, int add_c, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a2/a5  W4A16 GEMM for decode batches, y = x . dequant(W)^T (+ epilogue).
 * Replaces nn::gptq::gptq_gemm_k_major for M <= 40 / KERNEL_gemm_warp_reduce
 * (src/nn/quant/gptq/q_gemm_k_major.cu:957-1116, 127-237) and nn::gptq::gemm_fuse_gate_in (:765-829).
 * Arithmetic per 8-weight word is the reference's: exact fp16 (q - z), two fp16 hfma2 accumulators,
 * f32(lo) + f32(hi), fp32 fma with the group scale, lanes walking words l, l+32, ... and a 32-lane
 * shuffle-down tree -- results are bit-identical to the CUDA kernel's.
 *
 * x fp16 (M, K) row stride ldx elements; y fp16 (M, N) (N/2 columns when ZL_EPI_SILU_MUL).
 * prologue: if norm_weight != NULL the input rows are RMS-normalised on the fly,
 *           x' = T(x * rsqrt(mean(x^2) + eps) * w)  (LayerNorm::forward, src/nn/layernorm/layernorm.cu:10-42)
 * epilogue flags:
 *   ZL_EPI_BIAS      y = half(acc + bias[n])
 *   ZL_EPI_ADD_C     y = half(float(y) + acc + bias)            (kernel's ADD_C)
 *   ZL_EPI_RESIDUAL  y = half( float(residual[m,n]) + float(half(acc + bias)) )
 *                    = element_add_scale_out(residual, linear_out, scale = 1)  (src/nn/block/block_kernel.cu:8-17)
 *   ZL_EPI_SILU_MUL  weights packed with row_interleave: out[m,j] = half(silu(float(half(g))) * float(half(u)))
 *                    = gate_mul_inplace("silu") on the two fp16 linears (src/nn/linear/activation_kernel.cu:70-106)
 *   ZL_EPI_SILU_MUL_F32  same without the intermediate fp16 roundings = KERNEL_gemm_fuse_gate_in (:529-578)
 * ---------------------------------------------------------------------------------------------- */
enum {
    ZL_EPI_BIAS = 1, ZL_EPI_ADD_C = 2, ZL_EPI_RESIDUAL = 4, ZL_EPI_SILU_MUL = 8, ZL_EPI_SILU_MUL_F32 = 16
};

int zl_w4a16_gemm(const uint16_t* x, int64_t ldx,
                  const uint32_t* qw, const uint16_t* scales, const uint16_t* zeros,
                  const uint16_t* bias, const uint16_t* residual, uint16_t* y,
                  int64_t m, int64_t n, int64_t k, int64_t group_size, int sym,
                  const uint16_t* norm_weight, float norm_eps, int epilogue, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a2/a3  W4A16 GEMM on the matrix cores: y = x . dequant(W)^T for decode batches (16 activation rows per
 * weight pass, any M by passes -- the reference also chunks M by 16, q_gemm_k_major.cu:1109-1112).
 * Arithmetic = the reference's fp32-accumulating flavour (the M > 40 branch: dequant_k_major + fp32-compute
 * GEMM, src/nn/quant/gptq/q_gemm_k_major.cu:1083-1100): exact (q - z) in fp16, fp32 group sums, one fp32
 * fma with the group scale; NOT bit-identical to the warp-reduce kernel (zl_w4a16_gemm is), but within
 * fp32 rounding of the exact result.  Same prologue / epilogue flags as zl_w4a16_gemm.
 * ZLW4M layout (zl_w4m_pack from the same k-major tensors; group_size = a multiple of 128):
 *   qw   : u32 [N/16][K/128][64][4]   16-row x 128-k tiles, one dwordx4 per lane = its four MFMA B fragments
 *   meta : u32 [N/16][K/128][16]      per row and tile: f16 scale | f16 -(1024 + zero) << 16
 * zl_w4m_layout reports qw_bytes and the meta size in scales_bytes (zeros_bytes = 0).
 * ---------------------------------------------------------------------------------------------- */
int zl_w4m_layout(int64_t n, int64_t k, int64_t group_size, zl_w4_layout_t* out);
int zl_w4m_pack(const uint32_t* qweight_km, const uint8_t* qzeros_km, const uint16_t* scales_km,
                int64_t n, int64_t k, int64_t group_size, int row_interleave,
                uint32_t* qw, uint32_t* meta, zl_stream_t s);
/* inverse of zl_w4m_pack: the k-major operands (N,K/8) u32 / (N,K/G) u8 zeros (already +1, low nibble) / (N,K/G) f16
 * scales back out of a ZLW4M weight -- what a checker (or dequant_k_major, q_gemm_k_major.cu:907-952) reads. */
int zl_w4m_unpack(const uint32_t* qw, const uint32_t* meta, int64_t n, int64_t k, int64_t group_size, int row_interleave,
                  uint32_t* qweight_km, uint8_t* qzeros_km, uint16_t* scales_km, zl_stream_t s);
int zl_w4a16_gemm_mfma(const uint16_t* x, int64_t ldx,
                       const uint32_t* qw, const uint32_t* meta,
                       const uint16_t* bias, const uint16_t* residual, uint16_t* y,
                       int64_t m, int64_t n, int64_t k, int64_t group_size,
                       const uint16_t* norm_weight, float norm_eps, int epilogue, zl_stream_t s);

/* Options of the W4A16 matrix-core launchers (zero / NULL = the launcher's own choice everywhere).
 *   scratch        device memory for the K-split paths (13..32 rows with K > 8192 on the streaming kernel; short grids of
 *                  the M-tiled kernel): zl_w4a16_scratch_bytes(m, n) bytes, whose first ZL_SCRATCH_HEADER bytes hold
 *                  arrival counters that must be ZERO before the first launch (launches leave them zero) -- one scratch
 *                  per stream that may run such launches concurrently.  NULL / too small: the launchers take their
 *                  unsplit routes (same results, slower for those shapes).
 *   the rest       overrides the tests / micro-benchmarks sweep: tiles per workgroup of the phase kernel (1..8), its K split
 *                  (2 or 4; -1 = off) and the row count it starts at, the row range the phase kernel takes, the row count from
 *                  which the M-tiled kernel takes over, its M-tile height (32 / 64 / 128) and forced split count, k-slices
 *                  and rounds of k_w4a16_mfma. */
#define ZL_SCRATCH_HEADER 65536
typedef struct {
    void* scratch;
    int64_t scratch_bytes;
    int phase_rounds, phase_ksplit, phase_ksplit_min_m, phase_min_m, phase_max_m, phase_small_off;
    int tiled_min_m, tiled_bm, tiled_splitk;
    int mfma_ks, mfma_rounds;
    int small_algo;   /* 1..4 rows: 2 = the integer-plane arithmetic on the loader / consumer engine (w4_engine.hip: LDS-DMA weight
                         ring filled by a loader wave; bit-identical to 0), 0 = the integer-plane kernel (w4_i8p.hip: nibbles expanded to int8, activations as three byte planes of a
                         per-group block-floating integer, v_mfma_i32_16x16x64_i8; the default), 1 = the fp16-dequant kernels of
                         rounds 1-2 (k_w4a16_phase / k_w4a16_mfma) */
    int tiled_wide;   /* prompt-chunk tiles (128 / 256 x 256 outputs per workgroup, M >= 128): 0 default (on, height by cost model), -1 off,
                         1 also for 32 < M < 128, 2 128-row tiles only, 3 256-row tiles whenever M > 128, 4 192-column tiles, 5 128 x 128 tiles without
                         a K split (the cost model picks those itself for N = 4096, K = 4096: attn_out of a prompt chunk), 6 no two-height plan (256-row
                         tiles + 192-row tiles in two launches when one height leaves the last round of workgroups partly empty) */
    /* round 6 (appended: zero-initialised structs of older callers keep their meaning) */
    int slab;         /* 5..32 rows without a fused norm on the 2-D K-split tiles of w4_slab.hip: 0 default (on), -1 off */
    int slab_min_m;   /* rows from which the slab kernel takes over (default 3; 1..4 rows with K <= 4096 -- 1..2 up to 16384 -- stay on
                         k_w4a16_i8p, which is asked first; the fused qkv + rotary entry point: 5) */
    int slab_nw, slab_gpw;   /* its geometry, for sweeps: waves per workgroup (4 / 8) and 128-k groups per wave (1 / 2 / 4); 0 = planned */
    int slab_r;       /* ... and 16-column tiles per workgroup (1 / 2 / 4 / 8); 0 = planned */
    int defer_norm;   /* 1: norm_weight with 9..32 rows may take the phase kernel's DEFERRED norm (not zl_rmsnorm's roundings, see
                         below); 0: such calls return ZL_ESHAPE and the caller runs zl_rmsnorm first */
    /* ROW STATISTICS HAND-OFF (5..32 rows): the launch that PRODUCES the residual stream leaves, per row and 16-column tile, the
       sum of squares of the fp16 values it stored; the launch that NORMALISES that stream (norm_weight) reads them instead of
       walking the rows again.  Same rounding points as zl_rmsnorm + GEMM -- T(x rs w) per element, rs = rsqrt_rn(sum / K + eps) --
       with the row's sum taken in the statistics' order (tile sums by a 16-lane butterfly, tiles by lane then a 64-lane
       butterfly) instead of zl_rmsnorm's: rs agrees to an fp32 rounding. */
    const float* row_ss;   /* in:  [m][k / 16] tile sums of x (zl_row_ss, or the producing launch's row_ss_out); with norm_weight */
    float* row_ss_out;     /* out: [m][n / 16] tile sums of the rows this launch stores (plain / bias / ADD_C / residual epilogues of
                                   the slab kernel; other routes ignore it: check zl_w4a16_emits_row_ss) */
} zl_w4_opts_t;
int64_t zl_w4a16_scratch_bytes(int64_t m, int64_t n);
/* tile sums of squares of m rows of k fp16 values, out[row][k / 16] (k % 16 == 0): the stand-alone producer of zl_w4_opts_t::row_ss
 * (the first layer's input; tests) -- bit for bit what a producing GEMM launch leaves in row_ss_out for the same stored rows */
int zl_row_ss(const uint16_t* x, int64_t ldx, int64_t m, int64_t k, float* out, zl_stream_t s);
/* 1 when zl_w4a16_gemm_mfma_ex(m rows, n, k, this epilogue, these options) would fill row_ss_out, 0 when its route ignores it */
int zl_w4a16_emits_row_ss(int64_t m, int64_t n, int64_t k, int64_t group_size, int epilogue, const zl_w4_opts_t* opts);
/* ... and 1 when the same call with norm_weight (or zl_w4a16_qkv_rope_scatter_ex when rope != 0) would consume row_ss */
int zl_w4a16_takes_row_ss(int64_t m, int64_t n, int64_t k, int64_t group_size, int rope, const zl_w4_opts_t* opts);
/* WHICH KERNEL A SHAPE TAKES (defaults; group size a multiple of 128 = the ZLW4M operands of this entry point; M = rows of x):
 *   M = 1..4, K <= 4096 (1..2 rows up to K = 16384)   k_w4a16_i8p   (w4_i8p.hip: integer planes, v_mfma_i32_16x16x64_i8; fused RMSNorm
 *                                                     prologue with norm_weight; the batch-1 decode step's four projections)
 *   M = 5..32 (17..32: K <= 8192, or any K with        k_w4a16_phase (w4_phase.hip: fp16 dequant, activations through LDS in 1024-k
 *     scratch for the K split)                         phases; one row block to 16 rows, two to 32; norm_weight: register-resident
 *                                                     fused RMSNorm up to 8 rows, the DEFERRED norm for 9..32 -- see below)
 *   M = 1..8 with norm_weight and K <= 4096            k_w4a16_phase<NORM> (bit-identical to zl_rmsnorm + this call)
 *   M >= 17 otherwise, M > 32 always                   k_w4a16_gemm_tiled / k_w4a16_gemm_wide (w4_gemm_tiled.hip: the reference's
 *                                                     M > 40 arithmetic W16 = rn16(rn16(q - z) s); 128 / 256-row tiles from M = 128,
 *                                                     split K for short grids)
 *   anything else (odd K tails, huge K)                k_w4a16_mfma (w4_mfma.hip: whole activation block staged in LDS, 16 rows per pass)
 *   group size not a multiple of 128                   not this entry point: zl_w4a16_gemm (w4_gemv.hip, the warp-reduce arithmetic)
 *   M = 5..32 (3..4 where k_w4a16_i8p does not      k_w4a16_slab 
 *     reach: K > 4096) without norm_weight, K % 128 == 0  (w4_slab.hip, round 6: 16 R-column x K-slice tiles, activations by
 *     (scratch for the K split; else the phase kernel)  LDS-DMA into wave-private fragment stores, split-K slabs folded by the last arriver)
 *   M = 5..32 with norm_weight AND zl_w4_opts_t::row_ss, k_w4a16_slab<NORM>: rs from the rows' tile sums, T(x rs w) on the fragments (see
 *     K % 1024 == 0, 2048 < K <= 8192                   zl_w4_opts_t::row_ss; zl_w4a16_takes_row_ss answers without launching)
 * norm_weight with 9..32 rows and zl_w4_opts_t::defer_norm = 1 runs the phase kernel's DEFERRED norm: the staged activation is T(x w) and the row's
 * rsqrt(mean x^2 + eps) multiplies the fp32 totals in the epilogue -- one launch less, but NOT the roundings of zl_rmsnorm + GEMM
 * (T(x w) rs against T(x rs w): ~5e-4 rms of an output, 1e-2 of the largest logit after a few layers of the synthetic network),
 * so the decode step does not use it by default (ZL_DEFER_NORM=1 in zhilight_amd/llama.py).
 * Opt-in routes that were measured and do not pay (zl_w4_opts_t::small_algo = 2, the *_planes entry points, the fused
 * attn_out + gate|up launch) exist only in a ZL_BUILD_EXPERIMENTAL=1 build: see the #ifdef ZL_EXPERIMENTAL blocks below. */
int zl_w4a16_gemm_mfma_ex(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, const uint16_t* bias,
                          const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k, int64_t group_size,
                          const uint16_t* norm_weight, float norm_eps, int epilogue, const zl_w4_opts_t* opts, zl_stream_t s);
/* M > 16 flavour on the same ZLW4M operands (prefill chunks, decode batches > 16): W16 = rn16(rn16(q - z) * s)
 * formed in registers, fp32-accumulating MFMA GEMM -- bit for bit the weights dequant_k_major writes out for
 * the reference's M > 40 branch (q_gemm_k_major.cu:843-952, 1083-1100) without materialising them; K must be
 * a multiple of 128; no norm prologue.  zl_w4a16_gemm_mfma forwards to it for m > 16. */
int zl_w4a16_gemm_tiled(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta,
                        const uint16_t* bias, const uint16_t* residual, uint16_t* y,
                        int64_t m, int64_t n, int64_t k, int64_t group_size, int epilogue, zl_stream_t s);
int zl_w4a16_gemm_tiled_ex(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta,
                           const uint16_t* bias, const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k,
                           int64_t group_size, int epilogue, const zl_w4_opts_t* opts, zl_stream_t s);


/* ------------------------------------------------------------------------------------------------
 * a21  Dense NT GEMM for small M (lm_head, NormalLinear decode): y = T(alpha * x . W^T + bias),
 * fp32 accumulate.  Replaces functions::Gemm::forward (bm/functions/gemm.cpp:505-542) on the decode
 * path: RawEmbedding::projection (src/nn/embedding/embedding.cu:274-289).
 * ---------------------------------------------------------------------------------------------- */
int zl_gemm_nt_small_m(const uint16_t* x, int64_t ldx, const uint16_t* w /* (N,K) */, const uint16_t* bias,
                       uint16_t* y, int64_t m, int64_t n, int64_t k, float alpha, int dtype,
                       const uint16_t* norm_weight, float norm_eps, zl_stream_t s);
/* The same product for more than a handful of rows (lm_head at decode batches > 4, unquantised linears,
 * prompt chunks) on the matrix cores: fp32-accumulating MFMA GEMM, one rounding to T; K % 128 == 0. */
int zl_gemm_nt(const uint16_t* x, int64_t ldx, const uint16_t* w /* (N,K) */, const uint16_t* bias, uint16_t* y,
               int64_t m, int64_t n, int64_t k, float alpha, int dtype, zl_stream_t s);
/* ZLD16M: a dense (N, K) fp16 / bf16 matrix re-tiled for streaming with a few rows -- [N / 16][K / 128][4][64 lanes][8 values], so that a
 * wavefront's load of one MFMA B-fragment step is 1 KiB contiguous (what ZLW4M is for int4 weights).  zl_dense_m_bytes: size of the
 * packed copy (rows padded to 16 with zeros); zl_dense_pack_m: once at load; zl_gemm_nt_packed: zl_gemm_nt's results bit for bit for
 * m <= 32 (the lm_head of a decode batch above the 4 rows the wave-per-row GEMV takes, RawEmbedding::projection, embedding.cu:274-289). */
int64_t zl_dense_m_bytes(int64_t n, int64_t k);
int zl_dense_pack_m(const uint16_t* w, uint16_t* out, int64_t n, int64_t k, zl_stream_t s);
int zl_gemm_nt_packed(const uint16_t* x, int64_t ldx, const uint16_t* wp, const uint16_t* bias, uint16_t* y, int64_t m,
                      int64_t n, int64_t k, float alpha, int dtype, zl_stream_t s);
/* The same product with fp32 OUTPUT and no rounding to T: functions::Gemm / nn::Linear after set_output_type(kFloat) -- the MoE
 * router's logits (src/nn/feedforward/feedforward.cpp:285-286), whose top-k must see the fp32 sums.  K % 8 == 0; small N x M. */
int zl_gemm_nt_f32(const uint16_t* x, int64_t ldx, const uint16_t* w /* (N,K) */, float* y, int64_t m, int64_t n, int64_t k, float alpha,
                   int dtype, zl_stream_t s);

/* lm_head + greedy pick without a separate argmax pass over the logits: the GEMV leaves each wavefront's
 * best (rounded logit, row index) in argmax_ws (zl_argmax_workspace_bytes(m, n) bytes); zl_greedy_advance
 * reduces them per activation row (first index on ties, like torch.argmax) and does the between-steps
 * bookkeeping of a decode batch on the device -- tokens <- pick, positions / placement / valid_lens += 1
 * (any of the five output pointers may be NULL) -- the job of fill_search_tokens on the host in the
 * reference (src/generator/batch_generator.cpp:1226-1335). */
int64_t zl_argmax_workspace_bytes(int64_t m, int64_t n);
int zl_gemm_nt_small_m_argmax(const uint16_t* x, int64_t ldx, const uint16_t* w /* (N,K) */, const uint16_t* bias,
                              uint16_t* y, int64_t m, int64_t n, int64_t k, float alpha, int dtype,
                              const uint16_t* norm_weight, float norm_eps, void* argmax_ws, zl_stream_t s);
int zl_greedy_advance(const void* argmax_ws, int64_t m, int64_t n, int32_t* tokens, int32_t* positions,
                      int32_t* placement, int32_t* valid_lens, int64_t* next_tokens, zl_stream_t s);


/* ------------------------------------------------------------------------------------------------
 * a2 (W4A8, FP8 activations)  gptq_gemm_k_major's W4_FP8_ALGO branch (src/nn/quant/gptq/q_gemm_k_major.cu:1003-1035) for
 * M > W4_A8_M_THRES rows: nn::fp8::calc_scale (one fp32 scale per tensor = max|x| / MAX, src/nn/quant/fp8/fp8_util.cu:100-195),
 * nn::fp8::dynamic_scaled_quant's cast (OCP E4M3FN codes of T(x) * T(1 / scale), round to nearest even, saturating at 448;
 * :56-78, 197-229) -- the same two calls on the dequantised weight matrix give Int4GPTQ::calc_w4a8_scale + KERNEL_dequant<half, 2>
 * (linear.cpp:1124-1129, q_gemm_k_major.cu:843-906; MAX_WEIGHT_E4M3 = 256, MAX_ACT_E4M3 = 448) -- and the fp8 x fp8 GEMM with
 * fp32 accumulation scaled by scale_a * scale_b (functions::Gemm kFP8_E4M3, cuBLASLt there; v_mfma_f32_16x16x32_fp8_fp8 here).
 * K % 64 == 0.  `scale` is a DEVICE pointer throughout (no synchronisation).
 * ---------------------------------------------------------------------------------------------- */
int zl_fp8_calc_scale(const uint16_t* x, int64_t numel, float max_e4m3, float* scale, int dtype, zl_stream_t s);
int zl_fp8_cvt_half(const uint16_t* x, const float* scale, uint8_t* out, int64_t numel, int dtype, zl_stream_t s);
int zl_fp8_gemm_nt(const uint8_t* a /* (M, K) */, const uint8_t* b /* (N, K) */, const float* scale_a, const float* scale_b,
                   uint16_t* out /* (M, N) fp16 */, int64_t m, int64_t n, int64_t k, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a20 (second half)  INT8-compressed tensor-parallel reduce: the three kernels ModelContext::reduce_tp_int8
 * (src/model/model_context.cpp:244-326, REDUCE_TP_INT8_THRES) runs around its send / recv rounds --
 * int8_op::quant_group_32, dequant_sum_quant_g32, dequant_group_32 (src/nn/quant/int8/quant_kernel.h:95-128,
 * quant_reduce_kernel.cu:13-105, 270-330, 107-150).  Values in groups of 32: code = rint(v * 127 / absmax), scale =
 * T(absmax / 127); the owner of a slice adds its OWN unquantised rows to the peers' dequantised codes (in the order the
 * caller stacks them: rank + 1, rank + 2, ... mod world), re-quantises the sum, and every rank dequantises what it gathers.
 * Bit-exact restatements (integer codes and T scales); world_size in {2, 4, 8} like the reference.  The exchange rounds
 * themselves are zl_comm_send / zl_comm_recv inside zl_comm_group_start / _end (include/zhilight_amd_comm.h):
 * zhilight_amd/parallel.py::DirectTPGroup.reduce_tp_int8 composes them.
 * ---------------------------------------------------------------------------------------------- */
int zl_quant_group_32(const uint16_t* x, int8_t* q, uint16_t* scale, int64_t groups, int dtype, zl_stream_t s);
int zl_dequant_sum_quant_g32(const uint16_t* my /* (groups, 32) */, const int8_t* q_others /* (world - 1, groups, 32) */,
                             const uint16_t* scale_others /* (world - 1, groups) */, int8_t* q_sum, uint16_t* scale_sum,
                             int64_t groups, int world_size, int dtype, zl_stream_t s);
int zl_dequant_group_32(const int8_t* q, const uint16_t* scale, uint16_t* out, int64_t groups, int dtype, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a17  RMSNorm and fused residual-add + RMSNorm.
 * Replaces nn::LayerNorm::forward / fuse_add (src/nn/layernorm/layernorm.cu:408-432, 227-302).
 * out = T(v * rsqrt(mean(v^2)+eps) * w / scale), v = f32(x) (+ f32(x2); out_sum = T(v) if given).
 * ---------------------------------------------------------------------------------------------- */
int zl_rmsnorm(const uint16_t* x, const uint16_t* weight, uint16_t* out, int64_t rows, int64_t dim,
               float eps, float scale, const uint16_t* x2, uint16_t* out_sum, int dtype, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a13  RoPE.  Replaces RopePreparer::compute_cos_sin (src/nn/position/rope_preparer.cu:49-69,
 * 124-160), rotary_embedding_qk (rotary_embedding_fuse.cu:70-123) and rope_qk_cache
 * (rotary_embedding_fuse_cache.cu:65-125).  cos/sin are fp32 (S, D).
 * ---------------------------------------------------------------------------------------------- */
int zl_rope_cos_sin(const int32_t* pos, float* cosv, float* sinv, int64_t s_len, int64_t d, float base,
                    int neox, zl_stream_t s);
int zl_rope_cos_sin_llama3(const int32_t* pos, float* cosv, float* sinv, int64_t s_len, int64_t d, float base,
                           float factor, float low_freq_factor, float high_freq_factor,
                           float old_context_len, int neox, zl_stream_t s);
/* dynamic-NTK and YaRN angle tables for the same fused rotation kernels (RotaryEmbedding "dynamic", YarnImpl:
 * src/nn/position/rotary_embedding.cu:19-61, 400-447, 506-553).  dynamic: theta scaled per row from its sequence length
 * (seq_len[t], NULL = pos[t]: a decode row) with the reference's integer exponent dim_head / (dim_head - 2); yarn: inv_freq
 * blended between interpolation and extrapolation over the pair index (low, high from yarn_find_correction_dim, computed by
 * the caller in double like YarnImpl), cos / sin multiplied by mscale.  The reference rotates with these angles in T
 * arithmetic (cos / sin rounded to T first); the fused kernels here rotate in fp32 with one rounding (<= 2 ulp apart). */
int zl_rope_cos_sin_dynamic(const int32_t* pos, const int32_t* seq_len, float* cosv, float* sinv, int64_t s_len, int64_t d,
                            float base, float factor, float max_position_embeddings, int neox, zl_stream_t s);
int zl_rope_cos_sin_yarn(const int32_t* pos, float* cosv, float* sinv, int64_t s_len, int64_t d, float base, float factor,
                         float low, float high, float mscale, int neox, zl_stream_t s);
/* per-head norms of q / k before the rotation (x rows ld_in elements apart, head h at column h * d; in place when out == x):
 *   mode 0: RMSNorm over dim_head with one (d) weight -- Qwen3's q_norm / k_norm (src/nn/attention/attention.cpp:110-113,871-876)
 *   mode 1: KERNEL_layernorm_multi_head, weight (heads, d), mean-subtracted (src/nn/layernorm/layernorm.cu:305-325, use_qk_norm) */
int zl_head_norm(const uint16_t* x, const uint16_t* weight, uint16_t* out, int64_t rows, int64_t heads, int64_t d, int64_t ld_in,
                 int64_t ld_out, float eps, int mode, int dtype, zl_stream_t s);
int zl_rotary_embedding_qk(const int32_t* pos, const uint16_t* in, uint16_t* q, uint16_t* k, uint16_t* v,
                           int64_t s_len, int64_t h, int64_t hkv, int64_t d, float theta, int dtype,
                           zl_stream_t s);
int zl_rope_qk_cache(const float* cosv, const float* sinv, const uint16_t* in, uint16_t* q, uint16_t* k,
                     uint16_t* v, int64_t s_len, int64_t h, int64_t hkv, int64_t d, int neox, int dtype,
                     zl_stream_t s);
/* RotaryEmbedding::rotate / rotate_inplace (src/nn/position/rotary_embedding.h:20-31) on a STRIDED operand: x viewed as
 * (n, heads, d) with element strides (x_stride_n, x_stride_h, 1), rotated by the rows' cached cos / sin (n, d) into out
 * (out_stride_n, out_stride_h, 1); out == x allowed.  MLAImpl rotates the 64 rope dimensions inside q's 192-wide heads and
 * inside the fused qkv_a projection this way (multi_head_latent_attention.cpp:527, 540-541, 586-600).  Same fp32 expression and
 * single rounding as zl_rope_qk_cache. */
int zl_rope_rotate(const float* cosv, const float* sinv, const uint16_t* x, uint16_t* out, int64_t n, int64_t heads, int64_t d,
                   int64_t x_stride_n, int64_t x_stride_h, int64_t out_stride_n, int64_t out_stride_h, int neox, int dtype, zl_stream_t s);
/* valid_lens[b] = 1 + the last visible key of task b's last query row in the concatenated int8 visibility mask the reference hands
 * its search kernels (DynBatchContext::s_mask: task b = len_q rows of buf_lens[b] entries): the bridge from that mask to the
 * prefix-visibility kernels (zl_mla_decode_attn, zl_decode_attn's valid_lens). */
int zl_mask_valid_lens(const int8_t* mask, const int32_t* buf_lens, int32_t* valid_lens, int64_t b, int64_t len_q, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a14  Scatter new K/V rows into the per-task ragged buffers.
 * Replaces nn::copy_to_rag_buffer2 (src/kvcache/ragged_buffer_kernel.cu:254-300, kernel :194-222).
 * k_bufs / v_bufs: DEVICE arrays of B raw pointers (RagBufferContext::buf_k_addr,
 * src/model/rag_buffer_context.h:141-188); layout per task BSHD (len_buf,Hkv,D) or BHSD (Hkv,len_buf,D).
 * ---------------------------------------------------------------------------------------------- */
int zl_copy_to_rag_buffer2(const int32_t* placement, const int32_t* buf_lens, const uint16_t* k_src,
                           const uint16_t* v_src, uint16_t* const* k_bufs, uint16_t* const* v_bufs,
                           int64_t b, int64_t len_q, int64_t hkv, int64_t d, int bshd, zl_stream_t s);
/* the same scatter with the row size in BYTES (row_bytes % 4 == 0) for sources that are not 16-bit: the INT8 KV cache's u8 code rows
 * (dim_head bytes) and its fp32 scale "rows" (4 bytes; copy_to_rag_buffer2(..., is_scale = true), attention.cpp:663-669) as the
 * reference issues them one after the other (zl_quant_copy_to_rag_buffer is the fused form). */
int zl_copy_to_rag_buffer_bytes(const int32_t* placement, const int32_t* buf_lens, const void* k_src, const void* v_src, void* const* k_bufs,
                                void* const* v_bufs, int64_t b, int64_t len_q, int64_t hkv, int64_t row_bytes, int bshd, zl_stream_t s);

/* Fused decode-step front end: split fused qkv rows, rotate q and k with cached cos/sin, write q
 * and scatter k,v straight into the ragged buffers (rope_qk_cache + copy_to_rag_buffer2 in one
 * launch; same roundings: one rounding to T after the fp32 rotation). len_q == 1 per task. */
int zl_rope_scatter_decode(const float* cosv, const float* sinv, const uint16_t* qkv /* (B,(H+2Hkv)D) */,
                           uint16_t* q /* (B,H*D) */, const int32_t* placement /* (B) */,
                           const int32_t* buf_lens, uint16_t* const* k_bufs, uint16_t* const* v_bufs,
                           int64_t b, int64_t h, int64_t hkv, int64_t d, int neox, int bshd, int dtype,
                           zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a15  Decode ("search") attention over ragged KV.
 * Replaces nn::multi_query_attention_rag_buffer / attention_qkv_rag_buffer
 * (src/nn/attention/attention_kernel.cu:1252-1457, 1150-1213; kernels :673-923) and
 * get_mqa_workspace (:1237-1250).
 *   out[b,q,h,:] = softmax_j( mask[b][q,j] ? scale * q.K_j : -inf ) . V      (max init -1e20, Z init 1e-20)
 * q/out (B, len_q, H, D) T; buf_lens (B) int32; k_bufs/v_bufs device arrays of B pointers;
 * mask int8 concatenated per task (len_q x len_buf_b), may be NULL with valid_lens (B) != NULL meaning
 * "the first valid_lens[b] entries are visible" (the greedy/sampling decode case).
 * workspace: zl_decode_attn_workspace_bytes() bytes of device scratch (split-KV partials).
 * ---------------------------------------------------------------------------------------------- */
int64_t zl_decode_attn_workspace_bytes(int64_t b, int64_t len_q, int64_t h, int64_t d, int64_t max_len_buf);
int zl_decode_attn(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                   const uint16_t* const* v_bufs, const int8_t* mask, const int32_t* valid_lens,
                   uint16_t* out, void* workspace, int64_t b, int64_t len_q, int64_t h, int64_t hkv,
                   int64_t d, float scale, int64_t max_len_buf, int bshd, int dtype, zl_stream_t s);
/* algo: 0 = the launcher's choice (matrix-core kernel where it applies), 1 = the VALU split-KV kernel (the one masks,
 * D != 128 and more than 16 query rows per kv head always take) */
int zl_decode_attn_ex(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                      const uint16_t* const* v_bufs, const int8_t* mask, const int32_t* valid_lens,
                      uint16_t* out, void* workspace, int64_t b, int64_t len_q, int64_t h, int64_t hkv,
                      int64_t d, float scale, int64_t max_len_buf, int bshd, int dtype, int algo, zl_stream_t s);

/* Fused decode attention front end (len_q == 1 per task, prefix visibility): rope_qk_cache +
 * copy_to_rag_buffer2 + multi_query_attention_rag_buffer in one pass over the fused qkv rows
 * (B, (H + 2 Hkv) D).  q and the new k are rotated with the cached cos/sin (one rounding to T, as the
 * unfused kernels do), the new k/v row is stored at placement[b] by the workgroup whose KV split owns
 * that slot and consumed from registers, so results equal the three-kernel sequence. */
int zl_decode_attn_fused(const float* cosv, const float* sinv, const uint16_t* qkv, const int32_t* placement,
                         const int32_t* buf_lens, const int32_t* valid_lens, uint16_t* const* k_bufs,
                         uint16_t* const* v_bufs, uint16_t* out, void* workspace, int64_t b, int64_t h,
                         int64_t hkv, int64_t d, float scale, int64_t max_len_buf, int neox, int bshd, int dtype,
                         zl_stream_t s);

/* Fused qkv projection + rotary + KV scatter for the len_q == 1 rows of a decode batch (the north star's "fused
 * qkv+rotary"): zl_w4a16_gemm_mfma of the fused (H + 2 Hkv) D x K projection [optionally with the RMSNorm prologue],
 * then rope_qk_cache (neox, cached cos/sin) on the fp16-rounded outputs and copy_to_rag_buffer2 of the new k / v rows,
 * all in the GEMV epilogue -- bit-identical to zl_w4a16_gemm_mfma + zl_rope_scatter_decode
 * (src/nn/attention/attention.cpp:846-900 issues project_q/k/v, rotary_embedding and copy_to_rag_buffer2 separately).
 * x (M, K) fp16, one row per task; q_out (M, H*D); tables as for zl_rope_scatter_decode.
 * Covers 1 <= M <= 32, D % 32 == 0, norm_weight only with M <= 8 and K <= 4096, K <= 8192 beyond 16 rows;
 * ZL_ESHAPE otherwise (use the two-call sequence). */
int zl_w4a16_qkv_rope_scatter(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta,
                              const uint16_t* bias, const uint16_t* norm_weight, float norm_eps, const float* cosv,
                              const float* sinv, const int32_t* placement, const int32_t* buf_lens,
                              uint16_t* const* k_bufs, uint16_t* const* v_bufs, uint16_t* q_out, int64_t m, int64_t h,
                              int64_t hkv, int64_t d, int64_t k, int64_t group_size, int bshd, zl_stream_t s);
/* the same with the launcher options (zl_w4_opts_t::small_algo selects the kernel family for 1..4 rows) */
int zl_w4a16_qkv_rope_scatter_ex(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta,
                                 const uint16_t* bias, const uint16_t* norm_weight, float norm_eps, const float* cosv,
                                 const float* sinv, const int32_t* placement, const int32_t* buf_lens,
                                 uint16_t* const* k_bufs, uint16_t* const* v_bufs, uint16_t* q_out, int64_t m, int64_t h,
                                 int64_t hkv, int64_t d, int64_t k, int64_t group_size, int bshd, const zl_w4_opts_t* opts,
                                 zl_stream_t s);

#ifdef ZL_EXPERIMENTAL   /* digit-plane route for 5..32 rows: built, exact, not faster -- only in a ZL_BUILD_EXPERIMENTAL=1 build */
/* Decode batches of 5..32 rows on the integer matrix cores (round 4).  The activation matrix becomes digit planes ONCE --
 * zl_w4a16_planes: per row and 128-k group X = rint(x * 2^(36 - Ef)) (Ef: exponent field of the group's largest fp16 magnitude,
 * |X| < 2^22: exact for every value within 2^-12 of the group maximum), three balanced byte digits in the register layout of
 * v_mfma_i32_16x16x64_i8's A operand, plus (2^(Ef - 36), that times sum X) per row and group; with norm_weight the RMSNorm of the
 * row (LayerNorm::forward, src/nn/layernorm/layernorm.cu:10-42) runs first, so the normalised row is never written.  `planes`:
 * zl_w4a16_planes_bytes(m, k) bytes, 16-byte aligned; 1 <= m <= 32, k % 128 == 0, k <= 16384 (ZL_ESHAPE otherwise).
 * zl_w4a16_gemm_planes is zl_w4a16_gemm_mfma_ex on such planes (no norm_weight: it went into the planes): the same ZLW4M
 * operands and epilogues, every 1 KiB weight item expanded to bytes once for all rows, the group sums exact integers -- the
 * branch of gptq_gemm_k_major for 5 <= M <= 40 (src/nn/quant/gptq/q_gemm_k_major.cu:580-686, which streams the weights once per
 * 16 rows) with one pass over the weights for up to 32 rows.  zl_w4a16_qkv_rope_scatter_planes: zl_w4a16_qkv_rope_scatter likewise. */
int64_t zl_w4a16_planes_bytes(int64_t m, int64_t k);
int zl_w4a16_planes(const uint16_t* x, int64_t ldx, int64_t m, int64_t k, const uint16_t* norm_weight, float norm_eps, void* planes,
                    zl_stream_t s);
int zl_w4a16_gemm_planes(const void* planes, const uint32_t* qw, const uint32_t* meta, const uint16_t* bias, const uint16_t* residual,
                         uint16_t* y, int64_t m, int64_t n, int64_t k, int64_t group_size, int epilogue, const zl_w4_opts_t* opts,
                         zl_stream_t s);
int zl_w4a16_qkv_rope_scatter_planes(const void* planes, const uint32_t* qw, const uint32_t* meta, const uint16_t* bias, const float* cosv,
                                     const float* sinv, const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                                     uint16_t* const* v_bufs, uint16_t* q_out, int64_t m, int64_t h, int64_t hkv, int64_t d, int64_t k,
                                     int64_t group_size, int bshd, zl_stream_t s);
#endif  /* ZL_EXPERIMENTAL */

/* Decode attention with the split merge folded into the attention output projection (len_q == 1 per task, prefix
 * visibility, D == 128, H / Hkv <= 16: the matrix-core kernel).  zl_decode_attn_splits is zl_decode_attn without its
 * merge launch: it leaves the split-KV partials in `workspace`; zl_w4a16_gemm_attn_merge is zl_w4a16_gemm_mfma of the
 * attn_out projection (src/nn/attention/attention.cpp:944-958) whose activation rows are merged from those partials in
 * the GEMV prologue, with the merge kernel's arithmetic and order -- bit-identical to zl_decode_attn +
 * zl_w4a16_gemm_mfma, one launch less per layer.  split_len = zl_decode_attn_split_len(b, hkv, max_len_buf),
 * max_splits = ceil(max_len_buf / split_len); the workspace, buf_lens and valid_lens are the ones given to
 * zl_decode_attn_splits.  Covers M = B <= 4, K = H * 128 <= 4096, max_splits <= 16, no gated epilogue, at most two
 * row tiles per CU; ZL_ESHAPE otherwise (use the two-call sequence). */
int64_t zl_decode_attn_split_len(int64_t b, int64_t hkv, int64_t max_len_buf);
/* The same pair with HALF-PRECISION partials (the default of the decode step since round 3): zl_decode_attn_splits_h leaves
 * each split's NORMALISED output row as fp16 plus its fp32 (max, sum) -- 264 instead of 520 bytes per (head, split), and every
 * workgroup of the merging projection re-reads all of them -- and zl_w4a16_gemm_attn_merge_h (the integer-plane kernel,
 * w4_i8p.hip) merges them in its prologue with k_decode_attn_combine's weights.  Not bit-identical to the merge launch (one
 * extra fp16 rounding of the partial rows: within 2^-11 of the largest partial); same coverage and arguments, fp16 only. */
int zl_decode_attn_splits_h(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                            const uint16_t* const* v_bufs, const int32_t* valid_lens, void* workspace, int64_t b, int64_t h,
                            int64_t hkv, int64_t d, float scale, int64_t max_len_buf, int bshd, zl_stream_t s);
/* zl_decode_attn_splits_h with the reference's int8 visibility mask (one row of buf_lens[b] entries per task, concatenated: what
 * nn::multi_query_attention_rag_buffer is handed, attention_kernel.cu:1252-1457) instead of prefix lengths: every split of the
 * BUFFER leaves a record (a split without a visible key: a zero row, (max, sum) = (-1e20, 0)); an invisible key's K / V are never
 * multiplied in.  The merging projection takes these records with valid_lens = buf_lens. */
int zl_decode_attn_splits_h_mask(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                                 const uint16_t* const* v_bufs, const int8_t* mask, void* workspace, int64_t b, int64_t h,
                                 int64_t hkv, int64_t d, float scale, int64_t max_len_buf, int bshd, zl_stream_t s);
/* The merge of the half-precision records as a launch of its own -- zl_w4a16_gemm_attn_merge_h's prologue arithmetic and order, so
 * out (B, H, 128) fp16 holds bit for bit the rows that projection would have multiplied: for a caller whose next call turned out
 * not to be the projection (the host library's deferred merge, hostcpp/nn_amd.cpp).  valid_lens may be NULL (mask form). */
int zl_decode_attn_combine_h(const void* workspace, const int32_t* buf_lens, const int32_t* valid_lens, uint16_t* out, int64_t b,
                             int64_t h, int64_t hkv, int64_t max_len_buf, zl_stream_t s);
int zl_w4a16_gemm_attn_merge_h(const void* attn_workspace, const int32_t* buf_lens, const int32_t* valid_lens,
                               int64_t split_len, int64_t max_splits, const uint32_t* qw, const uint32_t* meta,
                               const uint16_t* bias, const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k,
                               int64_t group_size, int epilogue, zl_stream_t s);
/* zl_w4a16_gemm_attn_merge_h with the launcher options (zl_w4_opts_t::small_algo == 2: the loader / consumer engine) */
int zl_w4a16_gemm_attn_merge_h_ex(const void* attn_workspace, const int32_t* buf_lens, const int32_t* valid_lens,
                                  int64_t split_len, int64_t max_splits, const uint32_t* qw, const uint32_t* meta,
                                  const uint16_t* bias, const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k,
                                  int64_t group_size, int epilogue, const zl_w4_opts_t* opts, zl_stream_t s);
#ifdef ZL_EXPERIMENTAL   /* the loader / consumer engine's fused launch (w4_engine.hip): ties the two launches it replaces */
/* Two projections of a decode layer in ONE launch (w4_engine.hip; 1..4 rows, fp16): the attention split merge + attn_out +
 * residual add into `hidden` (in place: EncoderLayer's first element_add_scale, src/nn/block/block.cpp:104-121), then
 * ln_ff + the fused w_in | w_gated projection + silu.mul into `act` (FeedForward::forward's first half,
 * src/nn/feedforward/feedforward.cpp:107-170).  Same results as zl_w4a16_gemm_attn_merge_h followed by zl_w4a16_gemm_mfma with
 * the fused norm, bit for bit.  The hidden rows cross between the workgroups INSIDE the launch as tagged 8-byte granules:
 *   granules   device memory, m * dim_model * 4 bytes, 8-byte aligned, zero before the first use; may be shared by every layer
 *   epoch      device word the caller advances (zl_engine_epoch_advance, by more than the largest epoch_add in use) between
 *              two uses of the same (granules, epoch_add) pair -- once per decode step; epoch_add = the layer index
 *   err        optional device word: bit 0 = a bounded wait inside a workgroup ran out, bit 1 = the hand-off did (every
 *              workgroup of the launch must be resident at once: dim_model / 16 <= the device's CU count, nothing else
 *              running on the device); the outputs are then undefined -- never a hang
 * ZL_ESHAPE outside m <= 4, dim_attn <= 4096 and dim_model <= 4096 multiples of 1024, max_splits <= 16, n_ff / 16 a multiple
 * (<= 8) of dim_model / 16 (callers fall back to the two launches). */
int zl_w4a16_attn_out_gate_up(const void* attn_workspace, const int32_t* buf_lens, const int32_t* valid_lens, int64_t split_len,
                              int64_t max_splits, const uint32_t* qw_o, const uint32_t* meta_o, const uint16_t* bias_o,
                              uint16_t* hidden, const uint32_t* qw_ff, const uint32_t* meta_ff, const uint16_t* bias_ff,
                              const uint16_t* norm_weight, float norm_eps, uint16_t* act, int64_t m, int64_t dim_model,
                              int64_t dim_attn, int64_t n_ff, int64_t group_size, void* granules, const uint32_t* epoch,
                              uint32_t epoch_add, uint32_t* err, zl_stream_t s);
int zl_engine_epoch_advance(uint32_t* epoch, uint32_t by, zl_stream_t s);
#endif  /* ZL_EXPERIMENTAL */
int zl_decode_attn_splits(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                          const uint16_t* const* v_bufs, const int32_t* valid_lens, void* workspace, int64_t b, int64_t h,
                          int64_t hkv, int64_t d, float scale, int64_t max_len_buf, int bshd, int dtype, zl_stream_t s);
int zl_w4a16_gemm_attn_merge(const void* attn_workspace, const int32_t* buf_lens, const int32_t* valid_lens,
                             int64_t split_len, int64_t max_splits, const uint32_t* qw, const uint32_t* meta,
                             const uint16_t* bias, const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k,
                             int64_t group_size, int epilogue, zl_stream_t s);

/* Decode attention with the split merge INSIDE the launch (round 5; len_q == 1, prefix visibility, D == 128, H / Hkv <= 16).
 * Replaces the same reference pair as zl_decode_attn -- KERNEL_mqa_rag_buffer_split_kv + KERNEL_mqa_combine
 * (src/nn/attention/attention_kernel.cu:729-923, launched :1384-1457) -- with ONE launch and no merging prologue downstream:
 * every (task, kv head, split) workgroup publishes its (acc[128], max, sum) record write-through, counts its arrival on the
 * pair's word, and the pair's LAST arriver merges the records (KERNEL_mqa_combine's formula, k_decode_attn_combine's order:
 * bit-identical to zl_decode_attn for the same split length up to 16 splits) and writes out (B, H, 128) T.  Splits are multiples
 * of 32 keys (one matrix-core chunk per wave; 1 / 2 / 4 waves per workgroup); split_len = 0: zl_decode_attn_la_split_len -- zl_decode_attn's
 * own split length for a few tasks (then the result IS zl_decode_attn's), and as few splits as give every CU one workgroup once
 * tasks x kv heads reach a quarter of the CU count (none at batch 32 x 8 kv heads: no record, no merge at all); at most 64 splits per task.  Measured (round 5,
 * profiles/r05_attn_la_ab.txt): pays from 2 rows up (the merge launch it removes costs 5 us per layer, its own tail 3.6-4);
 * at one row the merging projection (zl_w4a16_gemm_attn_merge_h) stays 1.4 us per layer ahead, and 32- / 64-key splits lose.
 * half_partials bit 0: records as fp16 normalised rows + fp32 (max, sum) (zl_decode_attn_splits_h's format and
 * zl_w4a16_gemm_attn_merge_h's arithmetic; fp16 only); bits 8..15: waves per workgroup for A/B runs (1 / 2 / 4 / 8; 0 = the launcher's:
 * 4 from 128-key splits on, one per 32 keys below; 8 was measured behind 4).
 * workspace: zl_decode_attn_la_workspace_bytes(b, h, hkv, max_len_buf, split_len) bytes, ZERO-INITIALISED ONCE by the caller
 * (the arrival words at its head; every launch leaves them zero), not shared by two launches that may overlap in time. */
int64_t zl_decode_attn_la_split_len(int64_t b, int64_t hkv, int64_t max_len_buf);
int64_t zl_decode_attn_la_workspace_bytes(int64_t b, int64_t h, int64_t hkv, int64_t max_len_buf, int64_t split_len);
int zl_decode_attn_la(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs, const uint16_t* const* v_bufs,
                      const int32_t* valid_lens, uint16_t* out, void* workspace, int64_t b, int64_t h, int64_t hkv, int64_t d,
                      float scale, int64_t max_len_buf, int bshd, int dtype, int64_t split_len, int half_partials, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a15q  INT8 KV cache (RagBufferContext::is_cache_quant, src/model/rag_buffer_context.h:96).
 * A cached K/V row of one kv head is D unsigned codes + one fp32 scale:
 *     code = 128 + rint(x * 127 / amax)   scale = amax / 127      (round half even)
 * i.e. int8_op::quant_calc_scale(ctx, x, 127, 128)  (src/nn/quant/int8/quant_kernel.cu:15-47, :49-80).
 * Per task: codes u8 (len_buf, Hkv, D) [bshd] or (Hkv, len_buf, D); scales fp32 (len_buf, Hkv) or
 * (Hkv, len_buf); both reached through device arrays of B pointers (buf_k_addr / scale_k_addr,
 * src/nn/attention/attention.cpp:664-667).
 *
 * zl_quant_calc_scale_zp        the bare op (q_zero = 128 for the cache, 0 gives zl_quant_calc_scale's codes + 0).
 * zl_quant_copy_to_rag_buffer   quantise k_src / v_src (B*len_q, Hkv, D) rows and scatter codes and scales to
 *                               slot placement[token] of their task: replaces 2 x quant_calc_scale + 2 x
 *                               copy_to_rag_buffer2 (attention.cpp:656-676) and TransformerBuffer::copy for
 *                               a quantised buffer (src/kvcache/transformer_buffer.cu:128-134).
 * zl_rope_quant_scatter_decode  the same with rope_qk_cache in front, over fused qkv rows (len_q == 1):
 *                               rotated q -> q (B, H*D); rotated k is rounded to T before it is quantised.
 * zl_decode_attn_quant          multi_query_attention_rag_buffer with scale_k/scale_v given
 *                               (attention_kernel.cu:1384-1416, KERNEL_mqa_rag_buffer_split_kv_quant :802-878,
 *                               quant_attention.cuh:39-123):
 *     out = softmax_j( mask ? scale * sk_j * q.(K_j - 128) : -inf ) . ( sv_j * (V_j - 128) )
 *                               fp32 accumulation (the reference forms q.(K-128) in fp16); same workspace
 *                               as zl_decode_attn.
 * ---------------------------------------------------------------------------------------------- */
int zl_quant_calc_scale_zp(const uint16_t* x, uint8_t* q, float* scale, int64_t m, int64_t k, int q_zero, int dtype,
                           zl_stream_t s);
/* rows of g codes back to T, out = rn_T((code - q_zero) * scale[row]); q_zero = 128: unsigned cache codes, 0: signed.  Replaces
 * int8_op::dequant_group (src/nn/quant/int8/quant_reduce_kernel.cu:144-190), which TransformerBuffer::copy uses to hand a prompt
 * chunk the already cached rows of a quantised buffer (src/kvcache/transformer_buffer.cu:135-151). */
int zl_dequant_group(const void* q, const float* scale, uint16_t* out, int64_t m, int64_t g, int q_zero, int dtype, zl_stream_t s);
int zl_quant_copy_to_rag_buffer(const int32_t* placement, const int32_t* buf_lens, const uint16_t* k_src,
                                const uint16_t* v_src, uint8_t* const* k_bufs, uint8_t* const* v_bufs,
                                float* const* k_scales, float* const* v_scales, int64_t b, int64_t len_q,
                                int64_t hkv, int64_t d, int bshd, int dtype, zl_stream_t s);
int zl_rope_quant_scatter_decode(const float* cosv, const float* sinv, const uint16_t* qkv, uint16_t* q,
                                 const int32_t* placement, const int32_t* buf_lens, uint8_t* const* k_bufs,
                                 uint8_t* const* v_bufs, float* const* k_scales, float* const* v_scales, int64_t b,
                                 int64_t h, int64_t hkv, int64_t d, int neox, int bshd, int dtype, zl_stream_t s);
int zl_decode_attn_quant(const uint16_t* q, const int32_t* buf_lens, const uint8_t* const* k_bufs,
                         const uint8_t* const* v_bufs, const float* const* k_scales, const float* const* v_scales,
                         const int8_t* mask, const int32_t* valid_lens, uint16_t* out, void* workspace, int64_t b,
                         int64_t len_q, int64_t h, int64_t hkv, int64_t d, float scale, int64_t max_len_buf,
                         int bshd, int dtype, zl_stream_t s);
int zl_decode_attn_quant_ex(const uint16_t* q, const int32_t* buf_lens, const uint8_t* const* k_bufs,
                         const uint8_t* const* v_bufs, const float* const* k_scales, const float* const* v_scales,
                         const int8_t* mask, const int32_t* valid_lens, uint16_t* out, void* workspace, int64_t b,
                         int64_t len_q, int64_t h, int64_t hkv, int64_t d, float scale, int64_t max_len_buf,
                         int bshd, int dtype, int algo, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a16  Prompt ("encode part") attention of ONE task's chunk, causal, on the matrix cores.
 * Replaces attn_encode_group -> FlashDecoding::mha_fwd (src/nn/attention/attention.cpp:442-622; the
 * arithmetic of the external flash-attn library): query row i of the chunk sits at position pos0 + i and
 * sees keys 0 .. pos0 + i of the task's K/V buffer (BSHD (len_buf, Hkv, D) or BHSD), which must already
 * hold the chunk's own rows (copy_to_rag_buffer2 first).  q / out (s_q, H, D) fp16 or bf16; D = 128.  fp32 softmax,
 * probabilities rounded to T into the P.V product, fp32 accumulation.
 * ---------------------------------------------------------------------------------------------- */
int zl_prefill_attn(const uint16_t* q, const uint16_t* k_buf, const uint16_t* v_buf, uint16_t* out, int64_t s_q,
                    int64_t pos0, int64_t h, int64_t hkv, int64_t d, float scale, int64_t len_buf, int bshd,
                    int dtype, zl_stream_t s);
/* groups: wave groups per workgroup, 1 / 2 / 4 (round 5): the key tiles of a 64-query tile are dealt round-robin to `groups` sets
 * of four waves with their own K / V staging and their own (O, max, sum), folded together once at the end -- every workgroup of a
 * prompt is resident at once, so the launch lasts as long as the query tile with the most key tiles, and this cuts that chain
 * `groups`-fold.  0 = the launcher's choice (zl_prefill_attn's): 4 from 8 key tiles on, 2 from 3.  Same arithmetic per tile; the
 * order in which tiles enter a row's running maximum / sum changes with `groups` (within the op's 1e-3 test bar). */
int zl_prefill_attn_ex(const uint16_t* q, const uint16_t* k_buf, const uint16_t* v_buf, uint16_t* out, int64_t s_q,
                       int64_t pos0, int64_t h, int64_t hkv, int64_t d, float scale, int64_t len_buf, int bshd,
                       int dtype, int groups, zl_stream_t s);


/* ------------------------------------------------------------------------------------------------
 * a18  Element-wise.  Replaces nn::element_add_scale_out (src/nn/block/block_kernel.cu:19-50) and
 * nn::gate_mul_inplace (src/nn/linear/activation_kernel.cu:82-106; act 0 = silu, 1 = gelu).
 * ---------------------------------------------------------------------------------------------- */
int zl_element_add_scale(const uint16_t* a, const uint16_t* b, uint16_t* c, int64_t n, float scale,
                         int scale_residual, int dtype, zl_stream_t s);
int zl_gate_mul(const uint16_t* gate, const uint16_t* up, uint16_t* out, int64_t n, int act, int dtype,
                zl_stream_t s);
/* nn::gate_fuse (src/nn/linear/ff_kernel.cu:33-90): out (rows, ff) = act(in[:, :ff]) * in[:, ff:] over the (rows, 2 ff) output of the
 * fused w_in | w_gated linear -- one launch, the arithmetic of zl_gate_mul on the two halves of every row. */
int zl_gate_fuse(const uint16_t* in, uint16_t* out, int64_t rows, int64_t ff, int act, int dtype, zl_stream_t s);
/* a6  act-order (desc_act) checkpoints: out[r, i] = x[r, perm[i]] over 16-bit elements, x rows ldx apart.  Replaces
 * nn::gptq::permute_input (src/nn/quant/gptq/gptq.h:155-159), which gptq_gemm_k_major runs in front of its kernels when
 * q_perm is given (q_gemm_k_major.cu:1095,1105); perm = argsort(g_idx), the order the weight rows were regrouped in. */
int zl_permute_input(const uint16_t* x, int64_t ldx, const int32_t* perm, uint16_t* out, int64_t rows, int64_t k, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a22  Embedding lookup.  Replaces RawEmbedding::forward (src/nn/embedding/embedding.cu:260-272,
 * kernel :23-44): out = in-range ? T(f32(weight[id-begin]) * scale) : 0.
 * ---------------------------------------------------------------------------------------------- */
int zl_embedding(const int32_t* ids, const uint16_t* weight, uint16_t* out, int64_t s_len, int64_t dim,
                 int32_t begin, int32_t end, float scale, int dtype, zl_stream_t s);
/* zl_embedding + zl_rope_cos_sin (llama3 = 0) / zl_rope_cos_sin_llama3 (llama3 = 1) in one launch: the two independent kernels a
 * decode step starts with (bit-identical outputs; d <= 256). */
int zl_embedding_rope(const int32_t* ids, const uint16_t* weight, uint16_t* out, int64_t s_len, int64_t dim, int32_t begin, int32_t end,
                      float scale, int dtype, const int32_t* pos, float* cosv, float* sinv, int64_t d, float base, int neox, int llama3,
                      float factor, float low_freq_factor, float high_freq_factor, float old_context_len, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a8..a11  INT8 (W8A8, dynamic per-token).  Replaces int8_op::quant_calc_scale, layernorm_quant,
 * quant_scale_back, quant_back_act_mul (src/nn/quant/int8/quant_kernel.h:15-128; kernels
 * quant_kernel.cu:15-47, 106-151, 231-246, 589-614) and the cuBLASLt IMMA call of
 * Int8Linear::forward (src/nn/linear/linear.cpp:557-635).  Integer results are bit-exact.
 * ---------------------------------------------------------------------------------------------- */
int zl_quant_calc_scale(const uint16_t* x, int8_t* q, float* scale, int64_t m, int64_t k, int dtype,
                        zl_stream_t s);
int zl_rmsnorm_quant(const uint16_t* x, const uint16_t* weight, uint16_t* out, int8_t* q, float* out_scale,
                     int64_t rows, int64_t dim, float eps, float scale, int dtype, zl_stream_t s);
int zl_int8_gemm_nt(const int8_t* a /* (M,K) */, const int8_t* b /* (N,K) */, int32_t* c /* (M,N) */,
                    int64_t m, int64_t n, int64_t k, zl_stream_t s);
int zl_quant_scale_back(const int32_t* c, const float* scale_x, const uint16_t* scale_y, uint16_t* out,
                        int64_t m, int64_t n, int dtype, zl_stream_t s);
int zl_quant_back_act_mul(const int32_t* a, const float* a_sx, const uint16_t* a_sy, const int32_t* b,
                          const float* b_sx, const uint16_t* b_sy, uint16_t* out, int64_t m, int64_t n,
                          int act, int dtype, zl_stream_t s);
/* W4A8, int8 activations on W4 weights (W4_INT8_ALGO: the M > W4_A8_M_THRES branch of gptq_gemm_k_major,
 * src/nn/quant/gptq/q_gemm_k_major.cu:1036-1073).  Load time, per row of the dequantised matrix W16 (N, K) fp16
 * (zl_w4_dequant): scale[n] = amax / 127 in fp32 (Int4GPTQ::calc_w4a8_scale, linear.cpp:1101-1112), w8 =
 * int8(nearbyintf(float(w) * (1.f / scale))) (KERNEL_dequant<int8_t, 1>, :843-905) -- bit-exact codes.  Forward:
 * zl_quant_calc_scale(a) -> zl_int8_gemm_nt(a_q, w8) -> zl_quant_scale_back_f32 (quant_scale_back with the fp32 weight
 * scale): y = half(float(acc) * sx[m] * sy[n]). */
int zl_w4a8_weight_to_int8(const uint16_t* w16, int8_t* w8, float* scale, int64_t n, int64_t k, zl_stream_t s);
int zl_quant_scale_back_f32(const int32_t* c, const float* scale_x, const float* scale_y, uint16_t* out, int64_t m, int64_t n,
                            zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a7  AWQ checkpoints in their on-disk layout (no repack): qweight (K, N/8) int32, qzeros (K/G, N/8) int32 (zero points
 * as used), scales (K/G, N) fp16; nibble i of a word = column 8c + {0,2,4,6,1,3,5,7}[i].
 * Replaces nn::awq::awq_dequantize / awq_gemm (src/nn/quant/awq/awq.h:10-25; dequantize_weights, gemm_kernels.cu:277-330;
 * gemm_forward_4bit_cuda_m16nXk32 :32-275 + KERNEL_sum_dim0 :381-400; dequantize_s4_to_fp16x2, dequantize.cuh:45-112).
 *   W16[k,n] = rn16(fp16(q - z) * s)                      (exact difference, ONE rounding)
 *   zl_awq_gemm: 32-row K tiles dealt round-robin to split_k_iters splits, fp32 accumulation of x * W16 inside a split,
 *   the split's partial rounded to FP16 (workspace: zl_awq_gemm_workspace_bytes = splits * M * N * 2 bytes), partials
 *   summed in fp32 in split order, one rounding to fp16.  OC % 64 == 0 and group_size % 32 == 0 as in the reference
 *   (:426-433).  Any M (8 rows per weight pass); the reference switches to dequantise + GEMM at M >= 256
 *   (linear.cpp:1560-1565): zl_awq_dequantize + zl_transpose_2d + zl_gemm_nt.
 * ---------------------------------------------------------------------------------------------- */
int zl_awq_dequantize(const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales, uint16_t* out /* (K,N) fp16 */,
                      int64_t k, int64_t n, int64_t group_size, zl_stream_t s);
int64_t zl_awq_gemm_workspace_bytes(int64_t m, int64_t n, int64_t split_k_iters);
int zl_awq_gemm(const uint16_t* x, int64_t ldx, const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales,
                uint16_t* y, void* workspace, int64_t m, int64_t n, int64_t k, int64_t group_size, int64_t split_k_iters,
                zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * a8-a11 fused for decode batches: Int8Linear::forward's GEMM + scale-back in ONE launch (1 <= M <= 32).
 * Replaces functions::Gemm int8 (src/nn/linear/linear.cpp:557-635) followed by quant_scale_back
 * (quant_kernel.cu:231-246), quant_back_element_add_scale (:311-340) or quant_back_act_mul (:589-614);
 * the integer product is exact, the float expression behind it is the reference's, so the result is
 * bit-identical to zl_int8_gemm_nt + zl_quant_scale_back / _back_element_add_scale / _back_act_mul.
 * Weights in the ZLW8M streaming layout (zl_w8m_pack from the (N, K) int8 rows; row_interleave packs
 * [w_in; w_gated] as (gate_n, up_n) row pairs and expects scale_y interleaved the same way):
 *   ZL_W8_BACK      out[m,n]   = T(float(c) * sx[m] * float(sy[n]))
 *   ZL_W8_BACK_ADD  out[m,n]   = T((float(c) * sx[m] * float(sy[n]) + float(addend[m,n])) * scale)
 *   ZL_W8_ACT_SILU / ZL_W8_ACT_GELU   out[m,j] = T(up_j * act(gate_j)),  N/2 columns
 * More rows: ZL_ESHAPE (use zl_int8_gemm_nt + the scale-back launchers).
 * ---------------------------------------------------------------------------------------------- */
enum zl_w8_epilogue { ZL_W8_BACK = 0, ZL_W8_BACK_ADD = 1, ZL_W8_ACT_SILU = 2, ZL_W8_ACT_GELU = 3 };
int64_t zl_w8m_bytes(int64_t n, int64_t k);
int zl_w8m_pack(const int8_t* w /* (N,K) */, int64_t n, int64_t k, int row_interleave, void* qw, zl_stream_t s);
int zl_w8a8_gemm_phase(const int8_t* xq /* (M,K) */, const float* scale_x /* (M) */, const void* qw,
                       const uint16_t* scale_y /* (N) T */, const uint16_t* addend, uint16_t* out, int64_t m, int64_t n,
                       int64_t k, float scale, int epilogue, int dtype, zl_stream_t s);
/* rounds: tiles per workgroup override (1..8; 0 = pick) */
int zl_w8a8_gemm_phase_ex(const int8_t* xq /* (M,K) */, const float* scale_x /* (M) */, const void* qw,
                       const uint16_t* scale_y /* (N) T */, const uint16_t* addend, uint16_t* out, int64_t m, int64_t n,
                       int64_t k, float scale, int epilogue, int dtype, int rounds, zl_stream_t s);
/* The INT8 route's fused qkv projection of a decode step: zl_w8a8_gemm_phase(ZL_W8_BACK) + zl_rope_scatter_decode in one
 * launch (neox, cached cos/sin; bit-identical to the two calls).  1 <= M <= 32, D % 32 == 0. */
int zl_w8a8_qkv_rope_scatter(const int8_t* xq, const float* scale_x, const void* qw, const uint16_t* scale_y,
                             const float* cosv, const float* sinv, const int32_t* placement, const int32_t* buf_lens,
                             uint16_t* const* k_bufs, uint16_t* const* v_bufs, uint16_t* q_out, int64_t m, int64_t h,
                             int64_t hkv, int64_t d, int64_t k, int bshd, int dtype, zl_stream_t s);

/* The other scale-back flavours of the reference, same arithmetic T(float(int32) * sx[row] * sy[col]):
 *   zl_quant_scale_back3             int8_op::quant_scale_back3 (quant_kernel.cu:311-384): fused qkv result -> q | k | v
 *   zl_quant_back_element_add_scale  quant_back_element_add_scale (:530-583): T((back + float(b)) * scale)
 *   zl_quant_back_transpose          quant_back_transpose (:475-527): (B, len_q, H, D) int32 -> (B, H, len_q, D) T
 *   zl_quant_back_copy_to_buffer     quant_back_copy_to_buffer (:389-469): scatter rows into (B, H, len_buf, D)
 *                                    buffers at placement[b, t] (NULL = identity, negative = padded row skipped);
 *                                    strides in elements, 0 for the 3-d (single task) form. */
int zl_quant_scale_back3(const int32_t* c, const float* scale_x, const uint16_t* scale_y, uint16_t* q, uint16_t* k,
                         uint16_t* v, int64_t m, int64_t n, int64_t dim_q, int64_t dim_kv, int dtype, zl_stream_t s);
int zl_quant_back_element_add_scale(const int32_t* a, const float* scale_x, const uint16_t* scale_y, const uint16_t* b,
                                    float scale, uint16_t* out, int64_t m, int64_t n, int dtype, zl_stream_t s);
int zl_quant_back_transpose(const int32_t* inp, const float* scale_x, const uint16_t* scale_y, uint16_t* out,
                            int64_t batch, int64_t len_q, int64_t heads, int64_t dim_head, int dtype, zl_stream_t s);
int zl_quant_back_copy_to_buffer(const int32_t* src, const float* scale_x, const uint16_t* scale_y,
                                 const int32_t* placement, uint16_t* dst, int64_t batch, int64_t len_kv, int64_t heads,
                                 int64_t dim_head, int64_t len_buf, int64_t src_stride, int64_t dst_stride,
                                 int64_t place_stride, int dtype, zl_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Load-time / glue tensor operations behind the bmengine::functions names the reference's layer code calls around its
 * GEMMs (the headers under 3rd/bmengine/bmengine/include/bmengine/functions; hostcpp/bm_functions.h binds them).  Off the decode step's
 * critical path.  Element type codes `zl_elem_t` are bmengine's DataType enumerators (core/dtype.h:12-22).
 *   zl_cast               functions::typecast (typecast.h:7): out[i] = OutT(in[i]) (through fp32; fp32 -> half/bf16 RNE)
 *   zl_copy_2d            rows of width_bytes between pitched buffers: functions::concat_tensor / slice_last_dim /
 *                         copy_last_dim (tensor_ops.h:8-12, index_select.h:33-47)
 *   zl_index_select       functions::index_select (index_select.h:8-14): out[o, j, :] = in[o, index[j], :]
 *   zl_reduce_abs_max     functions::reduce_abs_max (arthmetic.cu:14-45): per row max |x| (fp32 compare, start -1e4)
 *   zl_binary_op          functions::BinaryElementwiseOp (element.cu:44-150): op 0 add 1 sub 2 mul 3 div 4 max;
 *                         bmode 0 same shape, 1 b has one value per row (broadcast over the last dim), 2 b is one row
 *   zl_scale              nn::multiply (src/nn/functions/element.cu:11-31): c = a * T(b)
 *   zl_act_inplace        nn::silu_inplace / gelu_inplace (src/nn/linear/activation_kernel.cu:14-57): act 0 silu, 1 gelu(tanh)
 *   zl_count_nonfinite    functions::check_numeric: adds the number of NaN / Inf elements to *counter (device int32)
 *   zl_perm_narrow_u16    nn::gptq::int32_to_int16 (src/nn/quant/gptq/utils.cu:253-283)
 *   zl_perm_reverse_u16   nn::gptq::reverse_perm (utils.cu:287-319): out[perm[i]] = i
 *   zl_gptq_permute_rows  the make_sequential half of nn::gptq::gptq_shuffle with a q_perm (src/nn/quant/gptq/q_gemm.cu: the rows
 *                         of the (K/8, N) nibble matrix regrouped so that row i is the checkpoint's row perm[i]; perm =
 *                         argsort(g_idx), every value in [0, K))
 *   zl_permute_input_u16  nn::gptq::permute_input (utils.cu:323-395) with the 16-bit permutation the reference keeps
 * ---------------------------------------------------------------------------------------------- */
enum zl_elem_t { ZL_T_F64 = 0, ZL_T_F32 = 1, ZL_T_F16 = 2, ZL_T_I8 = 3, ZL_T_I16 = 4, ZL_T_I32 = 5, ZL_T_BF16 = 6 };
int zl_cast(const void* in, int in_type, void* out, int out_type, int64_t n, zl_stream_t s);
int zl_copy_2d(const void* src, int64_t src_pitch, void* dst, int64_t dst_pitch, int64_t width_bytes, int64_t rows, zl_stream_t s);
int zl_index_select(const void* in, void* out, const int32_t* index, int64_t outer, int64_t dim_in, int64_t n_index,
                    int64_t inner_bytes, zl_stream_t s);
/* the index plumbing of the MoE dispatch route (FeedForward::forward_gpu_dispatch, src/nn/feedforward/feedforward.cpp:599-629,
 * 1040-1075), bmengine's functions::arange (init.h:10), divide on int32 (element.h:25: truncating integer quotient), scatter_update_dim0
 * (scatter.h:7-13: dst[dst_index[i], :] = src[src_index ? src_index[i] : i, :]) and sort_pair_1d (sort.h:8-12; cub's stable radix sort
 * there): a STABLE sort of (int32 key, int32 value) pairs, one workgroup, n <= 2^20, `workspace` 8 n bytes.  max_key > 0: keys lie in
 * [0, max_key] and only the bits that can differ are sorted on; max_key <= 0: the full signed 32-bit order (negative keys first), as
 * cub sorts int32 -- an unfilled -1 expert id lands where the reference puts it. */
int zl_arange_i32(int32_t* out, int32_t start, int32_t step, int64_t n, zl_stream_t s);
int zl_divide_i32(const int32_t* a, int32_t* out, int32_t divisor, int64_t n, zl_stream_t s);
int zl_scatter_update_dim0(void* dst, const int32_t* dst_index, const void* src, const int32_t* src_index, int64_t n_index, int64_t row_bytes,
                           int64_t dst_rows, int64_t src_rows, zl_stream_t s);
int zl_sort_pairs_i32(const int32_t* keys, const int32_t* values, int32_t* keys_out, int32_t* values_out, void* workspace, int64_t n,
                      int32_t max_key, zl_stream_t s);
/* greedy pick over whole logit rows (rows, n; row stride ld elements; type = ZL_T_F16 / ZL_T_BF16 / ZL_T_F32) + the between-steps
 * bookkeeping in one launch: tokens[r] = next_tokens[r] = arg-max of row r (first index of the largest value, NaN largest -- torch.argmax),
 * positions / placement / valid_lens += 1 (any of the five may be null, tokens or next_tokens must be given).  The host-side loop it
 * stands for: fill_search_tokens, src/generator/batch_generator.cpp:1226-1335.  zl_gemm_nt_small_m_argmax + zl_greedy_advance are the
 * form that never writes the pick's logits pass for <= 4 rows. */
int zl_argmax_advance(const void* logits, int type, int64_t rows, int64_t n, int64_t ld, int32_t* tokens, int32_t* positions, int32_t* placement,
                      int32_t* valid_lens, int64_t* next_tokens, zl_stream_t s);
/* The logit post-processing of the reference's batch generator (src/generator/beam_util.cu, 3rd/bmengine/bmengine/functions/{softmax,topk}.cu):
 * what src/generator/batch_generator.cpp calls between a decode step and its host-side search, so that the py_export surface (zhilight.C)
 * links against this boundary.  Rows of n logits of type ZL_T_F16 / ZL_T_BF16 / ZL_T_F32; fp32 arithmetic, one rounding to T.
 *   zl_log_softmax_bias    beam_utility::log_softmax_bias (beam_util.cu:19-128): out = T((x - max) / temperature - log(sum) + bias[row]); temperature 0:
 *                          the form without the division (:44-66); bias may be NULL (zeros)
 *   zl_softmax_rows        functions::softmax (softmax.cu:8-30): out = T(exp(x / t - max / t) / sum)
 *   zl_topk_rows           functions::TopK::forward (topk.cu:280-293): the `top` largest of every row, descending, with int32 positions; ties -> lower index
 *   zl_gather_logits       beam_utility::gather_logits (:130-157): out[i] = float(logits[index[i]]) over the flattened logits
 *   zl_scatter_logits      beam_utility::scatter_update (:224-241): logits[batch_ids[i] * stride + token_ids[i]] = T(values[i]) (add: += in T)
 *   zl_repetition_penalty  beam_utility::beam_repetition_penalty (:199-222): l = presence != 0 ? l - T(presence) : (l < 0 ? l * T(factor) : l / T(factor)) */
int zl_log_softmax_bias(const void* logits, const float* bias, void* out, int64_t rows, int64_t n, float temperature, int type, zl_stream_t s);
int zl_softmax_rows(const void* logits, void* out, int64_t rows, int64_t n, float temperature, int type, zl_stream_t s);
int zl_topk_rows(const void* x, void* out_v, int32_t* out_i, int64_t rows, int64_t n, int top, int type, zl_stream_t s);
int zl_gather_logits(const int32_t* index, const void* logits, float* out, int64_t n, int type, zl_stream_t s);
int zl_scatter_logits(const float* values, const int32_t* token_ids, const int32_t* batch_ids, void* logits, int64_t n, int64_t stride, int add, int type,
                      zl_stream_t s);
int zl_repetition_penalty(const float* factor, const float* presence, const int32_t* tokens, const int32_t* batch_ids, void* logits, int64_t n, int64_t vocab,
                          int type, zl_stream_t s);
int zl_reduce_abs_max(const void* x, void* out, int64_t rows, int64_t cols, int type, zl_stream_t s);
int zl_binary_op(const void* a, const void* b, void* c, int64_t rows, int64_t cols, int op, int bmode, int type, zl_stream_t s);
int zl_scale(const void* in, void* out, int64_t n, float factor, int type, zl_stream_t s);
int zl_act_inplace(void* x, int64_t n, int act, int type, zl_stream_t s);
int zl_count_nonfinite(const void* x, int64_t n, int type, int32_t* counter, zl_stream_t s);
int zl_perm_narrow_u16(const int32_t* perm, uint16_t* out, int64_t k, zl_stream_t s);
int zl_perm_reverse_u16(const int32_t* perm, uint16_t* out, int64_t k, zl_stream_t s);
int zl_gptq_permute_rows(const uint32_t* qweight, uint32_t* out, const int32_t* perm, int64_t k8, int64_t n, zl_stream_t s);
int zl_permute_input_u16(const uint16_t* x, int64_t ldx, const uint16_t* perm, uint16_t* out, int64_t rows, int64_t k, zl_stream_t s);


/* ------------------------------------------------------------------------------------------------
 * f4, first part: the FP8 128x128-block linear and the MoE router.
 *   zl_fp8_per_token_cast   nn::fp8::per_token_cast_to_fp8 (src/nn/quant/fp8/fp8_util.cu:229-323): per (row, 128-column block)
 *                           amax = max|x| clamped at 1e-4, codes = e4m3(float(x) * (MAX / amax)) (fp32, RNE, saturating), scale =
 *                           amax / MAX at [block * aligned_m + row] (scale_col_major, what the GEMM reads) or [row * (n/128) + block].
 *                           Codes and scales bit-exact against the oracle.
 *   zl_fp8_block_dequant    nn::fp8::dequant_fp8_block_weight (:325-385): out = T(float(code) * scale[row/128][col/128]); bit-exact.
 *   zl_fp8_block_gemm_group deep_gemm_fp8_block_h20_group (3rd/deep_gemm/deep_gemm_api.h, the call of Fp8Block::forward /
 *                           grouped_gemm, src/nn/linear/linear.cpp:1863-1945): out[m, n] = T(sum_kb lhs_scales[kb, m] *
 *                           rhs_scales[g][n/128][kb] * sum_{k in kb} lhs[m, k] rhs[g][n, k]), g = m_indices[m] (NULL: one
 *                           matrix; negative: row skipped; groups contiguous and aligned to 16 rows).  fp32 accumulation per
 *                           block on v_mfma_f32_16x16x32_fp8_fp8.  K % 128 == 0.  The reference's kernel is a closed binary:
 *                           parity is against the format's definition in fp64 (oracle/zl_oracle.c: zlo_fp8_block_gemm).
 *   (both routers take logits of dtype ZL_F16 / ZL_BF16 / ZL_F32 -- the reference's router Linear writes fp32 logits, feedforward.cpp:285-286)
 *   zl_moe_top_k_softmax    nn::top_k_softmax (src/nn/feedforward/ff_kernel.cu:174-268); scoring 1 softmax, 2 sigmoid (as written
 *                           there: 1 / (1 + expf(+x))), 3 linear; out_v / out_idx (tokens, top_k_ext), slots >= top_k get weight 1;
 *                           worker_load[id % num_worker] / expert_load[id] are incremented when given.
 *   zl_moe_group_topk       nn::group_topk_softmax (ff_kernel.cu:296-515): DeepSeek-V3's group-limited routing (best groups by
 *                           their best biased score, weights = un-biased scores, renormalised, x weight_scale).
 * ---------------------------------------------------------------------------------------------- */
int zl_fp8_per_token_cast(const uint16_t* x, int64_t ldx, uint8_t* out, int64_t ld_out, float* scale, int64_t aligned_m, int64_t m, int64_t n,
                          int scale_col_major, float max_e4m3, int dtype, zl_stream_t s);
int zl_fp8_block_dequant(const uint8_t* w, const float* scale, uint16_t* out, int64_t rows, int64_t cols, int64_t stride_scale, int dtype,
                         zl_stream_t s);
int zl_fp8_block_gemm_group(const uint8_t* lhs, const float* lhs_scales, int64_t aligned_m, const uint8_t* rhs, const float* rhs_scales,
                            const int32_t* m_indices, uint16_t* out, int64_t m, int64_t n, int64_t k, int num_groups, int dtype, zl_stream_t s);
/* ZLF8M: the (groups, N, K) e4m3 codes of a block-scaled weight re-tiled for streaming -- [group][N / 16][K / 128][2][64 lanes][16 codes],
 * a wavefront's fragment load of one (16-row tile, 128-k block, half) 1 KiB contiguous (what ZLW4M / ZLD16M are for int4 / 16-bit
 * weights); the scales keep their layout.  zl_fp8_block_pack once at load (zl_fp8_block_packed_bytes: size, rows padded to 16 with
 * zero codes); zl_fp8_block_gemm_group_packed: zl_fp8_block_gemm_group's results bit for bit, for the decode shapes (m <= 32 per launch,
 * or the grouped form). */
int64_t zl_fp8_block_packed_bytes(int64_t n, int64_t k, int64_t num_groups);
int zl_fp8_block_pack(const uint8_t* w, uint8_t* out, int64_t n, int64_t k, int64_t num_groups, zl_stream_t s);
int zl_fp8_block_gemm_group_packed(const uint8_t* lhs, const float* lhs_scales, int64_t aligned_m, const uint8_t* rhs_packed, const float* rhs_scales,
                                   const int32_t* m_indices, uint16_t* out, int64_t m, int64_t n, int64_t k, int num_groups, int dtype,
                                   zl_stream_t s);
int zl_moe_top_k_softmax(const uint16_t* logits, int64_t tokens, int num_exp, int top_k, int top_k_ext, int renormalize, float weight_scale,
                         int scoring, int dtype, float* out_v, int32_t* out_idx, int32_t* worker_load, int32_t* expert_load, int num_worker,
                         zl_stream_t s);
int zl_moe_group_topk(const uint16_t* logits, const float* correction_bias, int64_t tokens, int num_exp, int top_k, int top_k_ext,
                      int renormalize, float weight_scale, int scoring, int num_group, int topk_group, int dtype, float* out_v,
                      int32_t* out_idx, int32_t* worker_load, int32_t* expert_load, int num_worker, zl_stream_t s);


/* f4: the prompt-side MoE dispatch / combine around the grouped GEMMs (src/nn/feedforward/ff_kernel.h:42-96, ff_kernel.cu:518-1082).
 * Integer outputs exact; the weighted sums accumulate in fp32 in slot order, one fma per term, one rounding to T.
 *   zl_moe_sum_experts       nn::sum_experts (one concatenated input): out[q] = sum_i input[index[q K + i]] * weight[q K + i]
 *   zl_moe_sum_experts_arr   nn::sum_experts (one input per expert; DEVICE array of row pointers): a single token reads row 0 of each
 *                            expert and weight[k]; exp_parallel keeps the experts with (id & (world_size - 1)) == local_rank
 *   zl_moe_route_shared_lb   nn::route_shared_lb: shared-expert slot s of token q -> the first rank with spare capacity
 *                            (max_load - worker_load_base[r]), pseudo expert id (num_local_experts + s) * world_size + rank
 *   zl_moe_plus_for_sort     nn::plus_for_sort: id + (id % world_size) * multiple (sort key that groups experts by rank)
 *   zl_moe_calc_reverse_idx  nn::calc_reverse_idx's kernel: rev[indices[i]] = i - expert_offsets[exp_ids[indices[i]]]
 *   zl_moe_fill_m_indices    nn::fill_m_indices_padded_indices' kernel: per local expert e with num_tokens[e] rows at offsets[e]:
 *                            padded_indices[offsets[e] + s] = aligned_offsets[e] + s; m_indices[aligned_offsets[e] ..
 *                            aligned_offsets[e + 1]) = e (every run padded to block_m rows: the layout zl_fp8_block_gemm_group reads)
 * The prefix sums those last two take are the host loops of the reference (ops.py / nn_amd.cpp restate them). */
int zl_moe_sum_experts(const uint16_t* input, const int32_t* index, const float* weight, uint16_t* out, int64_t seq_len, int top_k,
                       int64_t dim_model, int dtype, zl_stream_t s);
int zl_moe_sum_experts_arr(const uint16_t* const* input_arr, const int32_t* experts, const int32_t* index, const float* weight, uint16_t* out,
                           int64_t seq_len, int top_k, int64_t dim_model, int exp_parallel, int world_size, int local_rank, int dtype,
                           zl_stream_t s);
int zl_moe_route_shared_lb(int32_t* exp_ids, const int32_t* worker_load_base, int32_t* worker_load, int32_t* expert_load, int max_load,
                           int world_size, int64_t seq_len, int top_k, int top_k_ext, int num_local_experts, zl_stream_t s);
int zl_moe_plus_for_sort(const int32_t* exp_ids, int32_t* out, int multiple, int world_size, int64_t numel, zl_stream_t s);
int zl_moe_calc_reverse_idx(const int32_t* exp_ids, const int32_t* indices, const int32_t* expert_offsets, int32_t* rev_indices, int64_t numel,
                            zl_stream_t s);
int zl_moe_fill_m_indices(const int32_t* num_tokens, const int32_t* offsets, const int32_t* aligned_offsets, int32_t* padded_indices,
                          int32_t* m_indices, int local_experts, int max_num_token, int block_m, zl_stream_t s);


/* f4: multi-head latent attention (DeepSeek MLA) for decode rows over the compressed cache.  Replaces MLAImpl's attention over the latent
 * cache (src/nn/attention/multi_head_latent_attention.cpp:836-872 gemm + attn_softmax + gemm; :877-1004 FlashMLA, closed): every head of
 * task b attends to the same rows kv_bufs[b] (len_buf, kv_lora_rank + rope_dim): key = the whole row, value = its first kv_lora_rank values;
 * q_adj (B, H, kv_lora_rank + rope_dim) is the absorbed query (q_nope . W_UK | q_rope); out (B, H, kv_lora_rank); keys 0 ..
 * min(buf_lens[b], valid_lens[b]) - 1 are visible (valid_lens nullable).  fp32 scores / probabilities / accumulation, one rounding to T.
 * kv_lora_rank = 512, rope_dim = 64, H % 4 == 0.  workspace: zl_mla_decode_workspace_bytes(b, h, max_len_buf) bytes
 * (csrc/mla_attn.hip). */
int64_t zl_mla_decode_workspace_bytes(int64_t b, int64_t h, int64_t max_len_buf);
int zl_mla_decode_attn(const uint16_t* q_adj, const int32_t* buf_lens, const int32_t* valid_lens, const uint16_t* const* kv_bufs, uint16_t* out,
                       void* workspace, int64_t b, int64_t h, int64_t kv_lora_rank, int64_t rope_dim, float scale, int64_t max_len_buf, int dtype,
                       zl_stream_t s);

/* the same with the kernel choice spelled out: algo 0 = the matrix-core kernel (k_mla_decode_mfma: S^T = KV . Q^T on v_mfma_f32_16x16x32,
 * O^T += V^T . P^T on v_mfma_f32_16x16x16 with hi + lo probability halves; what zl_mla_decode_attn runs), 1 = the VALU kernel of the
 * first version (kept for the tests that compare the two). */
int zl_mla_decode_attn_ex(const uint16_t* q_adj, const int32_t* buf_lens, const int32_t* valid_lens, const uint16_t* const* kv_bufs, uint16_t* out,
                          void* workspace, int64_t b, int64_t h, int64_t kv_lora_rank, int64_t rope_dim, float scale, int64_t max_len_buf, int dtype,
                          int algo, zl_stream_t s);

/* The same attention over a PAGED latent cache: what ds::mha_fwd_kvcache_mla (src/nn/attention/ds_flash_mla_api.h:16-30, .cpp:67-197: the
 * binding of the closed FlashMLA library) computes for its non-causal calls (multi_head_latent_attention.cpp:913, :993 pass is_causal =
 * false).  kcache (num_blocks, page_block_size, 1, kv_lora_rank + rope_dim); page i of task b is block_table[b * max_blocks_per_seq + i];
 * keys 0 .. seqlens_k[b] - 1 are visible.  q_adj (B, H, 576) where H = len_q * num_heads (the reference folds the query rows into the head
 * axis the same way, .cpp:121-126); out (B, H, 512); softmax_lse (B, H) fp32 = log sum exp(scale * score), nullable.  page_block_size a
 * multiple of 64 (FlashMLA fixes 64).  workspace: zl_mla_decode_workspace_bytes(b, h, page_block_size * max_blocks_per_seq). */
int zl_mla_decode_attn_paged(const uint16_t* q_adj, const uint16_t* kcache, const int32_t* block_table, const int32_t* seqlens_k, uint16_t* out,
                             float* softmax_lse, void* workspace, int64_t b, int64_t h, int64_t kv_lora_rank, int64_t rope_dim,
                             int64_t page_block_size, int64_t max_blocks_per_seq, float scale, int dtype, zl_stream_t s);


#ifdef __cplusplus
}
#endif
#endif /* ZHILIGHT_AMD_H */

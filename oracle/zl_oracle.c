/*
 * zl_oracle.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (see zl_oracle.h header comment).
 *
 * Plain C99 restatement of the reference's hot-path arithmetic.  Every function cites the
 * reference file:line (relative to /root/reference) it follows.  No code is copied: the CUDA
 * kernels are re-expressed as scalar loops that make the same roundings in the same order.
 *
 * Conventions used to mirror what the CUDA binary computes under IEEE-754:
 *   - `a += b * c` on floats in device code is contracted by nvcc into one fp32 FMA  -> fmaf().
 *   - __hfma2 / __hadd2 / __hmul2 round ONCE to fp16 per lane                         -> h_fma() etc.
 *   - warp shuffle-down trees (offsets 16,8,4,2,1) are replayed pairwise in that order.
 * Build with -ffp-contract=off so that the C compiler adds no contractions of its own.
 */
#include "zl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* soft-float conversions                                                                      */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float zlo_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        float v = (float)man * 5.9604644775390625e-08f; /* man * 2^-24, exact */
        return sign ? -v : v;
    }
    if (exp == 31) return bits_f32(sign | 0x7f800000u | (man << 13));
    return bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

/* round-to-nearest-even of a double to a format with `mbits` explicit mantissa bits, minimum
 * normal exponent `emin`; returns (n, e) such that value = n * 2^(e - mbits), n in [0, 2^(mbits+1)] */
uint16_t zlo_f64_to_f16(double d) {
    uint16_t sign = signbit(d) ? 0x8000u : 0;
    double a = fabs(d);
    if (isnan(d)) return (uint16_t)(sign | 0x7e00u);
    if (a >= 65520.0) return (uint16_t)(sign | 0x7c00u);      /* overflow (or inf) -> inf */
    if (a < 2.9802322387695312e-08) return sign;               /* < 2^-25 -> 0; tie 2^-25 -> even 0 below */
    int e;
    (void)frexp(a, &e);                                        /* a = f * 2^e, f in [0.5,1) */
    e -= 1;                                                    /* a in [2^e, 2^(e+1)) */
    if (e < -14) {                                             /* subnormal: quantum 2^-24 */
        double n = nearbyint(ldexp(a, 24));                    /* RNE (default rounding mode) */
        return (uint16_t)(sign | (uint16_t)n);                 /* n == 1024 -> smallest normal */
    }
    double n = nearbyint(ldexp(a, 10 - e));                    /* in [1024, 2048] */
    if (n >= 2048.0) { n = 1024.0; e += 1; }
    return (uint16_t)(sign | (uint16_t)((e + 15) << 10) | ((uint16_t)n - 1024u));
}

uint16_t zlo_f32_to_f16(float f) { return zlo_f64_to_f16((double)f); }

float zlo_bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

uint16_t zlo_f32_to_bf16(float f) {
    uint32_t u = f32_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u); /* NaN */
    uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (uint16_t)(u >> 16);
}

static uint16_t f64_to_bf16(double d) {
    /* exact single rounding double -> bf16 (8-bit significand, emin -126) */
    uint16_t sign = signbit(d) ? 0x8000u : 0;
    double a = fabs(d);
    if (isnan(d)) return (uint16_t)(sign | 0x7fc0u);
    if (a >= 3.3961775292304601e38) return (uint16_t)(sign | 0x7f80u); /* (2-2^-8)*2^127 */
    if (a == 0.0) return sign;
    int e;
    (void)frexp(a, &e);
    e -= 1;
    if (e < -126) {
        double n = nearbyint(ldexp(a, 133));                   /* quantum 2^-133 */
        return (uint16_t)(sign | (uint16_t)n);
    }
    double n = nearbyint(ldexp(a, 7 - e));                     /* in [128, 256] */
    if (n >= 256.0) { n = 128.0; e += 1; }
    return (uint16_t)(sign | (uint16_t)((e + 127) << 7) | ((uint16_t)n - 128u));
}

void zlo_f32_to_f16_array(const float* in, uint16_t* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = zlo_f32_to_f16(in[i]);
}
void zlo_f16_to_f32_array(const uint16_t* in, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = zlo_f16_to_f32(in[i]);
}

static inline float T2f(uint16_t v, int dt) { return dt ? zlo_bf16_to_f32(v) : zlo_f16_to_f32(v); }
static inline uint16_t f2T(float f, int dt) { return dt ? zlo_f32_to_bf16(f) : zlo_f32_to_f16(f); }
static inline uint16_t d2T(double d, int dt) { return dt ? f64_to_bf16(d) : zlo_f64_to_f16(d); }

/* fp16 fused multiply-add with a single rounding (what __hfma2 does per lane).  a*b is exact in
 * double (22 significant bits); a*b+c in double is either exact or far from any fp16 tie. */
static inline uint16_t h_fma(uint16_t a, uint16_t b, uint16_t c) {
    return zlo_f64_to_f16((double)zlo_f16_to_f32(a) * (double)zlo_f16_to_f32(b) + (double)zlo_f16_to_f32(c));
}
static inline uint16_t h_mul(uint16_t a, uint16_t b) {
    return zlo_f64_to_f16((double)zlo_f16_to_f32(a) * (double)zlo_f16_to_f32(b));
}

/* shuffle-down tree of a 32-lane warp, result in lane 0 (bm/include/bmengine/functions/reduce.cuh:41-47) */
static inline float warp32_tree_sum(float* x) {
    for (int off = 16; off > 0; off >>= 1)
        for (int l = 0; l < off; ++l) x[l] = x[l] + x[l + off];
    return x[0];
}

/* block reduce (reduce.cuh:92-107): per-warp tree, then warp 0 trees the per-warp results */
static float block_tree_sum(const float* per_thread, int threads) {
    float warp_res[32];
    int nwarp = threads / 32;
    for (int w = 0; w < 32; ++w) warp_res[w] = 0.f;
    for (int w = 0; w < nwarp; ++w) {
        float lane[32];
        for (int l = 0; l < 32; ++l) lane[l] = per_thread[w * 32 + l];
        warp_res[w] = warp32_tree_sum(lane);
    }
    return warp32_tree_sum(warp_res);
}

static inline int round_up_i(int x, int m) { return (x + m - 1) / m * m; }

/* ------------------------------------------------------------------------------------------ */
/* a4: load-time layout transforms                                                             */
/* ------------------------------------------------------------------------------------------ */

/* src/nn/quant/gptq/qdq_4.cuh:16-35 (shuffle_4bit_8) applied to every word of the (K/8, N) matrix
 * (q_gemm.cu:778-791 shuffle_kernel): even weights go to the low 16 bits, odd ones to the high 16. */
void zlo_gptq_shuffle(uint32_t* qweight, int64_t k8, int64_t n) {
    for (int64_t i = 0; i < k8 * n; ++i) {
        uint32_t qa = qweight[i], qb = 0;
        for (int j = 0; j < 4; ++j) {
            uint32_t even = qa & 0xfu, odd = (qa >> 4) & 0xfu;
            qa >>= 8;
            qb |= even << (4 * j);
            qb |= odd << (4 * j + 16);
        }
        qweight[i] = qb;
    }
}

/* src/nn/quant/gptq/utils.cu:61-88: +1 on every nibble with 0xF wrapping to 0 (no carry) */
void zlo_gptq_increase_zero(uint32_t* qzeros, int64_t nwords) {
    for (int64_t i = 0; i < nwords; ++i) {
        uint32_t q = qzeros[i], r = 0;
        for (int j = 0; j < 8; ++j) {
            uint32_t nib = (q >> (4 * j)) & 0xfu;
            nib = (nib == 0xfu) ? 0u : nib + 1u;
            r |= nib << (4 * j);
        }
        qzeros[i] = r;
    }
}

/* src/nn/quant/gptq/utils.cu:177-214: nibble j of word i -> byte 8*i + j */
void zlo_gptq_q4_to_q8(const uint32_t* in, uint8_t* out, int64_t nwords) {
    for (int64_t i = 0; i < nwords; ++i)
        for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)((in[i] >> (4 * j)) & 0xfu);
}

#define DEF_TRANSPOSE(NAME, TYPE)                                                          \
    void NAME(const TYPE* in, TYPE* out, int64_t rows, int64_t cols) {                     \
        for (int64_t r = 0; r < rows; ++r)                                                 \
            for (int64_t c = 0; c < cols; ++c) out[c * rows + r] = in[r * cols + c];       \
    }
DEF_TRANSPOSE(zlo_transpose_u32, uint32_t)
DEF_TRANSPOSE(zlo_transpose_u16, uint16_t)
DEF_TRANSPOSE(zlo_transpose_u8, uint8_t)

/* src/nn/quant/gptq/utils.cu:25-58: AWQ nibble order [0,4,1,5,2,6,3,7] -> natural, in place */
void zlo_awq_un_shuffle(uint32_t* q, int64_t dim0, int64_t n) {
    static const int de[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    for (int64_t i = 0; i < dim0 * n; ++i) {
        uint32_t in = q[i], out = 0;
        for (int s = 0; s < 8; ++s) out |= ((in >> (de[s] * 4)) & 0xfu) << (s * 4);
        q[i] = out;
    }
}

/* src/nn/quant/gptq/utils.cu:121-174: AWQ (K, N/8) -> GPTQ-style (K/8, N); with use_exllama the 8
 * k-nibbles of an output word are stored in exllama order (even k low half, odd k high half) */
void zlo_awq_shuffle(const uint32_t* in, uint32_t* out, int64_t k, int64_t n, int use_exllama) {
    static const int de[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    static const int sfl[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    int64_t n8 = n / 8;
    for (int64_t kb = 0; kb < k / 8; ++kb)
        for (int64_t nb = 0; nb < n8; ++nb) {
            uint32_t dq[8][8];
            for (int r = 0; r < 8; ++r) {
                uint32_t q = in[(kb * 8 + r) * n8 + nb];
                for (int s = 0; s < 8; ++s) dq[r][s] = (q >> (de[s] * 4)) & 0xfu;
            }
            for (int c = 0; c < 8; ++c) {
                uint32_t q = 0;
                for (int s = 0; s < 8; ++s) q += dq[use_exllama ? sfl[s] : s][c] << (s * 4);
                out[kb * n + nb * 8 + c] = q;
            }
        }
}

/* Int4GPTQ::preprocess_weight + transpose_weight (src/nn/linear/linear.cpp:1139-1160, 1085-1099),
 * no act-order: shuffle qweight, +1 zeros, zeros nibble->byte, transpose all three to k-major. */
void zlo_gptq_prepare_k_major(const uint32_t* qweight_hf, const uint32_t* qzeros_hf,
                              const uint16_t* scales_hf, int64_t k, int64_t n, int64_t g,
                              uint32_t* qw_km, uint8_t* qz_km, uint16_t* sc_km) {
    int64_t k8 = k / 8, ng = k / g;
    uint32_t* qw = (uint32_t*)malloc(sizeof(uint32_t) * k8 * n);
    memcpy(qw, qweight_hf, sizeof(uint32_t) * k8 * n);
    zlo_gptq_shuffle(qw, k8, n);
    zlo_transpose_u32(qw, qw_km, k8, n);
    free(qw);
    uint32_t* qz = (uint32_t*)malloc(sizeof(uint32_t) * ng * (n / 8));
    memcpy(qz, qzeros_hf, sizeof(uint32_t) * ng * (n / 8));
    zlo_gptq_increase_zero(qz, ng * (n / 8));
    uint8_t* qz8 = (uint8_t*)malloc((size_t)ng * n);
    zlo_gptq_q4_to_q8(qz, qz8, ng * (n / 8));
    zlo_transpose_u8(qz8, qz_km, ng, n);
    free(qz);
    free(qz8);
    zlo_transpose_u16(scales_hf, sc_km, ng, n);
}

/* AutoGPTQ v1 on-disk format definition: W[k,n] = (q - (zstored + 1)) * s with
 * the +1 wrapping 15 -> 0 as the reference's increase_zero does.  Output (N, K) in fp64. */
void zlo_gptq_dequant_hf_naive(const uint32_t* qweight_hf, const uint32_t* qzeros_hf,
                               const uint16_t* scales_hf, const int32_t* g_idx,
                               int64_t k, int64_t n, int64_t g, double* w_nk) {
    int64_t n8 = n / 8;
    for (int64_t kk = 0; kk < k; ++kk) {
        int64_t grp = g_idx ? g_idx[kk] : kk / g;
        for (int64_t nn = 0; nn < n; ++nn) {
            int q = (int)((qweight_hf[(kk / 8) * n + nn] >> (4 * (kk % 8))) & 0xfu);
            int zs = (int)((qzeros_hf[grp * n8 + nn / 8] >> (4 * (nn % 8))) & 0xfu);
            int z = (zs == 15) ? 0 : zs + 1;
            w_nk[nn * k + kk] = (double)(q - z) * (double)zlo_f16_to_f32(scales_hf[grp * n + nn]);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a2: W4A16 k-major GEMM, R flavour                                                           */
/* ------------------------------------------------------------------------------------------ */

/* dequant_8x4bit (q_gemm_k_major.cu:74-99): word -> 8 fp16 values (q - z), exact.  Natural weight
 * order j = 0..7; weight 2p sits in nibble p, weight 2p+1 in nibble p+4 (exllama-shuffled word). */
static inline void dequant_word(uint32_t w, int zero, uint16_t d[8]) {
    for (int p = 0; p < 4; ++p) {
        int even = (int)((w >> (4 * p)) & 0xfu), odd = (int)((w >> (4 * p + 16)) & 0xfu);
        d[2 * p] = zlo_f64_to_f16((double)(even - zero));
        d[2 * p + 1] = zlo_f64_to_f16((double)(odd - zero));
    }
}

/* dot_8_half (q_gemm_k_major.cu:101-108): two interleaved fp16 accumulators (even k, odd k), four
 * hfma2 steps, then f32(lo) + f32(hi) */
static inline float dot8_half(const uint16_t d[8], const uint16_t* a) {
    uint16_t lo = 0, hi = 0;
    for (int i = 0; i < 4; ++i) {
        lo = h_fma(d[2 * i], a[2 * i], lo);
        hi = h_fma(d[2 * i + 1], a[2 * i + 1], hi);
    }
    return zlo_f16_to_f32(lo) + zlo_f16_to_f32(hi);
}

/* DEV_gemm_warp_reduce + warpReduceSum (q_gemm_k_major.cu:127-174, 217-220): lane l of a 32-lane
 * warp walks words l, l+32, ... with acc = fma(dot8, scale, acc); lanes are then tree-summed. */
static float gemv_row_R(const uint16_t* x, const uint32_t* qw_row, const uint8_t* qz_row,
                        const uint16_t* sc_row, int64_t k8, int64_t g8, int sym) {
    float lane_acc[32];
    for (int l = 0; l < 32; ++l) {
        float acc = 0.f;
        for (int64_t w = l; w < k8; w += 32) {
            int64_t gi = w / g8;
            float scale = zlo_f16_to_f32(sc_row[gi]);
            int zero = sym ? 8 : (int)qz_row[gi];
            uint16_t d[8];
            dequant_word(qw_row[w], zero, d);
            acc = fmaf(dot8_half(d, x + 8 * w), scale, acc);
        }
        lane_acc[l] = acc;
    }
    return warp32_tree_sum(lane_acc);
}

/* the 32 per-lane partial sums of DEV_gemm_warp_reduce<1> (what gemv_row_R feeds its tree with) */
static void gemv_row_lanes(const uint16_t* x, const uint32_t* qw_row, const uint8_t* qz_row, const uint16_t* sc_row, int64_t k8,
                           int64_t g8, int sym, float* lane_acc) {
    for (int l = 0; l < 32; ++l) {
        float acc = 0.f;
        for (int64_t w = l; w < k8; w += 32) {
            int64_t gi = w / g8;
            float scale = zlo_f16_to_f32(sc_row[gi]);
            int zero = sym ? 8 : (int)qz_row[gi];
            uint16_t d[8];
            dequant_word(qw_row[w], zero, d);
            acc = fmaf(dot8_half(d, x + 8 * w), scale, acc);
        }
        lane_acc[l] = acc;
    }
}

/* the token's expert t: id in the stacked arrays, or -1 when skipped (KERNEL_gemm_moe_up :276-288, _down :349-360) */
static int moe_expert(const int32_t* ids, int64_t m, int t, int top_k, int shared_base, int exp_parallel, int world, int rank) {
    if (t >= top_k) return shared_base + t - top_k;
    int e = ids[m * top_k + t];
    if (exp_parallel) {
        if ((e % world) != rank) return -1;
        e /= world;
    }
    return e;
}

/* KERNEL_gemm_moe_up (q_gemm_k_major.cu:243-320): C[m, t, n] = half(silu(acc1) * acc2), silu in double (:239-241), C zero-filled
 * for skipped experts (:423).  qw1 / qw2 etc.: (E, N, ...) stacks of k-major tensors (gate and up) */
void zlo_gptq_moe_up(const uint16_t* x, const uint32_t* qw1, const uint8_t* qz1, const uint16_t* sc1, const uint32_t* qw2,
                     const uint8_t* qz2, const uint16_t* sc2, const int32_t* ids, uint16_t* out, int64_t m, int64_t n, int64_t k,
                     int64_t g, int sym, int top_k, int n_shared, int shared_base, int exp_parallel, int world, int rank) {
    int64_t k8 = k / 8, g8 = g / 8, ng = k / g;
    int T = top_k + n_shared;
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t mm = 0; mm < m; ++mm)
        for (int t = 0; t < T; ++t) {
            int e = moe_expert(ids, mm, t, top_k, shared_base, exp_parallel, world, rank);
            for (int64_t nn = 0; nn < n; ++nn) {
                uint16_t* o = out + (mm * T + t) * n + nn;
                if (e < 0) {
                    *o = 0;
                    continue;
                }
                int64_t row = (int64_t)e * n + nn;
                float a1 = gemv_row_R(x + mm * k, qw1 + row * k8, qz1 + row * ng, sc1 + row * ng, k8, g8, sym);
                float a2 = gemv_row_R(x + mm * k, qw2 + row * k8, qz2 + row * ng, sc2 + row * ng, k8, g8, sym);
                float s = (float)((double)a1 / (1.0 + (double)expf(-a1)));
                *o = zlo_f32_to_f16(s * a2);
            }
        }
}

/* KERNEL_gemm_moe_down (q_gemm_k_major.cu:322-390): per lane acc_all += acc_th * weight over the token's experts (the product
 * and the sum contract to one fma under nvcc's default -fmad), then the 32-lane tree; C = half(acc) or half(float(C) + acc) */
void zlo_gptq_moe_down(const uint16_t* a, const uint32_t* qw, const uint8_t* qz, const uint16_t* sc, const int32_t* ids,
                       const float* weights, uint16_t* out, int64_t m, int64_t n, int64_t k, int64_t g, int sym, int top_k,
                       int n_shared, int shared_base, int exp_parallel, int world, int rank, int add_c) {
    int64_t k8 = k / 8, g8 = g / 8, ng = k / g;
    int T = top_k + n_shared;
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t mm = 0; mm < m; ++mm)
        for (int64_t nn = 0; nn < n; ++nn) {
            float all[32], part[32];
            for (int l = 0; l < 32; ++l) all[l] = 0.f;
            for (int t = 0; t < T; ++t) {
                int e = moe_expert(ids, mm, t, top_k, shared_base, exp_parallel, world, rank);
                if (e < 0) continue;
                int64_t row = (int64_t)e * n + nn;
                gemv_row_lanes(a + (mm * T + t) * k, qw + row * k8, qz + row * ng, sc + row * ng, k8, g8, sym, part);
                float w = t < top_k ? weights[mm * top_k + t] : 1.f;
                for (int l = 0; l < 32; ++l) all[l] = fmaf(part[l], w, all[l]);
            }
            float acc = warp32_tree_sum(all);
            uint16_t* o = out + mm * n + nn;
            *o = zlo_f32_to_f16(add_c ? zlo_f16_to_f32(*o) + acc : acc);
        }
}

/* KERNEL_gemm_warp_reduce (q_gemm_k_major.cu:176-237): C = half(alpha*acc + bias), alpha = 1;
 * ADD_C: half(float(C) + acc + bias).  Per-row arithmetic is identical for every WRAP_M, so any M
 * (the reference routes M <= 40 here, :1101-1115) is M independent row-vectors. */
void zlo_gptq_gemm_k_major(const uint16_t* x, const uint32_t* qw, const uint8_t* qz,
                           const uint16_t* sc, const uint16_t* bias, uint16_t* y,
                           int64_t m, int64_t n, int64_t k, int64_t g, int sym, int add_c) {
    int64_t k8 = k / 8, g8 = g / 8, ng = k / g;
#pragma omp parallel for schedule(static)
    for (int64_t nn = 0; nn < n; ++nn)
        for (int64_t mm = 0; mm < m; ++mm) {
            float acc = gemv_row_R(x + mm * k, qw + nn * k8, qz + nn * ng, sc + nn * ng, k8, g8, sym);
            float b = bias ? zlo_f16_to_f32(bias[nn]) : 0.f;
            float r;
            if (add_c) r = (zlo_f16_to_f32(y[mm * n + nn]) + acc) + b;
            else r = acc + b;
            y[mm * n + nn] = zlo_f32_to_f16(r);
        }
}

void zlo_gptq_gemm_k_major_exact(const uint16_t* x, const uint32_t* qw, const uint8_t* qz,
                                 const uint16_t* sc, const uint16_t* bias, double* y,
                                 int64_t m, int64_t n, int64_t k, int64_t g, int sym) {
    int64_t k8 = k / 8, g8 = g / 8, ng = k / g;
#pragma omp parallel for schedule(static)
    for (int64_t nn = 0; nn < n; ++nn)
        for (int64_t mm = 0; mm < m; ++mm) {
            double acc = 0.0;
            for (int64_t w = 0; w < k8; ++w) {
                int64_t gi = w / g8;
                double s = (double)zlo_f16_to_f32(sc[nn * ng + gi]);
                int zero = sym ? 8 : (int)qz[nn * ng + gi];
                uint32_t word = qw[nn * k8 + w];
                double part = 0.0;
                for (int p = 0; p < 4; ++p) {
                    int even = (int)((word >> (4 * p)) & 0xfu), odd = (int)((word >> (4 * p + 16)) & 0xfu);
                    part += (double)(even - zero) * (double)zlo_f16_to_f32(x[mm * k + 8 * w + 2 * p]);
                    part += (double)(odd - zero) * (double)zlo_f16_to_f32(x[mm * k + 8 * w + 2 * p + 1]);
                }
                acc += part * s;
            }
            if (bias) acc += (double)zlo_f16_to_f32(bias[nn]);
            y[mm * n + nn] = acc;
        }
}

/* KERNEL_dequant<half,0> (q_gemm_k_major.cu:843-886): W16[n,k] = rn16(rn16(q - z) * s) */
void zlo_gptq_dequant_k_major(const uint32_t* qw, const uint8_t* qz, const uint16_t* sc,
                              uint16_t* out, int64_t n, int64_t k, int64_t g) {
    int64_t k8 = k / 8, g8 = g / 8, ng = k / g;
#pragma omp parallel for schedule(static)
    for (int64_t nn = 0; nn < n; ++nn)
        for (int64_t w = 0; w < k8; ++w) {
            uint16_t d[8];
            dequant_word(qw[nn * k8 + w], (int)qz[nn * ng + w / g8], d);
            uint16_t s = sc[nn * ng + w / g8];
            for (int j = 0; j < 8; ++j) out[nn * k + 8 * w + j] = h_mul(d[j], s);
        }
}

/* KERNEL_gemm_fuse_gate_in (q_gemm_k_major.cu:529-578): out = half(silu(acc1) * acc2) with the
 * file-local silu(x) = x / (1.0 + expf(-x)) evaluated in double (:239-241), product in fp32 */
void zlo_gptq_gemm_fuse_gate_in(const uint16_t* x,
                                const uint32_t* qw1, const uint8_t* qz1, const uint16_t* sc1,
                                const uint32_t* qw2, const uint8_t* qz2, const uint16_t* sc2,
                                uint16_t* y, int64_t m, int64_t n, int64_t k, int64_t g, int sym) {
    int64_t k8 = k / 8, g8 = g / 8, ng = k / g;
#pragma omp parallel for schedule(static)
    for (int64_t nn = 0; nn < n; ++nn)
        for (int64_t mm = 0; mm < m; ++mm) {
            float a1 = gemv_row_R(x + mm * k, qw1 + nn * k8, qz1 + nn * ng, sc1 + nn * ng, k8, g8, sym);
            float a2 = gemv_row_R(x + mm * k, qw2 + nn * k8, qz2 + nn * ng, sc2 + nn * ng, k8, g8, sym);
            float sl = (float)((double)a1 / (1.0 + (double)expf(-a1)));
            y[mm * n + nn] = zlo_f32_to_f16(sl * a2);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a17: RMSNorm                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* KERNEL_layernorm_rms (src/nn/layernorm/layernorm.cu:10-42), launch shape :92-93:
 * threads = min(round_up(dim,32),1024); per-thread strided fma chain of v*v, block tree, /dim,
 * rsqrt(.+eps) (oracle: 1/sqrtf, the GPU instruction is a <=2ulp approximation),
 * y = T(((v*r)*w)/scale); fused add: v = f32(x)+f32(x2), out_sum = T(v), norm uses the fp32 v. */
void zlo_rmsnorm(const uint16_t* x, const uint16_t* w, uint16_t* out, int64_t rows, int64_t dim,
                 float eps, float scale, const uint16_t* x2, uint16_t* out_sum, int dtype) {
    int threads = round_up_i((int)dim, 32);
    if (threads > 1024) threads = 1024;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float* v = (float*)malloc(sizeof(float) * dim);
        float part[1024];
        for (int t = 0; t < threads; ++t) {
            float acc = 0.f;
            for (int64_t i = t; i < dim; i += threads) {
                float val = T2f(x[r * dim + i], dtype);
                if (x2) {
                    val += T2f(x2[r * dim + i], dtype);
                    if (out_sum) out_sum[r * dim + i] = f2T(val, dtype);
                }
                v[i] = val;
                acc = fmaf(val, val, acc);
            }
            part[t] = acc;
        }
        float ms = block_tree_sum(part, threads) / (float)dim;
        float rs = 1.0f / sqrtf(ms + eps);
        for (int64_t i = 0; i < dim; ++i)
            out[r * dim + i] = f2T(v[i] * rs * T2f(w[i], dtype) / scale, dtype);
        free(v);
    }
}

void zlo_rmsnorm_exact(const uint16_t* x, const uint16_t* w, double* out, int64_t rows, int64_t dim,
                       float eps, float scale, const uint16_t* x2, int dtype) {
    for (int64_t r = 0; r < rows; ++r) {
        double ss = 0.0;
        for (int64_t i = 0; i < dim; ++i) {
            double v = (double)T2f(x[r * dim + i], dtype) + (x2 ? (double)T2f(x2[r * dim + i], dtype) : 0.0);
            ss += v * v;
        }
        double rs = 1.0 / sqrt(ss / (double)dim + (double)eps);
        for (int64_t i = 0; i < dim; ++i) {
            double v = (double)T2f(x[r * dim + i], dtype) + (x2 ? (double)T2f(x2[r * dim + i], dtype) : 0.0);
            out[r * dim + i] = v * rs * (double)T2f(w[i], dtype) / (double)scale;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a13: RoPE                                                                                   */
/* ------------------------------------------------------------------------------------------ */

static inline int half_dim_index(int col, int half_dim, int neox) { /* rope_common.cuh:3-12 */
    return neox ? (col < half_dim ? col : col - half_dim) : col / 2;
}

/* KERNEL_rope_cos_sin (src/nn/position/rope_preparer.cu:49-69) */
void zlo_rope_cos_sin(const int32_t* pos, float* cosv, float* sinv, int64_t s, int64_t d,
                      float base, int neox) {
    for (int64_t t = 0; t < s; ++t)
        for (int col = 0; col < d; ++col) {
            int i = half_dim_index(col, (int)d / 2, neox);
            float inv_freq = powf(base, -(float)(i * 2) / (float)d);
            float freq = (float)pos[t] * inv_freq;
            cosv[t * d + col] = cosf(freq);
            sinv[t * d + col] = sinf(freq);
        }
}

/* KERNEL_rope_cos_sin_llama3 (rope_preparer.cu:124-160) */
void zlo_rope_cos_sin_llama3(const int32_t* pos, float* cosv, float* sinv, int64_t s, int64_t d,
                             float base, float factor, float low_freq_factor,
                             float high_freq_factor, float old_context_len, int neox) {
    for (int64_t t = 0; t < s; ++t)
        for (int col = 0; col < d; ++col) {
            int i = half_dim_index(col, (int)d / 2, neox);
            float inv_freq = powf(base, -(float)(i * 2) / (float)d);
            float low_wl = old_context_len / low_freq_factor;
            float high_wl = old_context_len / high_freq_factor;
            float pi = 3.141592653589793f;
            float wavelen = 2.f * pi / inv_freq;
            if (wavelen < high_wl) {
            } else if (wavelen > low_wl) {
                inv_freq = inv_freq / factor;
            } else {
                float smooth = (old_context_len / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor);
                /* (1-s)*inv/factor + s*inv : nvcc contracts the final mul+add into an fma */
                inv_freq = fmaf(smooth, inv_freq, (1.f - smooth) * inv_freq / factor);
            }
            float freq = (float)pos[t] * inv_freq;
            cosv[t * d + col] = cosf(freq);
            sinv[t * d + col] = sinf(freq);
        }
}

/* RotaryEmbedding::impl "dynamic" angles (src/nn/position/rotary_embedding.cu:19-61): theta grows with the row's sequence
 * length once it passes max_position_embeddings; the exponent dim_head / (dim_head - 2) is an int / int quotient in the
 * reference (:38), kept as written.  seq_len NULL: the row is its own last position (a decode row). */
void zlo_rope_cos_sin_dynamic(const int32_t* pos, const int32_t* seq_len, float* cosv, float* sinv, int64_t s, int64_t d,
                              float base, float factor, float max_pos, int neox) {
    for (int64_t t = 0; t < s; ++t) {
        float theta = base;
        int len = seq_len ? seq_len[t] : pos[t];
        if ((float)len > max_pos)
            theta *= powf((factor * (float)len / max_pos) - (factor - 1.f), (float)((int)d / ((int)d - 2)));
        for (int col = 0; col < d; ++col) {
            int i = half_dim_index(col, (int)d / 2, neox);
            float freq = (float)pos[t] * powf(theta, -(float)(i * 2) / (float)d);
            cosv[t * d + col] = cosf(freq);
            sinv[t * d + col] = sinf(freq);
        }
    }
}

/* YarnImpl's constructor (rotary_embedding.cu:506-553): low / high pair indices and the cos / sin multiplier, in double */
void zlo_yarn_params(double base, int dim_head, int original_max_position, double factor, int beta_fast, int beta_slow,
                     double attn_factor, int deepseek, double mscale, double mscale_all_dim, float* low, float* high,
                     float* out_mscale) {
    const double PI = 3.141592653589793;
    double c_fast = (dim_head * log((double)original_max_position / (beta_fast * 2 * PI))) / (2 * log(base));
    double c_slow = (dim_head * log((double)original_max_position / (beta_slow * 2 * PI))) / (2 * log(base));
    double lo = floor(c_fast), hi = ceil(c_slow);
    *low = (float)(lo > 0. ? lo : 0.);
    *high = (float)(hi < dim_head - 1. ? hi : dim_head - 1.);
    double g1 = factor <= 1. ? 1. : 0.1 * log(factor) + 1.0;
    float m = (float)(g1 * (double)(float)attn_factor);
    if (deepseek) {
        double ga = factor <= 1. ? 1. : 0.1 * mscale * log(factor) + 1.0;
        double gb = factor <= 1. ? 1. : 0.1 * mscale_all_dim * log(factor) + 1.0;
        m = (float)(ga / gb * (double)(float)attn_factor);
    }
    *out_mscale = m;
}

/* KERNEL_yarn_rope_neox_style's angle (rotary_embedding.cu:398-447): cos / sin of pos * blended inv_freq, times mscale */
void zlo_rope_cos_sin_yarn(const int32_t* pos, float* cosv, float* sinv, int64_t s, int64_t d, float base, float factor,
                           float low, float high, float mscale, int neox) {
    for (int64_t t = 0; t < s; ++t)
        for (int col = 0; col < d; ++col) {
            int i = half_dim_index(col, (int)d / 2, neox);
            float fi = (float)i;
            float pos_freq = powf(base, (float)(i * 2) / (float)d);
            float extrap = 1.0f / pos_freq, interp = 1.0f / (factor * pos_freq);
            float ramp = fi <= low ? 0.f : (fi >= high ? 1.f : (fi - low) / (high - low));
            float mask = 1.f - ramp;
            float inv_freq = fmaf(interp, ramp, extrap * mask); /* a*b + c*d: contracted to one fma by nvcc */
            float freq = (float)pos[t] * inv_freq;
            cosv[t * d + col] = cosf(freq) * mscale;
            sinv[t * d + col] = sinf(freq) * mscale;
        }
}

/* per-head norms of q / k: mode 0 = LayerNorm(dim_head) rms on the (rows, heads, dim_head) view, one weight for all heads
 * (Qwen3 q_norm / k_norm, src/nn/attention/attention.cpp:110-113,871-876 -> KERNEL_layernorm_rms, layernorm.cu:14-43);
 * mode 1 = KERNEL_layernorm_multi_head (layernorm.cu:329-353): mean-subtracted, weight (heads, d) */
void zlo_head_norm(const uint16_t* x, const uint16_t* w, uint16_t* out, int64_t rows, int64_t heads, int64_t d,
                   int64_t ld_in, int64_t ld_out, float eps, int mode, int dtype) {
    int threads = mode == 1 ? (int)d : round_up_i((int)d, 32);
    if (threads > 1024) threads = 1024;
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t h = 0; h < heads; ++h) {
            const uint16_t* xr = x + r * ld_in + h * d;
            uint16_t* orow = out + r * ld_out + h * d;
            float part[1024], v[1024];
            for (int64_t i = 0; i < d; ++i) v[i] = T2f(xr[i], dtype);
            if (mode == 1) {
                for (int t = 0; t < threads; ++t) part[t] = t < d ? v[t] : 0.f;
                float mean = block_tree_sum(part, threads) / (float)d;
                for (int64_t i = 0; i < d; ++i) v[i] -= mean;
            }
            for (int t = 0; t < threads; ++t) {
                float acc = 0.f;
                for (int64_t i = t; i < d; i += threads) acc = fmaf(v[i], v[i], acc);
                part[t] = acc;
            }
            float rs = 1.0f / sqrtf(block_tree_sum(part, threads) / (float)d + eps);
            for (int64_t i = 0; i < d; ++i)
                orow[i] = f2T(v[i] * rs * T2f(w[(mode == 1 ? h * d : 0) + i], dtype), dtype);
        }
}

/* rope_one_value (rope_common.cuh:14-34): a*cos -/+ b*sin in fp32 (second product fused) */
static inline float rope_val(float a, float b, float c, float s, int minus) {
    return minus ? fmaf(-b, s, a * c) : fmaf(b, s, a * c);
}

/* KERNEL_rotary_embedding_qk (src/nn/position/rotary_embedding_fuse.cu:19-67): split fused qkv,
 * neox rotation with freq = pos * powf(theta, -2i/D) computed in fp32, one rounding to T */
void zlo_rotary_embedding_qk(const int32_t* pos, const uint16_t* in, uint16_t* q, uint16_t* k,
                             uint16_t* v, int64_t s, int64_t h, int64_t hkv, int64_t d,
                             float theta, int dtype) {
    int64_t all = h + 2 * hkv, half = d / 2;
    for (int64_t t = 0; t < s; ++t)
        for (int64_t head = 0; head < all; ++head) {
            const uint16_t* src = in + (t * all + head) * d;
            if (head >= h + hkv) {
                memcpy(v + (t * hkv + (head - h - hkv)) * d, src, sizeof(uint16_t) * d);
                continue;
            }
            uint16_t* dst = head >= h ? k + (t * hkv + (head - h)) * d : q + (t * h + head) * d;
            for (int64_t col = 0; col < d; ++col) {
                int64_t i = col < half ? col : col - half;
                float freq = (float)pos[t] * powf(theta, -(float)(i * 2) / (float)d);
                float c = cosf(freq), sn = sinf(freq);
                float a = T2f(src[col], dtype);
                float r = col < half ? rope_val(a, T2f(src[col + half], dtype), c, sn, 1)
                                     : rope_val(a, T2f(src[col - half], dtype), c, sn, 0);
                dst[col] = f2T(r, dtype);
            }
        }
}

/* KERNEL_rope_qk_with_cache (src/nn/position/rotary_embedding_fuse_cache.cu:23-64) */
void zlo_rope_qk_cache(const float* cosv, const float* sinv, const uint16_t* in, uint16_t* q,
                       uint16_t* k, uint16_t* v, int64_t s, int64_t h, int64_t hkv, int64_t d,
                       int neox, int dtype) {
    int64_t all = h + 2 * hkv, half = d / 2;
    for (int64_t t = 0; t < s; ++t)
        for (int64_t head = 0; head < all; ++head) {
            const uint16_t* src = in + (t * all + head) * d;
            if (head >= h + hkv) {
                memcpy(v + (t * hkv + (head - h - hkv)) * d, src, sizeof(uint16_t) * d);
                continue;
            }
            uint16_t* dst = head >= h ? k + (t * hkv + (head - h)) * d : q + (t * h + head) * d;
            for (int64_t col = 0; col < d; ++col) {
                float c = cosv[t * d + col], sn = sinv[t * d + col];
                float a = T2f(src[col], dtype);
                float r;
                if (neox) r = col < half ? rope_val(a, T2f(src[col + half], dtype), c, sn, 1)
                                         : rope_val(a, T2f(src[col - half], dtype), c, sn, 0);
                else r = (col % 2 == 0) ? rope_val(a, T2f(src[col + 1], dtype), c, sn, 1)
                                        : rope_val(a, T2f(src[col - 1], dtype), c, sn, 0);
                dst[col] = f2T(r, dtype);
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a14: KV scatter                                                                             */
/* ------------------------------------------------------------------------------------------ */

/* KERNEL_copy_to_rag_buffer2 (src/kvcache/ragged_buffer_kernel.cu:194-222): placement < 0 skips */
void zlo_copy_to_rag_buffer2(const int32_t* placement, const int32_t* buf_lens,
                             const uint16_t* k_src, const uint16_t* v_src,
                             uint16_t* const* k_bufs, uint16_t* const* v_bufs,
                             int64_t b, int64_t len_q, int64_t hkv, int64_t d, int bshd) {
    for (int64_t bi = 0; bi < b; ++bi)
        for (int64_t qi = 0; qi < len_q; ++qi) {
            int64_t xi = bi * len_q + qi;
            int64_t p = placement[xi];
            if (p < 0) continue;
            int64_t len_buf = buf_lens[bi];
            for (int64_t head = 0; head < hkv; ++head) {
                int64_t so = (xi * hkv + head) * d;
                int64_t dof = bshd ? (p * hkv + head) * d : (head * len_buf + p) * d;
                memcpy(k_bufs[bi] + dof, k_src + so, sizeof(uint16_t) * d);
                memcpy(v_bufs[bi] + dof, v_src + so, sizeof(uint16_t) * d);
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a15: decode attention                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* q . k for one key: multiply_q_k_block (attention_kernel.cu:26-50) for D == 128 (lane handles 4
 * consecutive d, fma chain from 0, 32-lane tree); other D: one sequential fp32 fma chain
 * (association of the D != 128 variants is not replayed). */
static float qk_dot(const uint16_t* q, const uint16_t* kr, int64_t d, int dtype) {
    if (d == 128) {
        float lane[32];
        for (int l = 0; l < 32; ++l) {
            float res = 0.f;
            for (int j = 0; j < 4; ++j) res = fmaf(T2f(q[4 * l + j], dtype), T2f(kr[4 * l + j], dtype), res);
            lane[l] = res;
        }
        return warp32_tree_sum(lane);
    }
    float res = 0.f;
    for (int64_t j = 0; j < d; ++j) res = fmaf(T2f(q[j], dtype), T2f(kr[j], dtype), res);
    return res;
}

/* softmax_mask_block (attention_kernel.cu:434-489) with blockDim = 1024: every thread's partial sum
 * starts at 1e-20, max starts at -1e20; p = e / Z (fp32 division).  Returns max and Z. */
static void softmax_mask(float* s, const int8_t* mask, float scale, int64_t len, float* out_max,
                         float* out_sum) {
    float mx = -1e20f;
    for (int64_t i = 0; i < len; ++i) {
        s[i] = mask[i] ? s[i] * scale : -INFINITY;
        mx = fmaxf(mx, s[i]);
    }
    float part[1024];
    for (int t = 0; t < 1024; ++t) {
        float acc = 1e-20f;
        for (int64_t i = t; i < len; i += 1024) {
            float e = expf(s[i] - mx);
            s[i] = e;
            acc += e;
        }
        part[t] = acc;
    }
    float z = block_tree_sum(part, 1024);
    for (int64_t i = 0; i < len; ++i) s[i] = s[i] / z;
    if (out_max) *out_max = mx;
    if (out_sum) *out_sum = z;
}

/* multiply_score_v_block2 (attention_kernel.cu:166-202): NUM_SPLIT = 1024/D contiguous shards of
 * ceil(len/NUM_SPLIT) keys, each an fp32 fma chain, shards summed by a shuffle-down tree */
static void score_v(const float* p, const uint16_t* v, int64_t stride, int64_t len, int64_t d,
                    int dtype, float* out) {
    int ns = (1024 % d == 0 && d <= 512) ? (int)(1024 / d) : 1;
    int64_t ls = (len + ns - 1) / ns;
    for (int64_t col = 0; col < d; ++col) {
        float sh[32];
        for (int s = 0; s < ns; ++s) {
            float res = 0.f;
            int64_t st = s * ls, en = st + ls < len ? st + ls : len;
            for (int64_t i = st; i < en; ++i) res = fmaf(p[i], T2f(v[i * stride + col], dtype), res);
            sh[s] = res;
        }
        for (int off = ns / 2; off > 0; off >>= 1)
            for (int l = 0; l < off; ++l) sh[l] += sh[l + off];
        out[col] = sh[0];
    }
}

static int64_t mask_offset(const int32_t* buf_lens, int64_t bi, int64_t len_q) {
    int64_t off = 0;
    for (int64_t i = 0; i < bi; ++i) off += buf_lens[i];
    return off * len_q;
}

/* KERNEL_mqa_rag_buffer1 (attention_kernel.cu:673-725), the default GQA decode kernel */
void zlo_mqa_rag_buffer(const uint16_t* q, const int32_t* buf_lens,
                        const uint16_t* const* k_bufs, const uint16_t* const* v_bufs,
                        const int8_t* mask, uint16_t* out, int64_t b, int64_t len_q, int64_t h,
                        int64_t hkv, int64_t d, float scale, int bshd, int dtype) {
    int64_t m_query = h / hkv;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int64_t bi = 0; bi < b; ++bi)
        for (int64_t head = 0; head < h; ++head) {
            int64_t len = buf_lens[bi], hk = head / m_query;
            int64_t stride = bshd ? hkv * d : d;
            int64_t off = bshd ? hk * d : hk * len * d;
            float* s = (float*)malloc(sizeof(float) * (len > 0 ? len : 1));
            float* o = (float*)malloc(sizeof(float) * d);
            for (int64_t qi = 0; qi < len_q; ++qi) {
                const uint16_t* qv = q + ((bi * len_q + qi) * h + head) * d;
                for (int64_t j = 0; j < len; ++j) s[j] = qk_dot(qv, k_bufs[bi] + off + j * stride, d, dtype);
                const int8_t* mk = mask + mask_offset(buf_lens, bi, len_q) + qi * len;
                softmax_mask(s, mk, scale, len, NULL, NULL);
                score_v(s, v_bufs[bi] + off, stride, len, d, dtype, o);
                for (int64_t c = 0; c < d; ++c) out[((bi * len_q + qi) * h + head) * d + c] = f2T(o[c], dtype);
            }
            free(s);
            free(o);
        }
}

/* KERNEL_mqa_rag_buffer_split_kv + KERNEL_mqa_combine (attention_kernel.cu:729-800, 880-923):
 * len <= 512 -> single pass; else num_split contiguous chunks of ceil(len/num_split), each
 * normalised by its own Z, combined with scale2 = Z_s / Zg * exp(m_s - m) in fp32 */
void zlo_mqa_rag_buffer_split_kv(const uint16_t* q, const int32_t* buf_lens,
                                 const uint16_t* const* k_bufs, const uint16_t* const* v_bufs,
                                 const int8_t* mask, uint16_t* out, int64_t b, int64_t len_q,
                                 int64_t h, int64_t hkv, int64_t d, float scale, int bshd,
                                 int dtype, int num_split) {
    int64_t m_query = h / hkv;
    for (int64_t bi = 0; bi < b; ++bi)
        for (int64_t head = 0; head < h; ++head) {
            int64_t len = buf_lens[bi], hk = head / m_query;
            int64_t stride = bshd ? hkv * d : d;
            int64_t off = bshd ? hk * d : hk * len * d;
            float* s = (float*)malloc(sizeof(float) * (len > 0 ? len : 1));
            float* cache = (float*)malloc(sizeof(float) * d * (num_split > 0 ? num_split : 1));
            for (int64_t qi = 0; qi < len_q; ++qi) {
                const uint16_t* qv = q + ((bi * len_q + qi) * h + head) * d;
                const int8_t* mk = mask + mask_offset(buf_lens, bi, len_q) + qi * len;
                uint16_t* dst = out + ((bi * len_q + qi) * h + head) * d;
                if (len <= 512 || num_split <= 1) {
                    for (int64_t j = 0; j < len; ++j) s[j] = qk_dot(qv, k_bufs[bi] + off + j * stride, d, dtype);
                    softmax_mask(s, mk, scale, len, NULL, NULL);
                    score_v(s, v_bufs[bi] + off, stride, len, d, dtype, cache);
                    for (int64_t c = 0; c < d; ++c) dst[c] = f2T(cache[c], dtype);
                    continue;
                }
                float lmax[32], lsum[32];
                int64_t ls = (len + num_split - 1) / num_split;
                for (int sp = 0; sp < num_split; ++sp) {
                    int64_t st = sp * ls;
                    int64_t l2 = ls < len - st ? ls : len - st;
                    if (l2 < 0) l2 = 0;
                    for (int64_t j = 0; j < l2; ++j) s[j] = qk_dot(qv, k_bufs[bi] + off + (st + j) * stride, d, dtype);
                    softmax_mask(s, mk + st, scale, l2, &lmax[sp], &lsum[sp]);
                    score_v(s, v_bufs[bi] + off + st * stride, stride, l2, d, dtype, cache + sp * d);
                }
                float t[32];
                float gmax = -1e20f;
                for (int sp = 0; sp < num_split; ++sp) gmax = fmaxf(gmax, lmax[sp]);
                float scale1[32];
                for (int l = 0; l < 32; ++l) {
                    scale1[l] = l < num_split ? expf(lmax[l] - gmax) : 0.f;
                    t[l] = l < num_split ? lsum[l] * scale1[l] : 0.f;
                }
                float gsum = warp32_tree_sum(t);
                for (int64_t c = 0; c < d; ++c) {
                    float res = 0.f;
                    for (int sp = 0; sp < num_split; ++sp)
                        res = fmaf(cache[sp * d + c], lsum[sp] / gsum * scale1[sp], res);
                    dst[c] = f2T(res, dtype);
                }
            }
            free(s);
            free(cache);
        }
}

void zlo_mqa_rag_buffer_exact(const uint16_t* q, const int32_t* buf_lens,
                              const uint16_t* const* k_bufs, const uint16_t* const* v_bufs,
                              const int8_t* mask, double* out, int64_t b, int64_t len_q, int64_t h,
                              int64_t hkv, int64_t d, float scale, int bshd, int dtype) {
    int64_t m_query = h / hkv;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int64_t bi = 0; bi < b; ++bi)
        for (int64_t head = 0; head < h; ++head) {
            int64_t len = buf_lens[bi], hk = head / m_query;
            int64_t stride = bshd ? hkv * d : d;
            int64_t off = bshd ? hk * d : hk * len * d;
            double* s = (double*)malloc(sizeof(double) * (len > 0 ? len : 1));
            for (int64_t qi = 0; qi < len_q; ++qi) {
                const uint16_t* qv = q + ((bi * len_q + qi) * h + head) * d;
                const int8_t* mk = mask + mask_offset(buf_lens, bi, len_q) + qi * len;
                double mx = -1e20;
                for (int64_t j = 0; j < len; ++j) {
                    double acc = 0.0;
                    const uint16_t* kr = k_bufs[bi] + off + j * stride;
                    for (int64_t c = 0; c < d; ++c) acc += (double)T2f(qv[c], dtype) * (double)T2f(kr[c], dtype);
                    s[j] = mk[j] ? acc * (double)scale : -INFINITY;
                    if (s[j] > mx) mx = s[j];
                }
                double z = 1e-20;
                for (int64_t j = 0; j < len; ++j) { s[j] = exp(s[j] - mx); z += s[j]; }
                for (int64_t c = 0; c < d; ++c) {
                    double acc = 0.0;
                    for (int64_t j = 0; j < len; ++j)
                        acc += s[j] / z * (double)T2f(v_bufs[bi][off + j * stride + c], dtype);
                    out[((bi * len_q + qi) * h + head) * d + c] = acc;
                }
            }
            free(s);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a18: element-wise                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* element_add_scale (src/nn/block/block_kernel.cu:8-17): arithmetic in T, one rounding per op */
void zlo_element_add_scale(const uint16_t* a, const uint16_t* b, uint16_t* c, int64_t n,
                           float scale, int scale_residual, int dtype) {
    uint16_t st = f2T(scale, dtype);
    for (int64_t i = 0; i < n; ++i) {
        double av = T2f(a[i], dtype), bv = T2f(b[i], dtype), sv = T2f(st, dtype);
        if (scale_residual) c[i] = d2T((double)T2f(d2T(av + bv, dtype), dtype) * sv, dtype);
        else c[i] = d2T(av + (double)T2f(d2T(bv * sv, dtype), dtype), dtype);
    }
}

/* KERNEL_silu_mul_inplace (src/nn/linear/activation_kernel.cu:70-80), silu (functions/activation.cuh:12-14) */
void zlo_silu_mul(const uint16_t* inp, const uint16_t* in2, uint16_t* out, int64_t n, int dtype) {
    for (int64_t i = 0; i < n; ++i) {
        float x = T2f(inp[i], dtype);
        float sl = x / (1.0f + expf(-x));
        out[i] = f2T(sl * T2f(in2[i], dtype), dtype);
    }
}

/* KERNEL_gelu_mul_inplace (activation_kernel.cu:59-69), gelu (functions/activation.cuh:8-10) */
void zlo_gelu_mul(const uint16_t* inp, const uint16_t* in2, uint16_t* out, int64_t n, int dtype) {
    for (int64_t i = 0; i < n; ++i) {
        float x = T2f(inp[i], dtype);
        float ge = 0.5f * x * (1.0f + tanhf(0.7978845608028654f * x * (1.0f + 0.044715f * x * x)));
        out[i] = f2T(ge * T2f(in2[i], dtype), dtype);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a22 / a21                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* BM_KERNEL(embedding) (src/nn/embedding/embedding.cu:23-44) */
void zlo_embedding(const int32_t* ids, const uint16_t* weight, uint16_t* out, int64_t s,
                   int64_t dim, int32_t begin, int32_t end, float scale, int dtype) {
    for (int64_t t = 0; t < s; ++t) {
        int32_t id = ids[t];
        int in_range = id >= begin && id < end;
        for (int64_t c = 0; c < dim; ++c)
            out[t * dim + c] = in_range ? f2T(T2f(weight[(int64_t)(id - begin) * dim + c], dtype) * scale, dtype)
                                        : f2T(0.f, dtype);
    }
}

/* functions::Gemm (bm/functions/gemm.cpp:258-344) with fp32 compute type: y = T(alpha*(x.w^T) + bias).
 * cuBLASLt's summation order is not in the tree -> sequential fp32 fma chain (association unpinned). */
void zlo_gemm_nt(const uint16_t* x, const uint16_t* w, const uint16_t* bias, uint16_t* y,
                 int64_t m, int64_t n, int64_t k, float alpha, int dtype) {
#pragma omp parallel for schedule(static)
    for (int64_t nn = 0; nn < n; ++nn)
        for (int64_t mm = 0; mm < m; ++mm) {
            float acc = 0.f;
            for (int64_t kk = 0; kk < k; ++kk) acc = fmaf(T2f(x[mm * k + kk], dtype), T2f(w[nn * k + kk], dtype), acc);
            float r = alpha * acc + (bias ? T2f(bias[nn], dtype) : 0.f);
            y[mm * n + nn] = f2T(r, dtype);
        }
}

void zlo_gemm_nt_exact(const uint16_t* x, const uint16_t* w, const uint16_t* bias, double* y,
                       int64_t m, int64_t n, int64_t k, float alpha, int dtype) {
#pragma omp parallel for schedule(static)
    for (int64_t nn = 0; nn < n; ++nn)
        for (int64_t mm = 0; mm < m; ++mm) {
            double acc = 0.0;
            for (int64_t kk = 0; kk < k; ++kk) acc += (double)T2f(x[mm * k + kk], dtype) * (double)T2f(w[nn * k + kk], dtype);
            y[mm * n + nn] = (double)alpha * acc + (bias ? (double)T2f(bias[nn], dtype) : 0.0);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a8..a11: INT8                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* quant_calc_scale<T,127> (src/nn/quant/int8/quant_kernel.cu:15-47) */
void zlo_quant_calc_scale(const uint16_t* x, int8_t* q, float* scale, int64_t m, int64_t k,
                          int dtype) {
    for (int64_t r = 0; r < m; ++r) {
        float amax = 0.f;
        for (int64_t i = 0; i < k; ++i) {
            float v = fabsf(T2f(x[r * k + i], dtype));
            amax = v > amax ? v : amax;
        }
        float bs = 127.f / amax;
        for (int64_t i = 0; i < k; ++i) q[r * k + i] = (int8_t)nearbyintf(T2f(x[r * k + i], dtype) * bs);
        scale[r] = amax / 127.f;
    }
}

/* fuse_layernorm_rms_quant (quant_kernel.cu:106-151): abs-max of the fp32 products x*w is reduced
 * in T (rounded to T), 127.0/amax and amax*rs/127. go through double */
void zlo_rmsnorm_quant(const uint16_t* x, const uint16_t* w, uint16_t* out, int8_t* q,
                       float* out_scale, int64_t rows, int64_t dim, float eps, float scale,
                       int dtype) {
    int threads = round_up_i((int)dim, 32);
    if (threads > 1024) threads = 1024;
    for (int64_t r = 0; r < rows; ++r) {
        float* vw = (float*)malloc(sizeof(float) * dim);
        float part[1024];
        float amax = 0.f;
        for (int t = 0; t < threads; ++t) {
            float acc = 0.f;
            for (int64_t i = t; i < dim; i += threads) {
                float v = T2f(x[r * dim + i], dtype);
                acc = fmaf(v, v, acc);
                float p = v * T2f(w[i], dtype);
                vw[i] = p;
                float pa = fabsf(p);
                amax = pa > amax ? pa : amax;
            }
            part[t] = acc;
        }
        float ss = block_tree_sum(part, threads);
        float rs = 1.0f / sqrtf(ss / (float)dim + eps);
        amax = T2f(f2T(amax, dtype), dtype);
        float bs = (float)(127.0 / (double)amax);
        for (int64_t i = 0; i < dim; ++i) {
            float v = vw[i] / scale;
            out[r * dim + i] = f2T(v * rs, dtype);
            q[r * dim + i] = (int8_t)nearbyintf(v * bs);
        }
        out_scale[r] = (float)((double)(amax * rs) / 127.);
        free(vw);
    }
}

/* cuBLASLt IMMA int8 x int8 -> int32 (src/nn/linear/linear.cpp:557-635): exact integer result */
void zlo_int8_gemm_nt(const int8_t* a, const int8_t* b, int32_t* c, int64_t m, int64_t n,
                      int64_t k) {
#pragma omp parallel for schedule(static)
    for (int64_t nn = 0; nn < n; ++nn)
        for (int64_t mm = 0; mm < m; ++mm) {
            int32_t acc = 0;
            for (int64_t kk = 0; kk < k; ++kk) acc += (int32_t)a[mm * k + kk] * (int32_t)b[nn * k + kk];
            c[mm * n + nn] = acc;
        }
}

/* KERNEL_quant_scale_back (quant_kernel.cu:231-246) */
void zlo_quant_scale_back(const int32_t* c, const float* sx, const uint16_t* sy, uint16_t* out,
                          int64_t m, int64_t n, int dtype) {
    for (int64_t r = 0; r < m; ++r)
        for (int64_t col = 0; col < n; ++col)
            out[r * n + col] = f2T((float)c[r * n + col] * sx[r] * T2f(sy[col], dtype), dtype);
}

/* quant_back_act_mul (quant_kernel.cu:589-614): act 0 = silu, 1 = gelu */
void zlo_quant_back_act_mul(const int32_t* a, const float* asx, const uint16_t* asy,
                            const int32_t* b, const float* bsx, const uint16_t* bsy,
                            uint16_t* out, int64_t m, int64_t n, int act, int dtype) {
    for (int64_t r = 0; r < m; ++r)
        for (int64_t col = 0; col < n; ++col) {
            float ab = (float)a[r * n + col] * asx[r] * T2f(asy[col], dtype);
            float bb = (float)b[r * n + col] * bsx[r] * T2f(bsy[col], dtype);
            float gate = act == 0 ? ab / (1.0f + expf(-ab))
                                  : 0.5f * ab * (1.0f + tanhf(0.7978845608028654f * ab * (1.0f + 0.044715f * ab * ab)));
            out[r * n + col] = f2T(bb * gate, dtype);
        }
}

/* quant_scale_back3 (quant_kernel.cu:311-384): the fused qkv GEMM's int32 result split into q | k | v */
void zlo_quant_scale_back3(const int32_t* c, const float* sx, const uint16_t* sy, uint16_t* q, uint16_t* k,
                           uint16_t* v, int64_t m, int64_t n, int64_t dim_q, int64_t dim_kv, int dtype) {
    for (int64_t r = 0; r < m; ++r)
        for (int64_t col = 0; col < n; ++col) {
            const uint16_t val = f2T((float)c[r * n + col] * sx[r] * T2f(sy[col], dtype), dtype);
            if (col < dim_q) q[r * dim_q + col] = val;
            else if (col < dim_q + dim_kv) k[r * dim_kv + col - dim_q] = val;
            else v[r * dim_kv + col - dim_q - dim_kv] = val;
        }
}

/* quant_back_element_add_scale (quant_kernel.cu:530-545): T((float(a) sx sy + float(b)) * scale) */
void zlo_quant_back_element_add_scale(const int32_t* a, const float* sx, const uint16_t* sy, const uint16_t* b,
                                      float scale, uint16_t* out, int64_t m, int64_t n, int dtype) {
    for (int64_t r = 0; r < m; ++r)
        for (int64_t col = 0; col < n; ++col) {
            const float qb = (float)a[r * n + col] * sx[r] * T2f(sy[col], dtype);
            out[r * n + col] = f2T((qb + T2f(b[r * n + col], dtype)) * scale, dtype);
        }
}

/* quant_back_transpose (quant_kernel.cu:475-489): (batch, len_q, heads, d) int32 -> (batch, heads, len_q, d) T */
void zlo_quant_back_transpose(const int32_t* inp, const float* sx, const uint16_t* sy, uint16_t* out,
                              int64_t batch, int64_t len_q, int64_t heads, int64_t d, int dtype) {
    for (int64_t b = 0; b < batch; ++b)
        for (int64_t t = 0; t < len_q; ++t)
            for (int64_t h = 0; h < heads; ++h)
                for (int64_t e = 0; e < d; ++e) {
                    const float x = sx[b * len_q + t], y = T2f(sy[h * d + e], dtype);
                    out[((b * heads + h) * len_q + t) * d + e] =
                        f2T((float)inp[((b * len_q + t) * heads + h) * d + e] * x * y, dtype);
                }
}

/* quant_back_copy_to_buffer (quant_kernel.cu:389-415): scatter the scaled-back k/v rows into
 * (batch, heads, len_buf, d) buffers at placement[b, t] (negative = padded row, skipped; NULL = identity) */
void zlo_quant_back_copy_to_buffer(const int32_t* src, const float* sx, const uint16_t* sy, const int32_t* placement,
                                   uint16_t* dst, int64_t batch, int64_t len_kv, int64_t heads, int64_t d,
                                   int64_t len_buf, int64_t src_stride, int64_t dst_stride, int64_t place_stride,
                                   int dtype) {
    for (int64_t b = 0; b < batch; ++b)
        for (int64_t t = 0; t < len_kv; ++t) {
            const int64_t pos = placement ? placement[b * place_stride + t] : t;
            if (pos < 0) continue;
            for (int64_t h = 0; h < heads; ++h)
                for (int64_t e = 0; e < d; ++e) {
                    const float x = sx[b * len_kv + t], y = T2f(sy[h * d + e], dtype);
                    dst[b * dst_stride + (h * len_buf + pos) * d + e] =
                        f2T((float)src[b * src_stride + (t * heads + h) * d + e] * x * y, dtype);
                }
        }
}

/* quant_calc_scale with a zero point (quant_kernel.cu:15-47, q_zero != 0): the INT8 KV cache rows,
 * u8 = q_zero + rint(x * 127 / amax), scale = amax / 127 (attention.cpp:656-661 uses q_zero = 128) */
void zlo_quant_calc_scale_zp(const uint16_t* x, uint8_t* q, float* scale, int64_t m, int64_t k, int q_zero, int dtype) {
    for (int64_t r = 0; r < m; ++r) {
        float amax = 0.f;
        for (int64_t i = 0; i < k; ++i) {
            float v = fabsf(T2f(x[r * k + i], dtype));
            amax = v > amax ? v : amax;
        }
        float bs = amax > 0.f ? 127.f / amax : 0.f; /* a zero row: codes q_zero, scale 0 (the reference divides by 0) */
        for (int64_t i = 0; i < k; ++i)
            q[r * k + i] = (uint8_t)((float)q_zero + nearbyintf(T2f(x[r * k + i], dtype) * bs));
        scale[r] = amax / 127.f;
    }
}

/* Decode attention over an INT8 KV cache, exact (fp64) statement of KERNEL_mqa_rag_buffer_split_kv_quant
 * (attention_kernel.cu:802-878, quant_attention.cuh:39-123):
 *   logit_j = scale * sk_j * sum_d q_d (K_jd - 128);  p = softmax over the visible j;
 *   out_d = sum_j p_j * sv_j * (V_jd - 128).
 * K/V u8 (len_buf, hkv, d) [bshd] or (hkv, len_buf, d); scales float (len_buf, hkv) / (hkv, len_buf). */
void zlo_mqa_rag_buffer_quant_exact(const uint16_t* q, const int32_t* buf_lens, const uint8_t* const* k_bufs,
                                    const uint8_t* const* v_bufs, const float* const* k_scales,
                                    const float* const* v_scales, const int8_t* mask, double* out, int64_t b,
                                    int64_t len_q, int64_t h, int64_t hkv, int64_t d, float scale, int bshd, int dtype) {
    int64_t n_rep = h / hkv, mask_off = 0;
    for (int64_t bi = 0; bi < b; ++bi) {
        int64_t len = buf_lens[bi];
        double* lg = (double*)malloc(sizeof(double) * (len > 0 ? len : 1));
        for (int64_t qi = 0; qi < len_q; ++qi)
            for (int64_t hh = 0; hh < h; ++hh) {
                int64_t hk = hh / n_rep;
                const uint16_t* qv = q + ((bi * len_q + qi) * h + hh) * d;
                double mx = -1e300;
                for (int64_t j = 0; j < len; ++j) {
                    if (!mask[mask_off + qi * len + j]) { lg[j] = -1e300; continue; }
                    const uint8_t* kr = k_bufs[bi] + (bshd ? (j * hkv + hk) * d : (hk * len + j) * d);
                    double dot = 0;
                    for (int64_t e = 0; e < d; ++e) dot += (double)T2f(qv[e], dtype) * ((double)kr[e] - 128.0);
                    double sk = k_scales[bi][bshd ? j * hkv + hk : hk * len + j];
                    lg[j] = dot * sk * (double)scale;
                    if (lg[j] > mx) mx = lg[j];
                }
                double z = 0;
                for (int64_t j = 0; j < len; ++j) {
                    lg[j] = lg[j] <= -1e299 ? 0.0 : exp(lg[j] - mx);
                    z += lg[j];
                }
                double* o = out + ((bi * len_q + qi) * h + hh) * d;
                for (int64_t e = 0; e < d; ++e) o[e] = 0;
                for (int64_t j = 0; j < len; ++j) {
                    if (lg[j] == 0.0) continue;
                    const uint8_t* vr = v_bufs[bi] + (bshd ? (j * hkv + hk) * d : (hk * len + j) * d);
                    double w = lg[j] / (z + 1e-20) * (double)v_scales[bi][bshd ? j * hkv + hk : hk * len + j];
                    for (int64_t e = 0; e < d; ++e) o[e] += w * ((double)vr[e] - 128.0);
                }
            }
        free(lg);
        mask_off += len_q * len;
    }
}

/* ==== native AWQ ======================================================================================
 * On-disk AWQ "gemm" layout: qweight (K, N/8) int32, qzeros (K/G, N/8) int32, scales (K/G, N) fp16; nibble i of a word
 * holds column 8c + order[i], order = {0,2,4,6,1,3,5,7} (the inverse of the extraction table {0,4,1,5,2,6,3,7},
 * src/nn/quant/gptq/utils.cu:33,132; dequantize_s4_to_fp16x2, src/nn/quant/awq/dequantize.cuh:45-112, returns the eight
 * values in column order).  Zero points are stored as used (no -1). */
static const int kAwqOrder[8] = {0, 2, 4, 6, 1, 3, 5, 7};

/* dequantize_weights (src/nn/quant/awq/gemm_kernels.cu:277-330): W16[k,n] = rn16( fp16(q - z) * s ) -- the sub.f16x2 is
 * exact on these small integers, the fma.rn.f16x2(., s, 0) rounds the product once */
void zlo_awq_dequantize(const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales, uint16_t* out,
                        int64_t k, int64_t n, int64_t g) {
    const int64_t n8 = n / 8;
#pragma omp parallel for schedule(static)
    for (int64_t kk = 0; kk < k; ++kk)
        for (int64_t c = 0; c < n8; ++c) {
            const uint32_t w = qweight[kk * n8 + c], z = qzeros[(kk / g) * n8 + c];
            for (int i = 0; i < 8; ++i) {
                const int col = (int)(8 * c + kAwqOrder[i]);
                const int d = (int)((w >> (4 * i)) & 0xF) - (int)((z >> (4 * i)) & 0xF);
                out[kk * n + col] = h_mul(zlo_f32_to_f16((float)d), scales[(kk / g) * n + col]);
            }
        }
}

/* awq_gemm (gemm_kernels.cu:404-468) = gemm_forward_4bit_cuda_m16nXk32 (:32-275) + KERNEL_sum_dim0 (:381-400):
 * the 32-row K tiles are dealt round-robin to split_k_iters workgroups (tile t = i * split_k_iters + z, :114-117), each
 * accumulates its products x * W16 in fp32 (mma.sync m16n8k16 f32 += f16 x f16: products exact, summed here in k order --
 * the tensor core's internal order inside a k16 step is not specified) and writes its partial C as FP16 (:269); the
 * partials are then added in fp32 in split order and rounded to fp16.  w16 = zlo_awq_dequantize's output. */
void zlo_awq_gemm(const uint16_t* x, const uint16_t* w16, uint16_t* y, int64_t m, int64_t n, int64_t k, int64_t split_k_iters) {
    const int64_t tiles = (k + 31) / 32;
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t mm = 0; mm < m; ++mm)
        for (int64_t nn = 0; nn < n; ++nn) {
            float total = 0.f;
            for (int64_t z = 0; z < split_k_iters; ++z) {
                float acc = 0.f;
                for (int64_t t = z; t < tiles; t += split_k_iters)
                    for (int64_t kk = 32 * t; kk < 32 * t + 32 && kk < k; ++kk)
                        acc += zlo_f16_to_f32(x[mm * k + kk]) * zlo_f16_to_f32(w16[kk * n + nn]);
                total += zlo_f16_to_f32(zlo_f32_to_f16(acc));
            }
            y[mm * n + nn] = zlo_f32_to_f16(total);
        }
}

/* exact product with the same W16 (fp64 accumulation, no intermediate rounding) */
void zlo_awq_gemm_exact(const uint16_t* x, const uint16_t* w16, double* y, int64_t m, int64_t n, int64_t k) {
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t mm = 0; mm < m; ++mm)
        for (int64_t nn = 0; nn < n; ++nn) {
            double acc = 0.0;
            for (int64_t kk = 0; kk < k; ++kk) acc += (double)zlo_f16_to_f32(x[mm * k + kk]) * (double)zlo_f16_to_f32(w16[kk * n + nn]);
            y[mm * n + nn] = acc;
        }
}

/* ==== W4A8, int8 activations on W4 weights (the M > W4_A8_M_THRES branch of gptq_gemm_k_major,
 * src/nn/quant/gptq/q_gemm_k_major.cu:1036-1073) ====================================================================
 * load time (Int4GPTQ::calc_w4a8_scale, src/nn/linear/linear.cpp:1101-1112): scale[n] = max_k |W16[n,k]| / 127 in fp32;
 * KERNEL_dequant<int8_t, 1> (q_gemm_k_major.cu:843-905): w8[n,k] = int8(nearbyintf(float(W16[n,k]) * (1.f / scale[n]))).
 * forward: a_q, a_s = quant_calc_scale(a); acc = a_q . w8^T (int32, exact); y = half(float(acc) * a_s[m] * scale[n]). */
void zlo_w4a8_weight_to_int8(const uint16_t* w16, int8_t* w8, float* scale, int64_t n, int64_t k) {
#pragma omp parallel for schedule(static)
    for (int64_t nn = 0; nn < n; ++nn) {
        float amax = 0.f;
        for (int64_t kk = 0; kk < k; ++kk) amax = fmaxf(amax, fabsf(zlo_f16_to_f32(w16[nn * k + kk])));
        const float s = amax / 127.f;
        scale[nn] = s;
        const float r = 1.f / s;
        for (int64_t kk = 0; kk < k; ++kk) w8[nn * k + kk] = (int8_t)nearbyintf(zlo_f16_to_f32(w16[nn * k + kk]) * r);
    }
}
void zlo_quant_scale_back_f32(const int32_t* c, const float* sx, const float* sy, uint16_t* out, int64_t m, int64_t n) {
    for (int64_t mm = 0; mm < m; ++mm)
        for (int64_t nn = 0; nn < n; ++nn) out[mm * n + nn] = zlo_f32_to_f16((float)c[mm * n + nn] * sx[mm] * sy[nn]);
}

/* ---- INT8-compressed tensor-parallel reduce: the three kernels of ModelContext::reduce_tp_int8
 * (src/model/model_context.cpp:244-326; src/nn/quant/int8/quant_reduce_kernel.cu:13-105, 107-150, 270-330).  TEST INFRASTRUCTURE.
 *   quant_group_32:        per group of 32: abs_max = max |v| (exact: warpReduceMaxB<T> of values that ARE T);
 *                          q = int8(nearbyintf(v * 127.0f / abs_max)) [fmul, then fdiv]; scale = T(abs_max / 127.0f)
 *   dequant_sum_quant_g32: sum = f32(my); for r < WS - 1: sum = fmaf(f32(q_r), f32(T scale_r), sum) (nvcc contracts `sum += a * b`);
 *                          abs_max = max over the group of T(|sum|) -- the template argument of warpReduceMaxB<T> rounds the fp32
 *                          magnitude to T first; q = int8(nearbyintf(sum * 127.0f / abs_max)); scale = T(abs_max / 127.0f)
 *   dequant_group_32:      out = T(f32(q) * f32(scale))
 * A group of zeros divides by zero in the reference (NaN codes); here: codes 0, scale 0 (documented deviation, never hit by
 * real activations). */
void zlo_quant_group_32(const uint16_t* x, int8_t* q, uint16_t* scale, int64_t groups, int dtype) {
    for (int64_t g = 0; g < groups; ++g) {
        float amax = 0.f;
        for (int i = 0; i < 32; ++i) {
            float v = fabsf(T2f(x[g * 32 + i], dtype));
            amax = v > amax ? v : amax;
        }
        for (int i = 0; i < 32; ++i) {
            float v = T2f(x[g * 32 + i], dtype);
            q[g * 32 + i] = amax > 0.f ? (int8_t)nearbyintf(v * 127.0f / amax) : 0;
        }
        scale[g] = f2T(amax / 127.0f, dtype);
    }
}

void zlo_dequant_sum_quant_g32(const uint16_t* my, const int8_t* q_others, const uint16_t* scale_others, int8_t* out_q,
                               uint16_t* out_scale, int64_t groups, int world, int dtype) {
    for (int64_t g = 0; g < groups; ++g) {
        float sum[32], amax = 0.f;
        for (int i = 0; i < 32; ++i) {
            float s = T2f(my[g * 32 + i], dtype);
            for (int r = 0; r < world - 1; ++r)
                s = fmaf((float)q_others[(r * groups + g) * 32 + i], T2f(scale_others[r * groups + g], dtype), s);
            sum[i] = s;
            float a = T2f(f2T(fabsf(s), dtype), dtype);
            amax = a > amax ? a : amax;
        }
        for (int i = 0; i < 32; ++i) out_q[g * 32 + i] = amax > 0.f ? (int8_t)nearbyintf(sum[i] * 127.0f / amax) : 0;
        out_scale[g] = f2T(amax / 127.0f, dtype);
    }
}

void zlo_dequant_group_32(const int8_t* q, const uint16_t* scale, uint16_t* out, int64_t groups, int dtype) {
    for (int64_t g = 0; g < groups; ++g)
        for (int i = 0; i < 32; ++i) out[g * 32 + i] = f2T((float)q[g * 32 + i] * T2f(scale[g], dtype), dtype);
}

/* ---- W4A8 with FP8 activations: gptq_gemm_k_major's W4_FP8_ALGO branch (src/nn/quant/gptq/q_gemm_k_major.cu:1003-1035) and
 * nn::fp8::calc_scale / dynamic_scaled_quant (src/nn/quant/fp8/fp8_util.cu:20-29, 56-78, 100-229); weights through
 * KERNEL_dequant<half, 2> (q_gemm_k_major.cu:843-906) with Int4GPTQ::calc_w4a8_scale's per-tensor scale
 * (src/nn/linear/linear.cpp:1124-1129).  TEST INFRASTRUCTURE.
 *   e4m3 = OCP E4M3FN (bias 7, 3 mantissa bits, no infinities, 0x7f = NaN, largest finite 448 = 0x7e); the reference converts
 *   with cvt.rn.satfinite.e4m3x2.f16x2: round to nearest even on the e4m3 grid, magnitudes beyond 448 saturate to 448.
 *   scale = max|x| / MAX (one fp32 per tensor); code = e4m3(T(x) * T(1 / scale)) -- the product is an fp16 hmul2 for fp16
 *   inputs (one rounding to fp16 before the conversion) and an fp32 product rounded to fp16 for bf16 inputs.
 *   GEMM: fp32 accumulation of exact fp8 x fp8 products (cuBLASLt), times scale_a * scale_b, one rounding to fp16; the
 *   accumulation ORDER is the library's -- the oracle sums in fp64 and the parity bar is the fp16 output rounding. */
uint8_t zlo_f32_to_e4m3(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    const uint32_t au = u & 0x7fffffffu;
    if (au > 0x7f800000u) return (uint8_t)(sign | 0x7fu);              /* NaN */
    float a;
    memcpy(&a, &au, 4);
    if (a >= 464.0f) return (uint8_t)(sign | 0x7eu);                   /* beyond the midpoint of 448 and the next grid point: satfinite */
    if (a < 0.015625f) {                                               /* below 2^-6: subnormal grid of 2^-9 */
        return (uint8_t)(sign | (uint8_t)nearbyintf(a * 512.0f));      /* 8 = the smallest normal: same code */
    }
    uint32_t r = au + 0x7ffffu + ((au >> 20) & 1u);                    /* RNE at bit 20 (3 mantissa bits kept) */
    const int e = (int)(r >> 23) - 127 + 7;
    uint32_t code = ((uint32_t)e << 3) | ((r >> 20) & 7u);
    if (code > 0x7eu) code = 0x7eu;
    return (uint8_t)(sign | code);
}

float zlo_e4m3_to_f32(uint8_t c) {
    const int e = (c >> 3) & 15, m = c & 7;
    float v;
    if ((c & 0x7f) == 0x7f) v = NAN;
    else if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return (c & 0x80) ? -v : v;
}

void zlo_fp8_calc_scale(const uint16_t* x, int64_t numel, float max_e4m3, float* scale, int dtype) {
    float amax = 0.f;
    for (int64_t i = 0; i < numel; ++i) {
        float v = fabsf(T2f(x[i], dtype));
        amax = v > amax ? v : amax;
    }
    *scale = amax / max_e4m3;
}

void zlo_fp8_cvt_half(const uint16_t* x, int64_t numel, float scale, uint8_t* out, int dtype) {
    const float inv = 1.f / scale;
    for (int64_t i = 0; i < numel; ++i) {
        uint16_t h;
        if (dtype) h = zlo_f32_to_f16(inv * zlo_bf16_to_f32(x[i]));                     /* __floats2half2_rn(scale * float(bf16)) */
        else h = zlo_f32_to_f16(zlo_f16_to_f32(x[i]) * zlo_f16_to_f32(zlo_f32_to_f16(inv)));   /* __hmul2(h2, half2(scale)): exact product, one rounding */
        out[i] = zlo_f32_to_e4m3(zlo_f16_to_f32(h));
    }
}

void zlo_fp8_gemm_nt(const uint8_t* a, const uint8_t* b, float scale_a, float scale_b, uint16_t* out, int64_t m, int64_t n, int64_t k) {
    #pragma omp parallel for
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int64_t t = 0; t < k; ++t) acc += (double)zlo_e4m3_to_f32(a[i * k + t]) * (double)zlo_e4m3_to_f32(b[j * k + t]);
            out[i * n + j] = zlo_f32_to_f16((float)(acc * ((double)scale_a * (double)scale_b)));
        }
}

/* ------------------------------------------------------------------------------------------ */
/* f4 (config 5, first part): FP8 128x128-block linear and the MoE router                       */
/* ------------------------------------------------------------------------------------------ */

/* KERNEL_per_token_cast_to_fp8 (src/nn/quant/fp8/fp8_util.cu:229-275): one scale per (row, 128-column block):
 * amax = max |float(x)| clamped at 1e-4 (max_e4m3 = 448 in the reference), codes = e4m3(float(x) * (448 / amax)) with both
 * operations in fp32, scale = amax / 448 stored at [block * aligned_m + row] (scale_col_major) or [row * nblocks + block]. */
void zlo_fp8_per_token_cast(const uint16_t* x, int64_t ldx, uint8_t* out, int64_t ld_out, float* scale, int64_t aligned_m,
                            int64_t m, int64_t n, int col_major, float max_e4m3, int dtype) {
    const int64_t nb = n / 128;
    for (int64_t r = 0; r < m; ++r)
        for (int64_t b = 0; b < nb; ++b) {
            float amax = 0.f;
            for (int j = 0; j < 128; ++j) amax = fmaxf(amax, fabsf(T2f(x[r * ldx + b * 128 + j], dtype)));
            if (amax < 1e-4f) amax = 1e-4f;
            const float mul = max_e4m3 / amax;
            for (int j = 0; j < 128; ++j) out[r * ld_out + b * 128 + j] = zlo_f32_to_e4m3(T2f(x[r * ldx + b * 128 + j], dtype) * mul);
            scale[col_major ? b * aligned_m + r : r * nb + b] = amax / max_e4m3;
        }
}

/* KERNEL_dequant_fp8_block (fp8_util.cu:325-357): out = T(float(code) * scale[row / 128][col / 128]) */
void zlo_fp8_block_dequant(const uint8_t* w, const float* scale, uint16_t* out, int64_t rows, int64_t cols, int64_t stride_scale, int dtype) {
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c)
            out[r * cols + c] = f2T(zlo_e4m3_to_f32(w[r * cols + c]) * scale[(r / 128) * stride_scale + c / 128], dtype);
}

/* The block-scaled product deep_gemm_fp8_block_h20_group computes (3rd/deep_gemm/deep_gemm_api.h; the kernel itself is a
 * closed binary, so this is the format's definition in fp64, not a restatement of its rounding points):
 * out[m, n] = T( sum_kb sa[kb, m] * sw[g(m)][n / 128][kb] * sum_{k in block kb} a[m, k] * w[g(m)][n, k] ),
 * g(m) = m_indices[m] (rows with a negative index are left untouched), lhs scales column-major with leading dimension aligned_m. */
void zlo_fp8_block_gemm(const uint8_t* a, const float* sa, int64_t aligned_m, const uint8_t* w, const float* sw, const int32_t* m_indices,
                        uint16_t* out, int64_t m, int64_t n, int64_t k, int dtype) {
    const int64_t kb = k / 128, nbw = (n + 127) / 128;
    #pragma omp parallel for
    for (int64_t i = 0; i < m; ++i) {
        const int64_t g = m_indices ? m_indices[i] : 0;
        if (g < 0) continue;
        const uint8_t* wg = w + g * n * k;
        const float* swg = sw + g * nbw * kb;
        for (int64_t j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int64_t b = 0; b < kb; ++b) {
                double blk = 0.0;
                for (int t = 0; t < 128; ++t) blk += (double)zlo_e4m3_to_f32(a[i * k + b * 128 + t]) * (double)zlo_e4m3_to_f32(wg[j * k + b * 128 + t]);
                acc += blk * ((double)sa[b * aligned_m + i] * (double)swg[(j / 128) * kb + b]);
            }
            out[i * n + j] = dtype ? zlo_f32_to_bf16((float)acc) : zlo_f64_to_f16(acc);
        }
    }
}

/* ---- MoE router (src/nn/feedforward/ff_kernel.cu:92-470) ---- */
static inline float warp32_tree_max(float* x) {      /* warpReduceMaxB: shuffle-down tree, lane 0 broadcast */
    for (int off = 16; off > 0; off >>= 1)
        for (int l = 0; l < off; ++l) x[l] = x[l] > x[l + off] ? x[l] : x[l + off];
    return x[0];
}
/* DEV_softmax_inplace with `threads` threads (32: warp trees; more: block trees), data in place */
static void route_softmax(float* data, int n, int threads) {
    float part[1024];
    for (int t = 0; t < threads; ++t) {
        float mx = -1e20f;
        for (int i = t; i < n; i += threads) mx = fmaxf(mx, data[i]);
        part[t] = mx;
    }
    float gmax;
    if (threads > 32) {      /* blockReduceMax: per-warp tree, then warp 0 over the per-warp results padded with -inf */
        float wr[32];
        for (int w = 0; w < 32; ++w) wr[w] = -INFINITY;
        for (int w = 0; w < threads / 32; ++w) wr[w] = warp32_tree_max(part + 32 * w);
        gmax = warp32_tree_max(wr);
    } else gmax = warp32_tree_max(part);
    for (int t = 0; t < threads; ++t) {
        float sm = 1e-20f;
        for (int i = t; i < n; i += threads) {
            data[i] = expf(data[i] - gmax);
            sm += data[i];
        }
        part[t] = sm;
    }
    const float gsum = threads > 32 ? block_tree_sum(part, threads) : warp32_tree_sum(part);
    for (int i = 0; i < n; ++i) data[i] /= gsum;
}

/* KERNEL_top_k_softmax (ff_kernel.cu:174-236), 32 threads per token.  scoring: 1 softmax, 2 "sigmoid" -- the kernel computes
 * 1 / (1 + expf(+x)) (DEV_route_score :158-160, no minus sign: restated as written), 3 linear.  Insertion sort, a later equal
 * value does not displace an earlier one.  Slots k .. top_k_ext - 1: weight 1 (indices untouched).  load counters optional. */
void zlo_moe_top_k_softmax(const uint16_t* logits, int64_t tokens, int num_exp, int k, int top_k_ext, int renormalize, float weight_scale,
                           int scoring, int dtype, float* out_v, int32_t* out_idx, int32_t* worker_load, int32_t* expert_load, int num_worker) {
    for (int64_t q = 0; q < tokens; ++q) {
        float data[256];
        for (int i = 0; i < num_exp; ++i) data[i] = T2f(logits[q * num_exp + i], dtype);
        if (scoring == 1) route_softmax(data, num_exp, 32);
        else if (scoring == 2) for (int i = 0; i < num_exp; ++i) data[i] = 1.f / (1.f + expf(data[i]));
        float value[17];
        int idx[17];
        for (int i = 0; i < k; ++i) { value[i] = -1e20f; idx[i] = 0; }
        for (int j = 0; j < num_exp; ++j) {
            const float v = data[j];
            int i;
            for (i = k - 1; i >= 0; --i) {
                if (v > value[i]) { value[i + 1] = value[i]; idx[i + 1] = idx[i]; }
                else { value[i + 1] = v; idx[i + 1] = j; break; }
            }
            if (i < 0) { value[0] = v; idx[0] = j; }
        }
        float sum_e = 1.f;
        if (renormalize) {
            sum_e = 1.e-20f;
            for (int i = 0; i < k; ++i) sum_e += value[i];
        }
        for (int i = 0; i < k; ++i) {
            out_v[q * top_k_ext + i] = value[i] / sum_e * weight_scale;
            out_idx[q * top_k_ext + i] = idx[i];
            if (worker_load) worker_load[idx[i] % num_worker] += 1;
            if (expert_load) expert_load[idx[i]] += 1;
        }
        for (int i = k; i < top_k_ext; ++i) out_v[q * top_k_ext + i] = 1.f;
    }
}

/* warpBitonicSort<T, N> (ff_kernel.cu:273-291): descending, equal values ordered by smaller position first */
static void bitonic_desc(float* v, int* pos, int n_lanes, int width) {
    for (int base = 0; base < n_lanes; base += width)
        for (int kk = 2; kk <= width; kk *= 2)
            for (int j = kk / 2; j > 0; j /= 2) {
                float nv[32];
                int np[32];
                for (int l = 0; l < width; ++l) {
                    const int lane = l, other = l ^ j;
                    const float v1 = v[base + lane], v2 = v[base + other];
                    const int p1 = pos[base + lane], p2 = pos[base + other];
                    const int desc = ((lane & kk) == 0) ^ 0, upper = (lane & j) != 0;
                    const int take = desc ^ (v1 > v2 || (v1 == v2 && p1 < p2)) ^ upper;
                    nv[l] = take ? v2 : v1;
                    np[l] = take ? p2 : p1;
                }
                for (int l = 0; l < width; ++l) { v[base + l] = nv[l]; pos[base + l] = np[l]; }
            }
}

/* KERNEL_group_topk (ff_kernel.cu:296-456), num_group warps per token: sigmoid 1 / (1 + expf(-x)) or block softmax; per group
 * a 32-lane sort of (score + correction_bias); group score = its best entry; the topk_group best groups (8-lane sort) keep
 * their k best entries; a final 32-lane sort picks k; the OUTPUT weight is the un-biased score when a bias is given.
 * sum over the k weights + 1e-20 when renormalising (shuffle-down tree over 32 lanes, zeros beyond k). */
void zlo_moe_group_topk(const uint16_t* logits, const float* correction_bias, int64_t tokens, int num_exp, int k, int top_k_ext,
                        int renormalize, float weight_scale, int scoring, int num_group, int topk_group, int dtype, float* out_v,
                        int32_t* out_idx, int32_t* worker_load, int32_t* expert_load, int num_worker) {
    const int nig = num_exp / num_group;
    for (int64_t q = 0; q < tokens; ++q) {
        float data[512];
        for (int i = 0; i < num_exp; ++i) data[i] = T2f(logits[q * num_exp + i], dtype);
        if (scoring == 2) for (int i = 0; i < num_exp; ++i) data[i] = 1.f / (1.f + expf(-data[i]));
        else route_softmax(data, num_exp, num_group * 32);
        float gs[32][32];
        int gp[32][32];
        float shared_val[32], group_score[32];
        int shared_pos[32], group_id[32], group_rank[32];
        for (int g = 0; g < num_group; ++g) {
            for (int l = 0; l < 32; ++l) {
                gs[g][l] = -1e20f;
                gp[g][l] = -1;
                if (l < nig) {
                    gp[g][l] = g * nig + l;
                    gs[g][l] = data[gp[g][l]] + (correction_bias ? correction_bias[gp[g][l]] : 0.f);
                }
            }
            bitonic_desc(gs[g], gp[g], 32, 32);
            group_rank[g] = -1;
        }
        for (int l = 0; l < 32; ++l) {
            group_score[l] = l < num_group ? gs[l][0] : -1e20f;
            group_id[l] = l < num_group ? l : 0;      /* shared_gid beyond num_group is uninitialised in the kernel; never selected */
        }
        bitonic_desc(group_score, group_id, 8, 8);     /* only the first 8 lanes matter (num_group <= 8 asserted) */
        for (int t = 0; t < topk_group; ++t) group_rank[group_id[t]] = t;
        for (int l = 0; l < 32; ++l) { shared_val[l] = 0.f; shared_pos[l] = 0; }
        for (int g = 0; g < num_group; ++g)
            if (group_rank[g] >= 0)
                for (int l = 0; l < k; ++l) {
                    shared_val[group_rank[g] * k + l] = gs[g][l];
                    shared_pos[group_rank[g] * k + l] = gp[g][l];
                }
        float fv[32];
        int fp[32];
        for (int l = 0; l < 32; ++l) {
            fv[l] = l < topk_group * k ? shared_val[l] : -1e20f;
            fp[l] = shared_pos[l];
        }
        bitonic_desc(fv, fp, 32, 32);
        float w[32];
        for (int l = 0; l < 32; ++l) w[l] = 0.f;
        for (int l = 0; l < k; ++l) w[l] = correction_bias ? data[fp[l]] : fv[l];
        float sum_e = 1.f;
        if (renormalize) {
            float tree[32];
            for (int l = 0; l < 32; ++l) tree[l] = w[l];
            sum_e = warp32_tree_sum(tree) + 1e-20f;
        }
        for (int l = 0; l < k; ++l) {
            out_v[q * top_k_ext + l] = w[l] / sum_e * weight_scale;
            out_idx[q * top_k_ext + l] = fp[l];
            if (worker_load) worker_load[fp[l] % num_worker] += 1;
            if (expert_load) expert_load[fp[l]] += 1;
        }
        for (int l = k; l < top_k_ext; ++l) { out_v[q * top_k_ext + l] = 1.f; out_idx[q * top_k_ext + l] = 0; }
    }
}

/* ---- MoE dispatch / combine (src/nn/feedforward/ff_kernel.cu:518-1082) ---- */
/* KERNEL_sum_experts (:520-538); nvcc contracts `acc += float(x) * w` into one fused multiply-add */
void zlo_moe_sum_experts(const uint16_t* input, const int32_t* index, const float* weight, uint16_t* out, int64_t seq_len, int k,
                         int64_t dim_model, int dtype) {
    for (int64_t q = 0; q < seq_len; ++q)
        for (int64_t d = 0; d < dim_model; ++d) {
            float acc = 0.f;
            for (int i = 0; i < k; ++i) acc = fmaf(T2f(input[(int64_t)index[q * k + i] * dim_model + d], dtype), weight[q * k + i], acc);
            out[q * dim_model + d] = f2T(acc, dtype);
        }
}
/* KERNEL_sum_experts_arr (:541-600): inputs[e] = expert e's output rows (NULL allowed when never referenced) */
void zlo_moe_sum_experts_arr(const uint16_t* const* inputs, const int32_t* experts, const int32_t* index, const float* weight, uint16_t* out,
                             int64_t seq_len, int k, int64_t dim_model, int exp_parallel, int world_size, int local_rank, int dtype) {
    for (int64_t q = 0; q < seq_len; ++q)
        for (int64_t d = 0; d < dim_model; ++d) {
            float acc = 0.f;
            for (int j = 0; j < k; ++j) {
                const int64_t i = q * k + j;
                const int e = experts[i];
                if (exp_parallel && ((e & (world_size - 1)) != local_rank)) continue;
                if (seq_len == 1) acc = fmaf(T2f(inputs[e][d], dtype), weight[j], acc);
                else acc = fmaf(T2f(inputs[e][(int64_t)index[i] * dim_model + d], dtype), weight[i], acc);
            }
            out[q * dim_model + d] = f2T(acc, dtype);
        }
}
/* KERNEL_route_shared_lb (:797-832) */
void zlo_moe_route_shared_lb(int32_t* exp_ids, const int32_t* worker_load_base, int32_t* worker_load, int32_t* expert_load, int max_load,
                             int world_size, int64_t seq_len, int top_k, int top_k_ext, int num_local_experts) {
    for (int s = 0; s < top_k_ext - top_k; ++s)
        for (int64_t q = 0; q < seq_len; ++q) {
            int r = 0;
            int64_t skip = seq_len * s + q;
            for (;;) {
                const int cap = worker_load_base[r] >= max_load ? 0 : max_load - worker_load_base[r];
                if (skip < cap || r == world_size - 1) break;
                skip -= cap;
                ++r;
            }
            const int e = (num_local_experts + s) * world_size + r;
            exp_ids[q * top_k_ext + top_k + s] = e;
            worker_load[r] += 1;
            expert_load[e] += 1;
        }
}
void zlo_moe_plus_for_sort(const int32_t* exp_ids, int32_t* out, int multiple, int world_size, int64_t numel) {
    for (int64_t i = 0; i < numel; ++i) out[i] = exp_ids[i] + (exp_ids[i] % world_size) * multiple;
}
/* calc_reverse_idx (:886-942): the expert offsets of the host loop + the kernel */
void zlo_moe_calc_reverse_idx(const int32_t* exp_ids, const int32_t* indices, const int32_t* all_loads, int num_experts, int world_size,
                              int sorted_by_rank, int32_t* expert_offset, int32_t* rev_indices, int64_t numel) {
    if (sorted_by_rank) {
        const int32_t* rank_loads = all_loads + num_experts;
        int rank_offset = 0;
        for (int rank = 0; rank < world_size; ++rank) {
            int offset = 0;
            for (int i = rank; i < num_experts; i += world_size) {
                expert_offset[i] = rank_offset + offset;
                offset += all_loads[i];
            }
            rank_offset += rank_loads[rank];
        }
    } else {
        int offset = 0;
        for (int i = 0; i < num_experts; ++i) {
            expert_offset[i] = offset;
            offset += all_loads[i];
        }
    }
    for (int64_t i = 0; i < numel; ++i) {
        const int idx = indices[i];
        rev_indices[idx] = (int32_t)i - expert_offset[exp_ids[idx]];
    }
}
/* fill_m_indices_padded_indices (:964-1057): returns the aligned total; padded_indices (sum of the local loads), m_indices (aligned total) */
int zlo_moe_fill_m_indices(const int32_t* all_loads, int block_m, int num_experts, int rank, int ws, int32_t* padded_indices, int32_t* m_indices) {
    int offset = 0, a_offset = 0;
    for (int i = 0, j = rank; j < num_experts; ++i, j += ws) {
        const int nt = all_loads[j], an = (nt + block_m - 1) / block_m * block_m;
        for (int s = 0; s < nt; ++s) padded_indices[offset + s] = a_offset + s;
        for (int s = 0; s < an; ++s) m_indices[a_offset + s] = i;
        offset += nt;
        a_offset += an;
    }
    return a_offset;
}

/* ------------------------------------------------------------------------------------------ */
/* f4: multi-head latent attention (DeepSeek MLA) over the latent cache, decode rows            */
/* ------------------------------------------------------------------------------------------ */
/* Attention::impl::MLAImpl over the compressed cache (src/nn/attention/multi_head_latent_attention.cpp:836-872, the open
 * "gemm + attn_softmax + gemm" route; the FlashMLA route :877-1004 is a closed binary): one latent row of kv_rank + rope_dim
 * values per key serves every head as key (all cache_dim values) and as value (its first kv_rank values):
 *   out[b, h, :kv_rank] = softmax_j( scale * q_adj[b, h, :] . kv_b[j, :] ) . kv_b[j, :kv_rank],   j < min(buf_len, valid_len)
 * flavour 0 = E (fp64).  flavour 1 = R, the rounding points of the open route: scores rounded to T (functions::Gemm output),
 * multiplied by T(scale) in T (fused_scale_mask_softmax, attention_softmax_kernel.cu:104-160), softmax in fp32, probabilities
 * rounded to T, second product accumulated in fp32, one rounding to T.  (The 1024-thread reduction trees of the softmax are not
 * restated: sequential fp32 sums; the difference is below the T rounding of the probabilities.) */
void zlo_mla_decode_attn(const uint16_t* q_adj, const int32_t* buf_lens, const int32_t* valid_lens, const uint16_t* const* kv_bufs,
                         uint16_t* out, int64_t b, int64_t h, int kv_rank, int rope_dim, float scale, int dtype, int flavour) {
    const int cd = kv_rank + rope_dim;
    for (int64_t t = 0; t < b; ++t) {
        const int len = valid_lens && valid_lens[t] < buf_lens[t] ? valid_lens[t] : buf_lens[t];
        const uint16_t* kv = kv_bufs[t];
        #pragma omp parallel for
        for (int64_t hh = 0; hh < h; ++hh) {
            const uint16_t* q = q_adj + (t * h + hh) * cd;
            uint16_t* o = out + (t * h + hh) * kv_rank;
            if (len <= 0) { for (int d = 0; d < kv_rank; ++d) o[d] = f2T(0.f, dtype); continue; }
            double* p = (double*)malloc(sizeof(double) * (size_t)len);
            double mx = -1e300;
            for (int j = 0; j < len; ++j) {
                double s = 0.0;
                for (int d = 0; d < cd; ++d) s += (double)T2f(q[d], dtype) * (double)T2f(kv[(int64_t)j * cd + d], dtype);
                if (flavour) {
                    const float sT = T2f(f2T((float)s, dtype), dtype);                       /* Gemm output in T */
                    s = (double)T2f(f2T(sT * T2f(f2T(scale, dtype), dtype), dtype), dtype);   /* x * T(scale) in T */
                } else s *= (double)scale;
                p[j] = s;
                if (s > mx) mx = s;
            }
            if (flavour) {
                float sum = 1e-20f;
                for (int j = 0; j < len; ++j) { const float e = expf((float)p[j] - (float)mx); p[j] = e; sum += e; }
                for (int j = 0; j < len; ++j) p[j] = (double)T2f(f2T((float)p[j] / sum, dtype), dtype);
                for (int d = 0; d < kv_rank; ++d) {
                    float acc = 0.f;
                    for (int j = 0; j < len; ++j) acc = fmaf((float)p[j], T2f(kv[(int64_t)j * cd + d], dtype), acc);
                    o[d] = f2T(acc, dtype);
                }
            } else {
                double sum = 0.0;
                for (int j = 0; j < len; ++j) { p[j] = exp(p[j] - mx); sum += p[j]; }
                for (int d = 0; d < kv_rank; ++d) {
                    double acc = 0.0;
                    for (int j = 0; j < len; ++j) acc += p[j] * (double)T2f(kv[(int64_t)j * cd + d], dtype);
                    o[d] = dtype ? zlo_f32_to_bf16((float)(acc / sum)) : zlo_f64_to_f16(acc / sum);
                }
            }
            free(p);
        }
    }
}

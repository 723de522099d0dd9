// zl_w4_exact.h -- the 8-weight word step of the bit-exact W4A16 arithmetic (KERNEL_gemm_warp_reduce's DEV_gemm_warp_reduce,
// src/nn/quant/gptq/q_gemm_k_major.cu:127-174), shared by the decode GEMV (w4_gemv.hip) and the fused MoE GEMVs (w4_moe.hip).
#pragma once
#include "zl_common.h"

namespace zlx {

typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ hv2 as_hv2(uint32_t u) { return __builtin_bit_cast(hv2, u); }

// ---- the 8-weight word step, hand-scheduled --------------------------------------------------
// The kernel is VALU-ISSUE bound (rocprofv3: ~115 VALU ops per 1 KiB of weights, SIMDs ~80 % busy
// issuing), so instruction count is the currency.  Two asm blocks per word:
//   DEQ (9 ops): the four (w & mask) | 0x6400 extractions as fused v_and_or_b32 (gfx950 VOP3 takes no
//        32-bit literals, so hipcc splits the C expression into v_and + v_or; with the mask in an
//        SGPR and the magic in a VGPR the fused form encodes), then the exact fp16 (q - z).
//   DOT (6 ops per activation row): the reference's hfma2 chain, f32(lo) + f32(hi) as ONE
//        v_fma_mix_f32 (f16 operands widened exactly, one fp32 rounding == __half2float(lo) +
//        __half2float(hi)), and acc = fma(dot, scale, acc) as v_fma_mix_f32 reading the f16 scale.
// Everything inside is plain dependent VALU (hardware-interlocked, no wait states needed).
struct DeqWord {
    uint32_t d0, d1, d2, d3;  // half2 (q - z) for weight pairs (0,1) (2,3) (4,5) (6,7)
};

__device__ __forceinline__ DeqWord deq_word(uint32_t w, uint32_t z1, uint32_t z16, uint32_t mask_lo,
                                            uint32_t mask_hi, uint32_t magic, uint32_t one16) {
    DeqWord d;
    uint32_t wb;
    asm("v_lshrrev_b32 %4, 8, %5\n\t"
        "v_and_or_b32 %0, %5, %8, %10\n\t"
        "v_and_or_b32 %1, %5, %9, %10\n\t"
        "v_and_or_b32 %2, %4, %8, %10\n\t"
        "v_and_or_b32 %3, %4, %9, %10\n\t"
        "v_pk_add_f16 %0, %0, %6\n\t"
        "v_pk_fma_f16 %1, %1, %11, %7\n\t"
        "v_pk_add_f16 %2, %2, %6\n\t"
        "v_pk_fma_f16 %3, %3, %11, %7"
        : "=&v"(d.d0), "=&v"(d.d1), "=&v"(d.d2), "=&v"(d.d3), "=&v"(wb)
        : "v"(w), "v"(z1), "v"(z16), "s"(mask_lo), "s"(mask_hi), "v"(magic), "v"(one16));
    return d;
}

// SCALE_HI selects which half of `scale2` (two packed fp16 scales) multiplies the dot
template <bool SCALE_HI>
__device__ __forceinline__ float dot_word(const DeqWord& d, const uint4& xa, uint32_t scale2, float acc) {
    uint32_t r;
    float t;
    if constexpr (SCALE_HI) {
        asm("v_pk_fma_f16 %0, %3, %7, 0\n\t"
            "v_pk_fma_f16 %0, %4, %8, %0\n\t"
            "v_pk_fma_f16 %0, %5, %9, %0\n\t"
            "v_pk_fma_f16 %0, %6, %10, %0\n\t"
            "v_fma_mix_f32 %1, %0, 1.0, %0 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
            "v_fma_mix_f32 %2, %1, %11, %2 op_sel:[0,1,0] op_sel_hi:[0,1,0]"
            : "=&v"(r), "=&v"(t), "+v"(acc)
            : "v"(d.d0), "v"(d.d1), "v"(d.d2), "v"(d.d3), "v"(xa.x), "v"(xa.y), "v"(xa.z), "v"(xa.w), "v"(scale2));
    } else {
        asm("v_pk_fma_f16 %0, %3, %7, 0\n\t"
            "v_pk_fma_f16 %0, %4, %8, %0\n\t"
            "v_pk_fma_f16 %0, %5, %9, %0\n\t"
            "v_pk_fma_f16 %0, %6, %10, %0\n\t"
            "v_fma_mix_f32 %1, %0, 1.0, %0 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
            "v_fma_mix_f32 %2, %1, %11, %2 op_sel:[0,0,0] op_sel_hi:[0,1,0]"
            : "=&v"(r), "=&v"(t), "+v"(acc)
            : "v"(d.d0), "v"(d.d1), "v"(d.d2), "v"(d.d3), "v"(xa.x), "v"(xa.y), "v"(xa.z), "v"(xa.w), "v"(scale2));
    }
    return acc;
}

}  // namespace zlx

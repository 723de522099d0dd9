// w4_i8p_common.h -- what the integer-plane W4A16 kernels share (w4_i8p.hip: one launch per projection, register ring;
// w4_engine.hip: LDS-DMA loader wave + consumer waves, fused launches): the launch parameters and the wave64 DPP helpers.
#pragma once
#include "zl_common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

struct I8Params {
    const uint16_t* x;
    int64_t ldx;
    const uint4* qw;
    const uint32_t* meta;
    uint32_t qw_bytes, meta_bytes;
    const uint16_t* bias;
    const uint16_t* residual;
    uint16_t* y;
    int m, n, k;
    int groups;        // 128-k items per row tile
    int tiles;         // 16-row tiles
    int epi, ld_out;
    const uint16_t* norm_w;
    float norm_eps;
    // ROPE instantiations (fused qkv projection of a decode step)
    const float* cosv;
    const float* sinv;
    const int32_t* placement;
    const int32_t* buf_lens;
    uint16_t* const* k_bufs;
    uint16_t* const* v_bufs;
    uint16_t* q_out;
    int h, hkv, d, bshd;
    int pair_stride;
    // MERGE instantiations (attn_out projection of a decode step): the activation rows are merged from the decode attention's
    // half-precision split partials (zl_decode_attn_splits_h: fp16 [row][head][split][128], then fp32 (max, sum) pairs)
    const uint16_t* mg_part;
    const float* mg_stat;
    const int32_t* mg_valid_lens;   // with buf_lens: keys per task -> live splits
    int mg_split_len, mg_max_splits;
};

// ---- DPP helpers (wave64, rows of 16 lanes) ----------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// all-reduce inside each row of 16 lanes by rotations (row_ror:8,4,2,1)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0x128>(0.f, v);
    v += dpp_f<0x124>(0.f, v);
    v += dpp_f<0x122>(0.f, v);
    v += dpp_f<0x121>(0.f, v);
    return v;
}
__device__ __forceinline__ int row16_sum(int v) {
    v += dpp_i<0x128>(0, v);
    v += dpp_i<0x124>(0, v);
    v += dpp_i<0x122>(0, v);
    v += dpp_i<0x121>(0, v);
    return v;
}
__device__ __forceinline__ int row16_max(int v) {
    v = max(v, dpp_i<0x128>(0, v));
    v = max(v, dpp_i<0x124>(0, v));
    v = max(v, dpp_i<0x122>(0, v));
    v = max(v, dpp_i<0x121>(0, v));
    return v;
}
// sum over the 64 lanes, valid in lanes 48..63 (row_bcast15 into rows 1 / 3, row_bcast31 into rows 2 / 3)
__device__ __forceinline__ float wave_sum_hi(float v) {
    v = row16_sum(v);
    v += dpp_f<0x142, 0xA>(0.f, v);
    v += dpp_f<0x143, 0xC>(0.f, v);
    return v;
}

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + expf(-x)); }

}  // namespace

// kv_quant.hip -- the INT8 KV cache write side (quantised variant; 8f rank 3).
//
// Reference: with a quantised cache every new K/V row (one kv head, D values) is stored as D unsigned
// codes and one fp32 scale:   code = 128 + rint(x * 127 / amax),  scale = amax / 127
//   int8_op::quant_calc_scale(ctx, x, 127, 128)      src/nn/quant/int8/quant_kernel.cu:15-47, :49-80
//   decode:  attention.cpp:656-676  (quantise h_k / h_v, copy_to_rag_buffer2 for codes AND scales)
//   prefill: TransformerBuffer::copy, src/kvcache/transformer_buffer.cu:128-134
// Per task the codes live in (len_buf, Hkv, D) [BSHD] or (Hkv, len_buf, D) u8 buffers and the scales in
// (len_buf, Hkv) / (Hkv, len_buf) fp32 buffers, reached through device pointer arrays like the fp16 cache.
//
// Here the quantisation and both scatters are one launch (three in the reference, six with the scales), and
// for the decode step the rotation of q and k is fused in front (rope_qk_cache + 2 x quant_calc_scale +
// 2 x copy_to_rag_buffer2 -> one launch).  Roundings are the reference's: the rotated k is rounded to T
// first, amax / codes are computed from the rounded values, rint is round-half-even.
// A row of zeros gets codes 128 and scale 0 (the reference computes 127/0 there; no caller relies on it).
#include "zl_common.h"

namespace {

__device__ __forceinline__ float rope1(float a, float partner, float c, float s, bool minus) {
    return minus ? __builtin_fmaf(-partner, s, a * c) : __builtin_fmaf(partner, s, a * c);
}

// one workgroup (256) per row, any k
template <int DT>
__global__ __launch_bounds__(256) void k_quant_rows_zp(const uint16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                       float* __restrict__ scale, int k, float q_zero) {
    __shared__ float red[16];
    const size_t off = (size_t)blockIdx.x * k;
    float amax = 0.f;
    for (int i = threadIdx.x; i < k; i += 256) amax = fmaxf(amax, fabsf(ZT<DT>::to_f32(x[off + i])));
    amax = zl_block_max(amax, red);
    const float bs = amax > 0.f ? 127.f / amax : 0.f;
    for (int i = threadIdx.x; i < k; i += 256)
        q[off + i] = (uint8_t)(q_zero + __builtin_rintf(ZT<DT>::to_f32(x[off + i]) * bs));
    if (threadIdx.x == 0) scale[blockIdx.x] = amax / 127.f;
}

// rows of g codes back to T: out = rn_T((code - q_zero) * scale[row]) -- int8_op::dequant_group (quant_reduce_kernel.cu:144-175), the
// fp32 product rounded once.  q_zero != 0: the codes are unsigned (the INT8 KV cache, q_zero 128).  grid (m), block 256
template <int DT>
__global__ __launch_bounds__(256) void k_dequant_group(const uint8_t* __restrict__ q, const float* __restrict__ scale, uint16_t* __restrict__ out, int g,
                                                       float q_zero) {
    const size_t off = (size_t)blockIdx.x * g;
    const float sc = scale[blockIdx.x];
    for (int i = threadIdx.x; i < g; i += 256) {
        const float v = q_zero != 0.f ? (float)q[off + i] : (float)(int8_t)q[off + i];
        out[off + i] = ZT<DT>::from_f32((v - q_zero) * sc);
    }
}

struct KvQuantParams {
    const float* cosv;          // (tokens, D) or null (no rotation: rows are final)
    const float* sinv;
    const uint16_t* qkv;        // fused rows (tokens, (H + 2 Hkv) D)           [fused form]
    const uint16_t* k_src;      // (tokens, Hkv, D)                              [plain form]
    const uint16_t* v_src;
    uint16_t* q_out;            // (tokens, H, D) rotated q                      [fused form]
    const int32_t* placement;   // (tokens) slot in the task's buffers, < 0: skip
    const int32_t* buf_lens;    // (B)
    uint8_t* const* k_bufs;
    uint8_t* const* v_bufs;
    float* const* k_scales;
    float* const* v_scales;
    int len_q, h, hkv, d, neox, bshd;
};

// grid (tokens, heads), block D (one thread per head-dim element; D <= 256: at most 4 waves).
// FUSED: heads = H + 2 Hkv over the fused qkv row; otherwise heads = 2 Hkv over k_src / v_src.
template <int DT, bool FUSED>
__global__ void k_kv_quant_store(const KvQuantParams p) {
    __shared__ float red[16];
    const int t = blockIdx.x, head = blockIdx.y, col = threadIdx.x, d = p.d;
    const int b = t / p.len_q;
    const int place = p.placement[t];
    const int len_buf = p.buf_lens[b];
    int hk;
    bool is_v;
    const uint16_t* src;
    if constexpr (FUSED) {
        src = p.qkv + ((size_t)t * (p.h + 2 * p.hkv) + head) * d;
        is_v = head >= p.h + p.hkv;
        hk = head - p.h - (is_v ? p.hkv : 0);
    } else {
        is_v = head >= p.hkv;
        hk = is_v ? head - p.hkv : head;
        src = (is_v ? p.v_src : p.k_src) + ((size_t)t * p.hkv + hk) * d;
    }
    float val = ZT<DT>::to_f32(src[col]);
    if (FUSED && !is_v) {
        const int half = d / 2;
        const float c = p.cosv[(size_t)t * d + col], s = p.sinv[(size_t)t * d + col];
        float r;
        if (p.neox)
            r = col < half ? rope1(val, ZT<DT>::to_f32(src[col + half]), c, s, true)
                           : rope1(val, ZT<DT>::to_f32(src[col - half]), c, s, false);
        else
            r = (col & 1) == 0 ? rope1(val, ZT<DT>::to_f32(src[col + 1]), c, s, true)
                               : rope1(val, ZT<DT>::to_f32(src[col - 1]), c, s, false);
        const uint16_t o = ZT<DT>::from_f32(r);
        if (head < p.h) {                                   // block-uniform
            p.q_out[((size_t)t * p.h + head) * d + col] = o;
            return;
        }
        val = ZT<DT>::to_f32(o);                            // the cache sees the rounded row
    }
    if (place < 0 || place >= len_buf) return;              // block-uniform; a slot outside the buffer is never written
    const float amax = zl_block_max(fabsf(val), red);
    const float bs = amax > 0.f ? 127.f / amax : 0.f;
    const uint8_t code = (uint8_t)(128.f + __builtin_rintf(val * bs));
    const size_t row = p.bshd ? (size_t)place * p.hkv + hk : (size_t)hk * len_buf + place;
    (is_v ? p.v_bufs : p.k_bufs)[b][row * d + col] = code;
    if (col == 0) (is_v ? p.v_scales : p.k_scales)[b][row] = amax / 127.f;
}

}  // namespace

#define ZL_DT_SWITCH(dtype, EXPR_F16, EXPR_BF16) \
    if ((dtype) == ZL_F16) { EXPR_F16; } else if ((dtype) == ZL_BF16) { EXPR_BF16; } else return ZL_EDTYPE;

extern "C" {

int zl_quant_calc_scale_zp(const uint16_t* x, uint8_t* q, float* scale, int64_t m, int64_t k, int q_zero, int dtype,
                           zl_stream_t s) {
    ZL_CHECK_ARG(x && q && scale && m > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(q_zero >= 0 && q_zero <= 128, ZL_EINVAL);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_quant_rows_zp<ZL_F16>, dim3((unsigned)m), dim3(256), 0, (hipStream_t)s, x, q, scale, (int)k, (float)q_zero),
        hipLaunchKernelGGL(k_quant_rows_zp<ZL_BF16>, dim3((unsigned)m), dim3(256), 0, (hipStream_t)s, x, q, scale, (int)k, (float)q_zero))
    return zl_launch_status();
}

int zl_dequant_group(const void* q, const float* scale, uint16_t* out, int64_t m, int64_t g, int q_zero, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(q && scale && out && m > 0 && g > 0, ZL_EINVAL);
    ZL_CHECK_ARG(q_zero >= 0 && q_zero <= 128 && m < ((int64_t)1 << 31), ZL_EINVAL);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_dequant_group<ZL_F16>, dim3((unsigned)m), dim3(256), 0, (hipStream_t)s, (const uint8_t*)q, scale, out, (int)g, (float)q_zero),
        hipLaunchKernelGGL(k_dequant_group<ZL_BF16>, dim3((unsigned)m), dim3(256), 0, (hipStream_t)s, (const uint8_t*)q, scale, out, (int)g, (float)q_zero))
    return zl_launch_status();
}

int zl_quant_copy_to_rag_buffer(const int32_t* placement, const int32_t* buf_lens, const uint16_t* k_src,
                                const uint16_t* v_src, uint8_t* const* k_bufs, uint8_t* const* v_bufs,
                                float* const* k_scales, float* const* v_scales, int64_t b, int64_t len_q, int64_t hkv,
                                int64_t d, int bshd, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(placement && buf_lens && k_src && v_src && k_bufs && v_bufs && k_scales && v_scales, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && len_q > 0 && hkv > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 256 && d % 64 == 0 && 2 * hkv <= 65535, ZL_ESHAPE);
    KvQuantParams p;
    p.cosv = p.sinv = nullptr; p.qkv = nullptr; p.k_src = k_src; p.v_src = v_src; p.q_out = nullptr;
    p.placement = placement; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs; p.k_scales = k_scales;
    p.v_scales = v_scales; p.len_q = (int)len_q; p.h = 0; p.hkv = (int)hkv; p.d = (int)d; p.neox = 1; p.bshd = bshd;
    const dim3 grid((unsigned)(b * len_q), (unsigned)(2 * hkv));
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL((k_kv_quant_store<ZL_F16, false>), grid, dim3((unsigned)d), 0, (hipStream_t)s, p),
        hipLaunchKernelGGL((k_kv_quant_store<ZL_BF16, false>), grid, dim3((unsigned)d), 0, (hipStream_t)s, p))
    return zl_launch_status();
}

int zl_rope_quant_scatter_decode(const float* cosv, const float* sinv, const uint16_t* qkv, uint16_t* q,
                                 const int32_t* placement, const int32_t* buf_lens, uint8_t* const* k_bufs,
                                 uint8_t* const* v_bufs, float* const* k_scales, float* const* v_scales, int64_t b,
                                 int64_t h, int64_t hkv, int64_t d, int neox, int bshd, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(cosv && sinv && qkv && q && placement && buf_lens && k_bufs && v_bufs && k_scales && v_scales, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && h > 0 && hkv > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 256 && d % 64 == 0 && h + 2 * hkv <= 65535, ZL_ESHAPE);
    KvQuantParams p;
    p.cosv = cosv; p.sinv = sinv; p.qkv = qkv; p.k_src = p.v_src = nullptr; p.q_out = q;
    p.placement = placement; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs; p.k_scales = k_scales;
    p.v_scales = v_scales; p.len_q = 1; p.h = (int)h; p.hkv = (int)hkv; p.d = (int)d; p.neox = neox; p.bshd = bshd;
    const dim3 grid((unsigned)b, (unsigned)(h + 2 * hkv));
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL((k_kv_quant_store<ZL_F16, true>), grid, dim3((unsigned)d), 0, (hipStream_t)s, p),
        hipLaunchKernelGGL((k_kv_quant_store<ZL_BF16, true>), grid, dim3((unsigned)d), 0, (hipStream_t)s, p))
    return zl_launch_status();
}

}  // extern "C"

// zl_stage.h -- stage MT activation rows (K elements, padded to kp with zeros) into LDS, optionally
// RMS-normalising them on the way (LayerNorm::forward semantics, src/nn/layernorm/layernorm.cu:10-42:
// y = T(f32(x) * rsqrt(mean(x^2)+eps) * f32(w)), block-wide fp32 sum of squares).
#pragma once
#include "zl_common.h"

template <int DT, int MT, int THREADS>
__device__ __forceinline__ void zl_stage_rows(const uint16_t* __restrict__ x, int64_t ldx, int m0, int m_total, int k,
                                              int kp, const uint16_t* __restrict__ norm_w, float eps,
                                              uint16_t* xs, float* red) {
#pragma unroll 1
    for (int m = 0; m < MT; ++m) {
        const bool live = (m0 + m) < m_total;
        const uint16_t* xrow = x + (size_t)(m0 + m) * ldx;
        uint16_t* xd = xs + (size_t)m * kp;
        float ss = 0.f;
        for (int i = threadIdx.x * 8; i < kp; i += THREADS * 8) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (live && i < k) v = *reinterpret_cast<const uint4*>(xrow + i);
            if (norm_w) {
                const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = ZT<DT>::to_f32((uint16_t)(u[e] & 0xffffu)), b = ZT<DT>::to_f32((uint16_t)(u[e] >> 16));
                    ss = __builtin_fmaf(a, a, ss);
                    ss = __builtin_fmaf(b, b, ss);
                }
            }
            *reinterpret_cast<uint4*>(xd + i) = v;
        }
        if (norm_w) {
            ss = zl_block_sum(ss, red);
            const float rs = zl_rsqrt_rn(ss / (float)k + eps);
            for (int i = threadIdx.x * 8; i < k; i += THREADS * 8) {
                uint4 v = *reinterpret_cast<uint4*>(xd + i);
                uint4 wv = *reinterpret_cast<const uint4*>(norm_w + i);
                uint32_t u[4] = {v.x, v.y, v.z, v.w};
                const uint32_t wu[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = ZT<DT>::to_f32((uint16_t)(u[e] & 0xffffu)) * rs * ZT<DT>::to_f32((uint16_t)(wu[e] & 0xffffu));
                    float b = ZT<DT>::to_f32((uint16_t)(u[e] >> 16)) * rs * ZT<DT>::to_f32((uint16_t)(wu[e] >> 16));
                    u[e] = (uint32_t)ZT<DT>::from_f32(a) | ((uint32_t)ZT<DT>::from_f32(b) << 16);
                }
                *reinterpret_cast<uint4*>(xd + i) = make_uint4(u[0], u[1], u[2], u[3]);
            }
        }
    }
}

// fp8_ops.hip -- the FP8-activation branch of W4A8: gptq_gemm_k_major with W4_FP8_ALGO (src/nn/quant/gptq/q_gemm_k_major.cu:
// 1003-1035): per-tensor scales (nn::fp8::calc_scale, src/nn/quant/fp8/fp8_util.cu:100-195), activations and dequantised weights
// cast to OCP E4M3 (dynamic_scaled_quant :197-229, T_KERNEL_cvt_half_fp8 :56-78; KERNEL_dequant<half, 2>,
// q_gemm_k_major.cu:843-906 + Int4GPTQ::calc_w4a8_scale, src/nn/linear/linear.cpp:1124-1129), and an fp8 x fp8 GEMM with fp32
// accumulation scaled by scale_a * scale_b (functions::Gemm kFP8_E4M3 = cuBLASLt in the reference; v_mfma_f32_16x16x32_fp8_fp8
// here).  The casts are written out in integer arithmetic (cvt.rn.satfinite.e4m3x2.f16x2: round to nearest even on the E4M3FN
// grid, saturate at 448) instead of relying on the conversion instructions' overflow mode; codes bit-exact against the oracle.
#include "zl_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint8_t f32_to_e4m3(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    const uint32_t au = u & 0x7fffffffu;
    if (au > 0x7f800000u) return (uint8_t)(sign | 0x7fu);
    const float a = __builtin_bit_cast(float, au);
    if (a >= 464.0f) return (uint8_t)(sign | 0x7eu);
    if (a < 0.015625f) return (uint8_t)(sign | (uint8_t)nearbyintf(a * 512.0f));
    const uint32_t r = au + 0x7ffffu + ((au >> 20) & 1u);
    uint32_t code = (((r >> 23) - 120u) << 3) | ((r >> 20) & 7u);
    if (code > 0x7eu) code = 0x7eu;
    return (uint8_t)(sign | code);
}

__global__ void k_fp8_zero(float* scale) { *scale = 0.f; }

// segmented_max_reduction (fp8_util.cu:100-140): every workgroup's maximum / MAX goes through an atomic maximum on the bit
// pattern (non-negative floats order like integers); IEEE division is monotonic, so the result is max|x| / MAX exactly
template <int DT>
__global__ __launch_bounds__(1024) void k_fp8_amax(const uint16_t* __restrict__ x, int64_t numel, float max_e4m3, float* scale) {
    __shared__ float red[16];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 1024) m = fmaxf(m, fabsf(ZT<DT>::to_f32(x[i])));
    m = zl_block_max(m, red);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(scale), __builtin_bit_cast(unsigned int, m / max_e4m3));
}

template <int DT>
__global__ __launch_bounds__(256) void k_fp8_cvt(const uint16_t* __restrict__ x, const float* __restrict__ scale_ptr,
                                                  uint8_t* __restrict__ out, int64_t numel) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= numel) return;
    const float inv = 1.f / *scale_ptr;
    _Float16 h;
    if (DT == ZL_F16) {
        float p = (float)__builtin_bit_cast(_Float16, x[i]) * (float)zl_f32_to_f16(inv);   // __hmul2: exact product, ...
        h = zl_f32_to_f16(p);                                                               // ... one rounding to fp16
    } else {
        h = zl_f32_to_f16(inv * ZT<ZL_BF16>::to_f32(x[i]));
    }
    out[i] = f32_to_e4m3((float)h);
}

// C[m, n] = T(scale_a * scale_b * sum_k A[m, k] B[n, k]), A / B E4M3 codes, K % 64 == 0.  Lane (row = lane & 15, kq = lane >> 4)
// loads 16 consecutive k of its row per 64-k chunk; MFMA 0 takes the low 8 bytes of every lane, MFMA 1 the high 8 (A and B
// alike, so the k assignment is consistent).  One wave per 16 weight rows x MT*16 activation rows.  Correctness first.
template <int MT>
__global__ __launch_bounds__(256) void k_fp8_gemm_nt(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                     const float* __restrict__ scale_a, const float* __restrict__ scale_b,
                                                     uint16_t* __restrict__ c, int m, int n, int k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= n) return;
    const int m0 = blockIdx.y * (MT * 16);
    const int col = lane & 15, kq = lane >> 4;
    f4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
    const bool ncol_ok = (n0 + col) < n;
    const uint8_t* brow = b + (size_t)(n0 + col) * k;
    for (int k0 = 0; k0 < k; k0 += 64) {
        const int kk = k0 + 16 * kq;
        uint4 bf = make_uint4(0, 0, 0, 0);
        if (ncol_ok) bf = zl_load_nt(reinterpret_cast<const uint4*>(brow + kk));
        const long b_lo = (long)(((unsigned long long)bf.y << 32) | bf.x), b_hi = (long)(((unsigned long long)bf.w << 32) | bf.z);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int row = m0 + t * 16 + col;
            uint4 af = make_uint4(0, 0, 0, 0);
            if (row < m) af = *reinterpret_cast<const uint4*>(a + (size_t)row * k + kk);
            const long a_lo = (long)(((unsigned long long)af.y << 32) | af.x), a_hi = (long)(((unsigned long long)af.w << 32) | af.z);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_lo, b_lo, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_hi, b_hi, acc[t], 0, 0, 0);
        }
    }
    const float sc = *scale_a * *scale_b;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + t * 16 + 4 * kq + i;
            if (row < m && ncol_ok) c[(size_t)row * n + n0 + col] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(acc[t][i] * sc));
        }
}

}  // namespace

extern "C" {

int zl_fp8_calc_scale(const uint16_t* x, int64_t numel, float max_e4m3, float* scale, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && scale && numel > 0 && max_e4m3 > 0.f, ZL_EINVAL);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    hipStream_t hs = (hipStream_t)s;
    hipLaunchKernelGGL(k_fp8_zero, dim3(1), dim3(1), 0, hs, scale);
    const int64_t blocks = (numel + 1024 * 8 - 1) / (1024 * 8);
    const dim3 grid((unsigned)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks)));
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_fp8_amax<ZL_F16>, grid, dim3(1024), 0, hs, x, numel, max_e4m3, scale);
    else hipLaunchKernelGGL(k_fp8_amax<ZL_BF16>, grid, dim3(1024), 0, hs, x, numel, max_e4m3, scale);
    return zl_launch_status();
}

int zl_fp8_cvt_half(const uint16_t* x, const float* scale, uint8_t* out, int64_t numel, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && scale && out && numel > 0, ZL_EINVAL);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    const dim3 grid((unsigned)((numel + 255) / 256));
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_fp8_cvt<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, x, scale, out, numel);
    else hipLaunchKernelGGL(k_fp8_cvt<ZL_BF16>, grid, dim3(256), 0, (hipStream_t)s, x, scale, out, numel);
    return zl_launch_status();
}

int zl_fp8_gemm_nt(const uint8_t* a, const uint8_t* b, const float* scale_a, const float* scale_b, uint16_t* out, int64_t m,
                   int64_t n, int64_t k, zl_stream_t s) {
    ZL_CHECK_ARG(a && b && scale_a && scale_b && out && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 64 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0, ZL_ESHAPE);
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((m + 63) / 64));
    hipLaunchKernelGGL(k_fp8_gemm_nt<4>, grid, dim3(256), 0, (hipStream_t)s, a, b, scale_a, scale_b, out, (int)m, (int)n, (int)k);
    return zl_launch_status();
}

}  // extern "C"

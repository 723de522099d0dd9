// zl_common.h -- shared device/host helpers for the gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/zhilight_amd.h"

#define ZL_WAVE 64

#define ZL_CHECK_ARG(cond, code) \
    do {                         \
        if (!(cond)) return (code); \
    } while (0)

// status of the launch that was just enqueued (hipGetLastError is per-thread, no sync)
static inline int zl_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ZL_OK : (int)e;
}

typedef _Float16 h16;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// streamed-once data (weights): non-temporal 16-byte load, keeps L2/MALL for activations and KV
__device__ __forceinline__ uint4 zl_load_nt(const uint4* p) {
    u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 zl_load_nt(const uint2* p) {
    u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint16_t zl_load_nt(const uint16_t* p) { return __builtin_nontemporal_load(p); }

// fp32 -> fp16 with the fp32 value MATERIALISED first.  Without the barrier LLVM folds
// fptrunc(fmul(a, fpext(b))) into v_fma_mixlo_f16, i.e. rounds the exact product once to fp16 and
// skips the fp32 rounding the reference (and the oracle) perform -- a real 1-ulp difference on
// near-ties (seen on quant_scale_back).  The empty asm costs no instruction.
__device__ __forceinline__ _Float16 zl_f32_to_f16(float f) {
    asm volatile("" : "+v"(f));
    return (_Float16)f;
}

// ---- scalar type helpers: activations travel as raw 16-bit patterns ----
template <int DT> struct ZT;
template <> struct ZT<ZL_F16> {
    static __device__ __forceinline__ float to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, zl_f32_to_f16(f)); }
};
template <> struct ZT<ZL_BF16> {
    static __device__ __forceinline__ float to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
    static __device__ __forceinline__ uint16_t from_f32(float f) {  // round-to-nearest-even
        uint32_t u = __builtin_bit_cast(uint32_t, f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
};

__device__ __forceinline__ float zl_wave_sum(float v) {  // 64-lane butterfly, result in every lane
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float zl_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (<= 16 waves); `red` is >= 16 floats of LDS; all threads
// get the result.  Order: lanes butterfly, then waves 0..nw-1 sequentially.
__device__ __forceinline__ float zl_block_sum(float v, float* red) {
    v = zl_wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float zl_block_max(float v, float* red) {
    v = zl_wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// correctly rounded 1/sqrt (the oracle's 1.0f/sqrtf): fp32 sqrt and divide are IEEE under hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt
__device__ __forceinline__ float zl_rsqrt_rn(float x) { return 1.0f / sqrtf(x); }

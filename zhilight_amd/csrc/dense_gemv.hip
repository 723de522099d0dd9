// dense_gemv.hip -- y = T(alpha * x . W^T + bias) for small M (decode lm_head / NormalLinear), W (N,K)
// fp16 or bf16 row-major.  SURVEY 8a row a21 (the reference calls cuBLASLt with fp32 compute:
// bm/functions/gemm.cpp:258-344, src/nn/embedding/embedding.cu:274-289).
//
// HBM-bound weight streaming (1.05 GB for the Llama-3 lm_head): one wavefront per output row, each
// lane loads 16 B (8 elements) per step -> 1 KiB fully coalesced non-temporal loads, kRing of them
// in flight per wave; activations staged once per workgroup in LDS (optionally RMS-normalised:
// the model's final norm fuses here); fp32 accumulation (v_dot2_f32_f16 for fp16), 64-lane
// butterfly per row, no atomics / split-K.
#include "zl_common.h"
#include "zl_stage.h"

namespace {

constexpr int kRing = 8;

struct DenseParams {
    const uint16_t* x;
    int64_t ldx;
    const uint16_t* w;
    const uint16_t* bias;
    uint16_t* y;
    const uint16_t* norm_w;
    float norm_eps, alpha;
    int m, n, k, kp;       // kp = k rounded up to 512 (one wave-load)
    int loads_per_row;     // kp / 512
    int rows_per_wave;
};

typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ float dot8(uint4 a, uint4 b, float acc) {
    const uint32_t au[4] = {a.x, a.y, a.z, a.w}, bu[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (DT == ZL_F16) {
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(hv2, au[e]), __builtin_bit_cast(hv2, bu[e]), acc, false);
        } else {
            acc = __builtin_fmaf(__builtin_bit_cast(float, au[e] << 16), __builtin_bit_cast(float, bu[e] << 16), acc);
            acc = __builtin_fmaf(__builtin_bit_cast(float, au[e] & 0xffff0000u),
                                 __builtin_bit_cast(float, bu[e] & 0xffff0000u), acc);
        }
    }
    return acc;
}

template <int DT, int MT>
__global__ __launch_bounds__(256) void k_dense_gemv(const DenseParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem);
    float* red = reinterpret_cast<float*>(smem + (size_t)MT * p.kp * 2);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.y * MT;
    const int L = p.loads_per_row;
    const int gw = blockIdx.x * 4 + wave;
    const int row0 = gw * p.rows_per_wave;
    int nrows = p.n - row0;
    nrows = nrows < 0 ? 0 : (nrows > p.rows_per_wave ? p.rows_per_wave : nrows);
    const int total = nrows * L;

    uint4 wq[kRing];
    int iss_row = row0, iss_l = 0;
    auto issue = [&](int slot) {
        const int col = iss_l * 512 + lane * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (col < p.k) v = zl_load_nt(reinterpret_cast<const uint4*>(p.w + (size_t)iss_row * p.k + col));
        wq[slot] = v;
        if (++iss_l == L) {
            iss_l = 0;
            ++iss_row;
        }
    };
#pragma unroll
    for (int s = 0; s < kRing; ++s)
        if (s < total) issue(s);

    zl_stage_rows<DT, MT, 256>(p.x, p.ldx, m0, p.m, p.k, p.kp, p.norm_w, p.norm_eps, xs, red);
    __syncthreads();

    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    int cl = 0, crow = row0;
    auto consume = [&](int slot) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            uint4 xa = *reinterpret_cast<const uint4*>(xs + (size_t)m * p.kp + cl * 512 + lane * 8);
            acc[m] = dot8<DT>(wq[slot], xa, acc[m]);
        }
        if (++cl == L) {
            cl = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v = zl_wave_sum(acc[m]);
                acc[m] = 0.f;
                if (lane == 0 && (m0 + m) < p.m) {
                    float b = p.bias ? ZT<DT>::to_f32(p.bias[crow]) : 0.f;
                    p.y[(size_t)(m0 + m) * p.n + crow] = ZT<DT>::from_f32(p.alpha * v + b);
                }
            }
            ++crow;
        }
    };
#pragma unroll 1
    for (int it = 0; it < total; it += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            if (it + s < total) {
                consume(s);
                if (it + s + kRing < total) issue(s);
            }
        }
    }
}

template <int DT, int MT>
int launch(const DenseParams& p, int gx, int gy, hipStream_t st) {
    size_t lds = (size_t)MT * p.kp * 2 + 64;
    if (lds > 160 * 1024) return ZL_ELIMIT;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_gemv<DT, MT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((k_dense_gemv<DT, MT>), dim3(gx, gy), dim3(256), lds, st, p);
    return zl_launch_status();
}

}  // namespace

extern "C" int zl_gemm_nt_small_m(const uint16_t* x, int64_t ldx, const uint16_t* w, const uint16_t* bias, uint16_t* y,
                                  int64_t m, int64_t n, int64_t k, float alpha, int dtype,
                                  const uint16_t* norm_weight, float norm_eps, zl_stream_t s) {
    ZL_CHECK_ARG(x && w && y && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 8 == 0 && ldx % 8 == 0 && ldx >= k, ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    DenseParams p;
    p.x = x; p.ldx = ldx; p.w = w; p.bias = bias; p.y = y;
    p.norm_w = norm_weight; p.norm_eps = norm_eps; p.alpha = alpha;
    p.m = (int)m; p.n = (int)n; p.k = (int)k;
    p.kp = (int)((k + 511) / 512 * 512);
    p.loads_per_row = p.kp / 512;
    int mt = m >= 4 ? 4 : (int)m;
    while (mt > 1 && (size_t)mt * p.kp * 2 + 64 > 64 * 1024) --mt;
    if (mt == 3) mt = 2;
    const int gy = (int)((m + mt - 1) / mt);
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    const int waves = cus * 8;  // two 4-wave workgroups per CU
    p.rows_per_wave = (int)((n + waves - 1) / waves);
    const int gx = (int)(((n + p.rows_per_wave - 1) / p.rows_per_wave + 3) / 4);
    hipStream_t hs = (hipStream_t)s;
#define ZL_DISPATCH(DT)                                      \
    switch (mt) {                                            \
        case 1: return launch<DT, 1>(p, gx, gy, hs);         \
        case 2: return launch<DT, 2>(p, gx, gy, hs);         \
        default: return launch<DT, 4>(p, gx, gy, hs);        \
    }
    if (dtype == ZL_F16) { ZL_DISPATCH(ZL_F16) }
    ZL_DISPATCH(ZL_BF16)
#undef ZL_DISPATCH
}

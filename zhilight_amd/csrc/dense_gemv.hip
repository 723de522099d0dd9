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
    float2* argmax_ws;     // optional: per (activation row, global wave) best (value, row index) of the wave
    int total_waves;
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
    float best_v[MT];
    int best_i[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc[m] = 0.f;
        best_v[m] = -INFINITY;
        best_i[m] = 0x7fffffff;
    }
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
                    const uint16_t y16 = ZT<DT>::from_f32(p.alpha * v + b);
                    p.y[(size_t)(m0 + m) * p.n + crow] = y16;
                    // greedy pick on the ROUNDED logit, first index on ties (rows ascend inside a wave)
                    const float yv = ZT<DT>::to_f32(y16);
                    if (yv > best_v[m] || best_i[m] == 0x7fffffff) {
                        best_v[m] = yv;
                        best_i[m] = crow;
                    }
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
    if (p.argmax_ws && lane == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if ((m0 + m) < p.m)
                p.argmax_ws[(size_t)(m0 + m) * p.total_waves + gw] = make_float2(best_v[m], __int_as_float(best_i[m]));
    }
}

// Greedy next token from the per-wave candidates + the per-step bookkeeping of a decode batch in ONE small
// launch (what fill_search_tokens does on the host in the reference between steps,
// src/generator/batch_generator.cpp:1226-1335): tokens <- argmax, positions / placement / valid_lens += 1.
// One workgroup per task; candidates are scanned in wave order (= ascending row index), ties keep the
// lowest index like torch.argmax / the reference's top-1.
__global__ __launch_bounds__(256) void k_greedy_advance(const float2* ws, int total_waves, int32_t* tokens,
                                                        int32_t* positions, int32_t* placement, int32_t* valid_lens,
                                                        int64_t* next_out) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    // four candidates per trip, loaded before any is compared: one at a time the loop was a chain of dependent round trips
    // (5.5 us for ~8 k candidates inside the decode step)
    for (int w0 = threadIdx.x; w0 < total_waves; w0 += 4 * 256) {
        float2 c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int w = w0 + u * 256;
            c[u] = w < total_waves ? ws[(size_t)b * total_waves + w] : make_float2(0.f, __int_as_float(0x7fffffff));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ci = __float_as_int(c[u].y);
            if (ci != 0x7fffffff && (c[u].x > bv || (c[u].x == bv && ci < bi) || bi == 0x7fffffff)) {
                bv = c[u].x;
                bi = ci;
            }
        }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float ov = sv[threadIdx.x + off];
            const int oi = si[threadIdx.x + off];
            if (oi != 0x7fffffff && (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]) ||
                                     si[threadIdx.x] == 0x7fffffff)) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int tok = si[0] == 0x7fffffff ? 0 : si[0];
        if (tokens) tokens[b] = tok;
        if (next_out) next_out[b] = tok;
        if (positions) positions[b] += 1;
        if (placement) placement[b] += 1;
        if (valid_lens) valid_lens[b] += 1;
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

// y (M, N) fp32 = alpha * x (M, K) . W (N, K)^T with NO rounding to T: the MoE router's logits (Linear::set_output_type(kFloat),
// src/nn/feedforward/feedforward.cpp:285-286: the top-k must see what the fp32 accumulation produced).  One wavefront per output,
// 16-byte loads, fp32 fma chain per lane + wave sum; N = experts (a few hundred) x M = tokens: small.
template <int DT>
__global__ __launch_bounds__(64) void k_gemm_nt_f32(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ w, float* __restrict__ y,
                                                    int n, int k, float alpha) {
    const int col = blockIdx.x, row = blockIdx.y, lane = threadIdx.x;
    const uint16_t* xr = x + (int64_t)row * ldx;
    const uint16_t* wr = w + (int64_t)col * k;
    float acc = 0.f;
    for (int i = lane * 8; i + 8 <= k; i += 64 * 8) {
        const uint4 a = *reinterpret_cast<const uint4*>(xr + i), b = *reinterpret_cast<const uint4*>(wr + i);
        const uint32_t au[4] = {a.x, a.y, a.z, a.w}, bu[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc = __builtin_fmaf(ZT<DT>::to_f32((uint16_t)(au[e] & 0xffffu)), ZT<DT>::to_f32((uint16_t)(bu[e] & 0xffffu)), acc);
            acc = __builtin_fmaf(ZT<DT>::to_f32((uint16_t)(au[e] >> 16)), ZT<DT>::to_f32((uint16_t)(bu[e] >> 16)), acc);
        }
    }
    acc = zl_wave_sum(acc);
    if (lane == 0) y[(int64_t)row * n + col] = alpha * acc;
}

}  // namespace

static int dense_waves(int64_t n, int* rows_per_wave, int* gx) {
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    const int waves = cus * 8;  // two 4-wave workgroups per CU
    *rows_per_wave = (int)((n + waves - 1) / waves);
    *gx = (int)(((n + *rows_per_wave - 1) / *rows_per_wave + 3) / 4);
    return *gx * 4;
}

extern "C" int64_t zl_argmax_workspace_bytes(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return ZL_EINVAL;
    int rpw, gx;
    return m * (int64_t)dense_waves(n, &rpw, &gx) * 8;
}

static int gemm_nt_small_m_impl(const uint16_t* x, int64_t ldx, const uint16_t* w, const uint16_t* bias, uint16_t* y,
                                int64_t m, int64_t n, int64_t k, float alpha, int dtype, const uint16_t* norm_weight,
                                float norm_eps, void* argmax_ws, zl_stream_t s) {
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
    int gx;
    p.total_waves = dense_waves(n, &p.rows_per_wave, &gx);
    p.argmax_ws = reinterpret_cast<float2*>(argmax_ws);
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

extern "C" int zl_gemm_nt_small_m(const uint16_t* x, int64_t ldx, const uint16_t* w, const uint16_t* bias, uint16_t* y,
                                  int64_t m, int64_t n, int64_t k, float alpha, int dtype,
                                  const uint16_t* norm_weight, float norm_eps, zl_stream_t s) {
    return gemm_nt_small_m_impl(x, ldx, w, bias, y, m, n, k, alpha, dtype, norm_weight, norm_eps, nullptr, s);
}

extern "C" int zl_gemm_nt_small_m_argmax(const uint16_t* x, int64_t ldx, const uint16_t* w, const uint16_t* bias,
                                         uint16_t* y, int64_t m, int64_t n, int64_t k, float alpha, int dtype,
                                         const uint16_t* norm_weight, float norm_eps, void* argmax_ws, zl_stream_t s) {
    ZL_CHECK_ARG(argmax_ws, ZL_EINVAL);
    return gemm_nt_small_m_impl(x, ldx, w, bias, y, m, n, k, alpha, dtype, norm_weight, norm_eps, argmax_ws, s);
}

extern "C" int zl_gemm_nt_f32(const uint16_t* x, int64_t ldx, const uint16_t* w, float* y, int64_t m, int64_t n, int64_t k, float alpha, int dtype,
                              zl_stream_t s) {
    ZL_CHECK_ARG(x && w && y && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 8 == 0 && ldx >= k && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && m <= 65535 && n < ((int64_t)1 << 31), ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    const dim3 grid((unsigned)n, (unsigned)m);
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_gemm_nt_f32<ZL_F16>, grid, dim3(64), 0, (hipStream_t)s, x, ldx, w, y, (int)n, (int)k, alpha);
    else hipLaunchKernelGGL(k_gemm_nt_f32<ZL_BF16>, grid, dim3(64), 0, (hipStream_t)s, x, ldx, w, y, (int)n, (int)k, alpha);
    return zl_launch_status();
}

extern "C" int zl_greedy_advance(const void* argmax_ws, int64_t m, int64_t n, int32_t* tokens, int32_t* positions,
                                 int32_t* placement, int32_t* valid_lens, int64_t* next_tokens, zl_stream_t s) {
    ZL_CHECK_ARG(argmax_ws && m > 0 && n > 0, ZL_EINVAL);
    int rpw, gx;
    const int tw = dense_waves(n, &rpw, &gx);
    hipLaunchKernelGGL(k_greedy_advance, dim3((unsigned)m), dim3(256), 0, (hipStream_t)s,
                       reinterpret_cast<const float2*>(argmax_ws), tw, tokens, positions, placement, valid_lens, next_tokens);
    return zl_launch_status();
}

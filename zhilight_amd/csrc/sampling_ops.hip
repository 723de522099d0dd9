// sampling_ops.hip -- the logit post-processing the reference's batch generator runs between a decode step and its host-side search
// (the py_export surface drops in only if src/generator/batch_generator.cpp links): row kernels over (rows, vocab) logits.
//   zl_log_softmax_bias   beam_utility::log_softmax_bias (src/generator/beam_util.cu:19-128): out = T((x - max) / temperature - log(sum) + bias[row]),
//                         max seeded with -1e20, sum with 1e-20, fp32; temperature == 0: the form without the division (:44-66)
//   zl_softmax_rows       functions::softmax (3rd/bmengine/bmengine/functions/softmax.cu:8-30): out = T(exp(x / t - max / t) / sum)
//   zl_topk_rows          functions::TopK::forward (functions/topk.cu:280-293): the `top` largest values of every row in descending order and
//                         their int32 positions (the reference runs a bitonic network; ties go to the LOWER index here)
//   zl_gather_logits      beam_utility::gather_logits (:130-157): out[i] = float(logits[index[i]])
//   zl_scatter_logits     beam_utility::scatter_update (:224-241, 285-315): logits[batch_ids[i], token_ids[i]] = / += T(values[i])
//   zl_repetition_penalty beam_utility::beam_repetition_penalty (:199-222, 243-283): l = presence != 0 ? l - T(presence) :
//                         (l < 0 ? l * T(factor) : l / T(factor)), arithmetic in T as written there
// One workgroup of 256 threads per row (the reference: up to 1024); every row is read a handful of times: a few hundred KB per
// decode step next to the 4.8 GB of weights -- nothing to tune, everything to get right.  T = fp16 / bf16 / fp32 (enum zl_elem_t).
#include <hip/hip_runtime.h>

#include "zl_common.h"

namespace {

template <int TY> struct ET;
template <> struct ET<ZL_T_F16> {
    typedef uint16_t type;
    static __device__ __forceinline__ float ld(const uint16_t* p, int64_t i) { return (float)__builtin_bit_cast(_Float16, p[i]); }
    static __device__ __forceinline__ uint16_t cvt(float f) {
        asm volatile("" : "+v"(f));
        return __builtin_bit_cast(uint16_t, (_Float16)f);
    }
};
template <> struct ET<ZL_T_BF16> {
    typedef uint16_t type;
    static __device__ __forceinline__ float ld(const uint16_t* p, int64_t i) { return __builtin_bit_cast(float, (uint32_t)p[i] << 16); }
    static __device__ __forceinline__ uint16_t cvt(float f) {
        uint32_t u = __builtin_bit_cast(uint32_t, f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
};
template <> struct ET<ZL_T_F32> {
    typedef float type;
    static __device__ __forceinline__ float ld(const float* p, int64_t i) { return p[i]; }
    static __device__ __forceinline__ float cvt(float f) { return f; }
};

constexpr int kThreads = 256;

__device__ __forceinline__ float block_max(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return v;
}

template <int TY>
__global__ __launch_bounds__(kThreads) void k_log_softmax_bias(const typename ET<TY>::type* __restrict__ x, const float* __restrict__ bias,
                                                               typename ET<TY>::type* __restrict__ out, int64_t n, float temperature) {
    __shared__ float sh[4];
    const int64_t off = (int64_t)blockIdx.x * n;
    float m = -1e20f;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) m = fmaxf(m, ET<TY>::ld(x, off + i));
    m = block_max(m, sh);
    const bool scaled = temperature != 0.f;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) {
        const float d = ET<TY>::ld(x, off + i) - m;
        s += expf(scaled ? d / temperature : d);
    }
    s = block_sum(s, sh) + 1e-20f;
    const float ls = logf(s), b = bias ? bias[blockIdx.x] : 0.f;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) {
        const float d = ET<TY>::ld(x, off + i) - m;
        out[off + i] = ET<TY>::cvt((scaled ? d / temperature : d) - ls + b);
    }
}

template <int TY>
__global__ __launch_bounds__(kThreads) void k_softmax_rows(const typename ET<TY>::type* __restrict__ x, typename ET<TY>::type* __restrict__ out, int64_t n,
                                                           float temperature) {
    __shared__ float sh[4];
    const int64_t off = (int64_t)blockIdx.x * n;
    float m = -1e20f;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) m = fmaxf(m, ET<TY>::ld(x, off + i));
    m = block_max(m, sh) / temperature;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) s += expf(ET<TY>::ld(x, off + i) / temperature - m);
    s = block_sum(s, sh) + 1e-20f;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) out[off + i] = ET<TY>::cvt(expf(ET<TY>::ld(x, off + i) / temperature - m) / s);
}

// `top` rounds of "the largest element that comes after the previous pick in (value descending, index ascending) order": no marks,
// no scratch; a row of 128 K logits is read `top` times (top <= a few dozen on the beam-search path)
template <int TY>
__global__ __launch_bounds__(kThreads) void k_topk_rows(const typename ET<TY>::type* __restrict__ x, typename ET<TY>::type* __restrict__ out_v,
                                                        int32_t* __restrict__ out_i, int64_t n, int top) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int64_t off = (int64_t)blockIdx.x * n;
    float prev_v = 0.f;
    int prev_i = -1;
    for (int t = 0; t < top; ++t) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        auto consider = [&](float v, int i) {
            if (!(v == v)) return;                                       // NaN never wins
            const bool after_prev = prev_i < 0 || v < prev_v || (v == prev_v && i > prev_i);
            if (after_prev && (v > bv || (v == bv && i < bi))) {
                bv = v;
                bi = i;
            }
        };
        int64_t i0 = 0;
        if constexpr (sizeof(typename ET<TY>::type) == 2) {
            // 16-byte loads where the row allows it (the pick is a maximum under a total order: any visiting order gives the same
            // element); ADVICE r05: 2-byte loads made a 128 K-entry row ~500 dependent loads per thread and pass
            if ((reinterpret_cast<uintptr_t>(x + off) & 15) == 0) {
                const uint4* row = reinterpret_cast<const uint4*>(x + off);
                const int64_t nv = n / 8;
                for (int64_t j = threadIdx.x; j < nv; j += kThreads) {
                    const uint4 u = row[j];
                    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const uint16_t h = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
                        consider(ET<TY>::ld(&h, 0), (int)(j * 8 + e));
                    }
                }
                i0 = nv * 8;
            }
        }
        for (int64_t i = i0 + threadIdx.x; i < n; i += kThreads) consider(ET<TY>::ld(x, off + i), (int)i);
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((threadIdx.x & 63) == 0) {
            sv[threadIdx.x >> 6] = bv;
            si[threadIdx.x >> 6] = bi;
        }
        __syncthreads();
        bv = sv[0];
        bi = si[0];
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) {
                bv = sv[w];
                bi = si[w];
            }
        __syncthreads();
        if (bi == 0x7fffffff) {                                           // fewer than `top` comparable elements: -inf at position 0
            bv = -INFINITY;
            bi = 0;
            if (threadIdx.x == 0) {
                out_v[(int64_t)blockIdx.x * top + t] = ET<TY>::cvt(bv);
                out_i[(int64_t)blockIdx.x * top + t] = 0;
            }
            continue;
        }
        if (threadIdx.x == 0) {
            out_v[(int64_t)blockIdx.x * top + t] = x[off + bi];
            out_i[(int64_t)blockIdx.x * top + t] = bi;
        }
        prev_v = bv;
        prev_i = bi;
    }
}

template <int TY>
__global__ void k_gather_logits(const int32_t* __restrict__ index, const typename ET<TY>::type* __restrict__ x, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ET<TY>::ld(x, index[i]);
}

template <int TY>
__global__ void k_scatter_logits(const float* __restrict__ values, const int32_t* __restrict__ token_ids, const int32_t* __restrict__ batch_ids,
                                  typename ET<TY>::type* __restrict__ logits, int64_t n, int64_t stride, int add) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t at = (int64_t)batch_ids[i] * stride + token_ids[i];
    const typename ET<TY>::type tv = ET<TY>::cvt(values[i]);
    if (add) logits[at] = ET<TY>::cvt(ET<TY>::ld(logits, at) + ET<TY>::ld(&tv, 0));
    else logits[at] = tv;
}
template <int TY>
__global__ void k_repetition_penalty(const float* __restrict__ factor, const float* __restrict__ presence, const int32_t* __restrict__ tokens,
                                     const int32_t* __restrict__ batch_ids, typename ET<TY>::type* __restrict__ logits, int64_t n, int64_t vocab) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t at = (int64_t)batch_ids[i] * vocab + tokens[i];
    const float l = ET<TY>::ld(logits, at);
    const float pp = presence ? presence[i] : 0.f;
    float r;
    if (pp != 0.f) {
        const typename ET<TY>::type tp = ET<TY>::cvt(pp);
        r = l - ET<TY>::ld(&tp, 0);
    } else {
        const typename ET<TY>::type tf = ET<TY>::cvt(factor[i]);
        const float f = ET<TY>::ld(&tf, 0);
        r = l < 0.f ? l * f : l / f;
    }
    logits[at] = ET<TY>::cvt(r);
}
}  // namespace

#define ZL_TYPE_SWITCH(type, CALL)                      \
    switch (type) {                                     \
        case ZL_T_F16: { CALL(ZL_T_F16) } break;        \
        case ZL_T_BF16: { CALL(ZL_T_BF16) } break;      \
        case ZL_T_F32: { CALL(ZL_T_F32) } break;        \
        default: return ZL_EDTYPE;                      \
    }

extern "C" {

int zl_log_softmax_bias(const void* logits, const float* bias, void* out, int64_t rows, int64_t n, float temperature, int type, zl_stream_t s) {
    ZL_CHECK_ARG(logits && out && rows > 0 && n > 0 && rows < ((int64_t)1 << 31), ZL_EINVAL);
#define CALL(TY) hipLaunchKernelGGL(k_log_softmax_bias<TY>, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)s, (const ET<TY>::type*)logits, bias, (ET<TY>::type*)out, n, temperature);
    ZL_TYPE_SWITCH(type, CALL)
#undef CALL
    return zl_launch_status();
}

int zl_softmax_rows(const void* logits, void* out, int64_t rows, int64_t n, float temperature, int type, zl_stream_t s) {
    ZL_CHECK_ARG(logits && out && rows > 0 && n > 0 && rows < ((int64_t)1 << 31) && temperature > 0.f, ZL_EINVAL);
#define CALL(TY) hipLaunchKernelGGL(k_softmax_rows<TY>, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)s, (const ET<TY>::type*)logits, (ET<TY>::type*)out, n, temperature);
    ZL_TYPE_SWITCH(type, CALL)
#undef CALL
    return zl_launch_status();
}

int zl_topk_rows(const void* x, void* out_v, int32_t* out_i, int64_t rows, int64_t n, int top, int type, zl_stream_t s) {
    ZL_CHECK_ARG(x && out_v && out_i && rows > 0 && n > 0 && top > 0 && rows < ((int64_t)1 << 31) && n < ((int64_t)1 << 31), ZL_EINVAL);
    ZL_CHECK_ARG(top <= n && top <= 4096, ZL_ELIMIT);
#define CALL(TY) hipLaunchKernelGGL(k_topk_rows<TY>, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)s, (const ET<TY>::type*)x, (ET<TY>::type*)out_v, out_i, n, top);
    ZL_TYPE_SWITCH(type, CALL)
#undef CALL
    return zl_launch_status();
}

int zl_gather_logits(const int32_t* index, const void* logits, float* out, int64_t n, int type, zl_stream_t s) {
    ZL_CHECK_ARG(index && logits && out && n > 0, ZL_EINVAL);
    const dim3 grid((unsigned)((n + 255) / 256));
#define CALL(TY) hipLaunchKernelGGL(k_gather_logits<TY>, grid, dim3(256), 0, (hipStream_t)s, index, (const ET<TY>::type*)logits, out, n);
    ZL_TYPE_SWITCH(type, CALL)
#undef CALL
    return zl_launch_status();
}

int zl_scatter_logits(const float* values, const int32_t* token_ids, const int32_t* batch_ids, void* logits, int64_t n, int64_t stride, int add, int type,
                      zl_stream_t s) {
    ZL_CHECK_ARG(values && token_ids && batch_ids && logits && n > 0 && stride > 0, ZL_EINVAL);
    const dim3 grid((unsigned)((n + 255) / 256));
#define CALL(TY) hipLaunchKernelGGL(k_scatter_logits<TY>, grid, dim3(256), 0, (hipStream_t)s, values, token_ids, batch_ids, (ET<TY>::type*)logits, n, stride, add);
    ZL_TYPE_SWITCH(type, CALL)
#undef CALL
    return zl_launch_status();
}

int zl_repetition_penalty(const float* factor, const float* presence, const int32_t* tokens, const int32_t* batch_ids, void* logits, int64_t n, int64_t vocab,
                          int type, zl_stream_t s) {
    ZL_CHECK_ARG(factor && tokens && batch_ids && logits && n > 0 && vocab > 0, ZL_EINVAL);
    const dim3 grid((unsigned)((n + 255) / 256));
#define CALL(TY) hipLaunchKernelGGL(k_repetition_penalty<TY>, grid, dim3(256), 0, (hipStream_t)s, factor, presence, tokens, batch_ids, (ET<TY>::type*)logits, n, vocab);
    ZL_TYPE_SWITCH(type, CALL)
#undef CALL
    return zl_launch_status();
}

}  // extern "C"

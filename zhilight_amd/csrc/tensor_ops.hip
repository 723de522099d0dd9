// tensor_ops.hip -- the small data-movement / element-wise launchers behind the bmengine::functions names the reference's
// layer code calls at LOAD time and around its GEMMs (hostcpp/bm_functions.h): type casts, strided row copies (concat /
// slice of the last dimension), row gathers, |x| row maxima, binary element-wise ops with the two broadcasts bmengine has,
// scalar scaling, in-place activations, a non-finite counter and the act-order permutation helpers.  None of it is on the
// decode step's critical path (weights are prepared once); the kernels are plain grid-stride loops, coalesced over the
// fastest dimension, 16-byte lanes where the shapes allow.  Reference behaviour restated per function in
// include/zhilight_amd.h.
#include "zl_common.h"

namespace {

inline int grid_for(int64_t n, int threads, int cap = 65535 * 8) {
    int64_t g = (n + threads - 1) / threads;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---- element codes = bmengine::core::DataType's enumerators (dtype.h:12-22) ------------------------------------------
enum { T_F64 = 0, T_F32 = 1, T_F16 = 2, T_I8 = 3, T_I16 = 4, T_I32 = 5, T_BF16 = 6 };

template <int T> struct EL;
template <> struct EL<T_F64> { typedef double type; static __device__ float ld(const double* p, int64_t i) { return (float)p[i]; } static __device__ void st(double* p, int64_t i, float v) { p[i] = (double)v; } };
template <> struct EL<T_F32> { typedef float type; static __device__ float ld(const float* p, int64_t i) { return p[i]; } static __device__ void st(float* p, int64_t i, float v) { p[i] = v; } };
template <> struct EL<T_F16> { typedef uint16_t type; static __device__ float ld(const uint16_t* p, int64_t i) { return ZT<ZL_F16>::to_f32(p[i]); } static __device__ void st(uint16_t* p, int64_t i, float v) { p[i] = ZT<ZL_F16>::from_f32(v); } };
template <> struct EL<T_BF16> { typedef uint16_t type; static __device__ float ld(const uint16_t* p, int64_t i) { return ZT<ZL_BF16>::to_f32(p[i]); } static __device__ void st(uint16_t* p, int64_t i, float v) { p[i] = ZT<ZL_BF16>::from_f32(v); } };
template <> struct EL<T_I8> { typedef int8_t type; static __device__ float ld(const int8_t* p, int64_t i) { return (float)p[i]; } static __device__ void st(int8_t* p, int64_t i, float v) { p[i] = (int8_t)v; } };
template <> struct EL<T_I16> { typedef int16_t type; static __device__ float ld(const int16_t* p, int64_t i) { return (float)p[i]; } static __device__ void st(int16_t* p, int64_t i, float v) { p[i] = (int16_t)v; } };
template <> struct EL<T_I32> { typedef int32_t type; static __device__ float ld(const int32_t* p, int64_t i) { return (float)p[i]; } static __device__ void st(int32_t* p, int64_t i, float v) { p[i] = (int32_t)v; } };

template <int TI, int TO>
__global__ void k_cast(const typename EL<TI>::type* __restrict__ in, typename EL<TO>::type* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        EL<TO>::st(out, i, EL<TI>::ld(in, i));
}
template <int TI> int cast_from(const void* in, void* out, int to, int64_t n, hipStream_t s) {
    typedef typename EL<TI>::type I;
    const dim3 g(grid_for(n, 256)), b(256);
    switch (to) {
    case T_F64: hipLaunchKernelGGL((k_cast<TI, T_F64>), g, b, 0, s, (const I*)in, (double*)out, n); break;
    case T_F32: hipLaunchKernelGGL((k_cast<TI, T_F32>), g, b, 0, s, (const I*)in, (float*)out, n); break;
    case T_F16: hipLaunchKernelGGL((k_cast<TI, T_F16>), g, b, 0, s, (const I*)in, (uint16_t*)out, n); break;
    case T_BF16: hipLaunchKernelGGL((k_cast<TI, T_BF16>), g, b, 0, s, (const I*)in, (uint16_t*)out, n); break;
    case T_I8: hipLaunchKernelGGL((k_cast<TI, T_I8>), g, b, 0, s, (const I*)in, (int8_t*)out, n); break;
    case T_I16: hipLaunchKernelGGL((k_cast<TI, T_I16>), g, b, 0, s, (const I*)in, (int16_t*)out, n); break;
    case T_I32: hipLaunchKernelGGL((k_cast<TI, T_I32>), g, b, 0, s, (const I*)in, (int32_t*)out, n); break;
    default: return ZL_EDTYPE;
    }
    return zl_launch_status();
}

// rows x width_bytes, source / destination rows src_pitch / dst_pitch bytes apart.  V = bytes per lane (16, 4 or 1)
template <typename V>
__global__ void k_copy_2d(const char* __restrict__ src, int64_t src_pitch, char* __restrict__ dst, int64_t dst_pitch, int64_t width_v,
                          int64_t rows) {
    const int64_t total = rows * width_v;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / width_v, c = i - r * width_v;
        reinterpret_cast<V*>(dst + r * dst_pitch)[c] = reinterpret_cast<const V*>(src + r * src_pitch)[c];
    }
}

// out[o, j, :] = in[o, index[j], :]; the inner run is inner_v lanes of V
template <typename V>
__global__ void k_index_select(const V* __restrict__ in, V* __restrict__ out, const int32_t* __restrict__ index, int64_t outer,
                               int64_t dim_in, int64_t n_index, int64_t inner_v) {
    const int64_t total = outer * n_index * inner_v;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i % inner_v, j = (i / inner_v) % n_index, o = i / (inner_v * n_index);
        const int64_t src = index[j];
        out[i] = (src >= 0 && src < dim_in) ? in[(o * dim_in + src) * inner_v + c] : V{};
    }
}

// one block per row: max |x| in fp32 (start value -1e4 as the reference, arthmetic.cu:14-27), written back as T
template <int T>
__global__ void k_reduce_abs_max(const typename EL<T>::type* __restrict__ x, typename EL<T>::type* __restrict__ out, int64_t cols) {
    __shared__ float red[16];
    const typename EL<T>::type* row = x + (int64_t)blockIdx.x * cols;
    float m = -1e4f;
    for (int64_t i = threadIdx.x; i < cols; i += blockDim.x) m = fmaxf(m, fabsf(EL<T>::ld(row, i)));
    m = zl_wave_max(m);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) m = fmaxf(m, red[i]);
        EL<T>::st(out, blockIdx.x, m);
    }
}

// c = a (op) b, computed in fp32 and rounded once to T (half / bf16 operands are exact in fp32; the reference computes in
// T with __half operators, which round the exact result once as well).  bmode 0: same shape, 1: b[row], 2: b[col]
template <int T>
__global__ void k_binary(const typename EL<T>::type* __restrict__ a, const typename EL<T>::type* __restrict__ b,
                         typename EL<T>::type* __restrict__ c, int64_t rows, int64_t cols, int op, int bmode) {
    const int64_t n = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = EL<T>::ld(a, i);
        const float y = EL<T>::ld(b, bmode == 0 ? i : bmode == 1 ? i / cols : i % cols);
        float r;
        switch (op) {
        case 0: r = x + y; break;
        case 1: r = x - y; break;
        case 2: r = x * y; break;
        case 3: r = x / y; break;
        default: r = x > y ? x : y; break;
        }
        EL<T>::st(c, i, r);
    }
}

// out = T(float(in) * T(factor)): the reference multiplies in T (functions.cu multiply: a[i] * T(b))
template <int T>
__global__ void k_scale(const typename EL<T>::type* __restrict__ in, typename EL<T>::type* __restrict__ out, int64_t n, float factor_t) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        EL<T>::st(out, i, EL<T>::ld(in, i) * factor_t);
}

template <int T>
__global__ void k_act_inplace(typename EL<T>::type* __restrict__ x, int64_t n, int act) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = EL<T>::ld(x, i);
        const float a = act == 0 ? v / (1.0f + expf(-v)) : 0.5f * v * (1.0f + tanhf(0.7978845608028654f * v * (1.0f + 0.044715f * v * v)));
        EL<T>::st(x, i, a);
    }
}

template <int T>
__global__ void k_count_nonfinite(const typename EL<T>::type* __restrict__ x, int64_t n, int32_t* __restrict__ counter) {
    int bad = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = EL<T>::ld(x, i);
        bad += !(fabsf(v) <= 3.4028234e38f);
    }
    if (bad) atomicAdd(counter, bad);
}

__global__ void k_perm_narrow(const int32_t* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (uint16_t)src[i];
}
__global__ void k_perm_reverse(const int32_t* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = src[i];
        if (j >= 0 && j < n) dst[j] = (uint16_t)i;
    }
}
// act-order regrouping of a (K/8, N) nibble matrix along K: nibble j of out[k8, n] = the nibble of row perm[8 k8 + j]
__global__ void k_gptq_permute_rows(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const int32_t* __restrict__ perm,
                                    int64_t k8, int64_t n) {
    const int64_t total = k8 * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n, c = i - r * n;
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int32_t src = perm[r * 8 + j];
            w |= ((in[(int64_t)(src >> 3) * n + c] >> ((src & 7) * 4)) & 0xfu) << (4 * j);
        }
        out[i] = w;
    }
}
// out[r, i] = x[r, perm[i]], 16-bit elements, one block per row (the gather side is the uncoalesced one)
__global__ void k_permute_input_u16(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ perm,
                                    uint16_t* __restrict__ out, int64_t k) {
    const uint16_t* row = x + (int64_t)blockIdx.x * ldx;
    uint16_t* o = out + (int64_t)blockIdx.x * k;
    for (int64_t i = threadIdx.x; i < k; i += blockDim.x) o[i] = row[perm[i]];
}

}  // namespace

#define ZL_T_SWITCH_FLOAT(t, F16, BF16, F32)   \
    switch (t) {                               \
    case T_F16: F16; break;                    \
    case T_BF16: BF16; break;                  \
    case T_F32: F32; break;                    \
    default: return ZL_EDTYPE;                 \
    }

// Greedy pick over whole logit rows + the between-steps bookkeeping in ONE launch (decode batches past the small-M lm_head, TP's gathered
// rows): row r's arg-max -- the FIRST index of the largest value, a NaN counting as largest, as torch.argmax / the reference's host-side
// pick over the logits (src/generator/batch_generator.cpp:1226-1335 fill_search_tokens) -- into tokens / next_tokens, the three
// counters += 1.  One 64-bit key per element: (monotone(value) << 32) | ~index, so "larger value, then smaller index" is one unsigned max.
template <int TI>
__global__ __launch_bounds__(1024) void k_argmax_advance(const typename EL<TI>::type* __restrict__ x, int64_t ld, int n, int32_t* __restrict__ tokens,
                                                         int32_t* __restrict__ positions, int32_t* __restrict__ placement, int32_t* __restrict__ valid_lens,
                                                         int64_t* __restrict__ next_tokens) {
    __shared__ unsigned long long red[16];
    const typename EL<TI>::type* row = x + (size_t)blockIdx.x * ld;
    unsigned long long best = 0;
    auto take = [&](float v, int i) {
        uint32_t u = __builtin_bit_cast(uint32_t, v);
        u = (v != v) ? 0xffffffffu : (u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u));
        const unsigned long long key = ((unsigned long long)u << 32) | (uint32_t)~(uint32_t)i;
        best = key > best ? key : best;
    };
    int i0 = 0;
    if constexpr (sizeof(typename EL<TI>::type) == 2) {
        // 16-byte lanes, four loads in flight per thread: one element per thread and trip was a chain of ~125 dependent 2-byte
        // round trips -- 46 us per step at 8 and at 32 rows (profiles/r05_bench_b8_kernel_stats.csv), 2 % of the batch-8 step
        if ((((uintptr_t)row) & 15) == 0) {
            const int nv = n / 8;
            for (int c0 = threadIdx.x; c0 < nv; c0 += 4 * 1024) {
                uint4 v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + u * 1024;
                    v4[u] = *reinterpret_cast<const uint4*>(row + (size_t)(c < nv ? c : c0) * 8);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + u * 1024;
                    if (c < nv) {
                        const uint32_t w[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const uint16_t h = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
                            take(EL<TI>::ld(&h, 0), c * 8 + e);
                        }
                    }
                }
            }
            i0 = nv * 8;
        }
    }
    for (int i = i0 + threadIdx.x; i < n; i += 1024) take(EL<TI>::ld(row, i), i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) best = red[w] > best ? red[w] : best;
        const int32_t idx = (int32_t)~(uint32_t)(best & 0xffffffffull);
        if (tokens) tokens[blockIdx.x] = idx;
        if (next_tokens) next_tokens[blockIdx.x] = idx;
        if (positions) positions[blockIdx.x] += 1;
        if (placement) placement[blockIdx.x] += 1;
        if (valid_lens) valid_lens[blockIdx.x] += 1;
    }
}

// ---- the MoE dispatch route's index plumbing (functions::arange / sort_pair_1d / divide / scatter_update_dim0) ------------------------
__global__ void k_arange_i32(int32_t* __restrict__ out, int start, int step, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = start + (int)i * step;
}
__global__ void k_divide_i32(const int32_t* __restrict__ a, int32_t* __restrict__ out, int divisor, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] / divisor;
}
// dst[dst_index[i], :] = src[src_index ? src_index[i] : i, :]; rows of row_bytes (a multiple of 2); grid n_index
__global__ __launch_bounds__(256) void k_scatter_rows(unsigned char* __restrict__ dst, const int32_t* __restrict__ dst_index,
                                                       const unsigned char* __restrict__ src, const int32_t* __restrict__ src_index, int64_t row_bytes,
                                                       int64_t dst_rows, int64_t src_rows) {
    const int64_t i = blockIdx.x;
    const int64_t x = dst_index[i], y = src_index ? src_index[i] : i;
    if (x < 0 || x >= dst_rows || y < 0 || y >= src_rows) return;            // (the reference asserts; here an index outside is dropped)
    const unsigned char* sp = src + y * row_bytes;
    unsigned char* dp = dst + x * row_bytes;
    if (row_bytes % 16 == 0 && (((uintptr_t)sp | (uintptr_t)dp) & 15) == 0) {
        for (int64_t o = threadIdx.x * 16; o < row_bytes; o += 256 * 16) *reinterpret_cast<uint4*>(dp + o) = *reinterpret_cast<const uint4*>(sp + o);
    } else {
        for (int64_t o = threadIdx.x * 2; o < row_bytes; o += 256 * 2) *reinterpret_cast<uint16_t*>(dp + o) = *reinterpret_cast<const uint16_t*>(sp + o);
    }
}
// Stable least-significant-digit radix sort of (int32 key >= 0, int32 value) pairs in ONE workgroup of 512 threads, 4 bits per pass:
// thread t owns the contiguous chunk [t c, (t + 1) c) of the current order, counts its 16 digit values, an exclusive scan over
// (digit, thread) in LDS gives every thread the output position of its first element per digit, and the chunk is written in order --
// stable by construction (cub::DeviceRadixSort::SortPairs, which functions::sort_pair_1d wraps, is stable too: the MoE dispatch
// relies on tokens staying in order inside an expert's run, feedforward.cpp:599-629).  bits = ceil(log2(max_key + 1)); max_key <= 0: all 32 bits in signed order (sign bit flipped).  n <= 2^20: a decode / prompt-chunk routing table (tokens x top_k), not a general-purpose sort.
__global__ __launch_bounds__(512) void k_sort_pairs_i32(const int32_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in, int32_t* keys_a,
                                                          int32_t* vals_a, int32_t* keys_b, int32_t* vals_b, int n, int bits, uint32_t flip) {
    __shared__ int cnt[16 * 512];                    // [digit][thread]
    __shared__ int wsum[16];
    const int t = threadIdx.x, c = (n + 511) / 512, lo = min(n, t * c), hi = min(n, lo + c);
    const int passes = (bits + 3) / 4;
    // the result must land in (keys_a, vals_a): with an odd number of passes the first pass writes a, else b
    const int32_t* ksrc = keys_in;
    const int32_t* vsrc = vals_in;
    for (int p = 0; p < passes; ++p) {
        const bool to_a = ((passes - 1 - p) & 1) == 0;
        int32_t* kdst = to_a ? keys_a : keys_b;
        int32_t* vdst = to_a ? vals_a : vals_b;
        const int shift = 4 * p;
        int local[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) local[d] = 0;
        for (int i = lo; i < hi; ++i) {
            const int d = (int)((((uint32_t)ksrc[i]) ^ flip) >> shift) & 15;
#pragma unroll
            for (int e = 0; e < 16; ++e) local[e] += (e == d);
        }
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d * 512 + t] = local[d];
        __syncthreads();
        // exclusive scan over the flattened (digit, thread) order: 16 x 512 entries, 16 per thread
        int run = 0, mine[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            mine[j] = cnt[t * 16 + j];
            run += mine[j];
        }
        // block-wide exclusive scan of `run`
        __shared__ int part[512];
        part[t] = run;
        __syncthreads();
        for (int off = 1; off < 512; off <<= 1) {
            const int v = t >= off ? part[t - off] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        int base = part[t] - run;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            cnt[t * 16 + j] = base;
            base += mine[j];
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < 16; ++d) local[d] = cnt[d * 512 + t];
        for (int i = lo; i < hi; ++i) {
            const int k = ksrc[i], d = (int)((((uint32_t)k) ^ flip) >> shift) & 15;
            int pos = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (e == d) { pos = local[e]; local[e] += 1; }
            }
            kdst[pos] = k;
            vdst[pos] = vsrc[i];
        }
        __threadfence_block();
        __syncthreads();
        ksrc = kdst;
        vsrc = vdst;
        (void)wsum;
    }
}

extern "C" {

int zl_cast(const void* in, int in_type, void* out, int out_type, int64_t n, zl_stream_t s) {
    ZL_CHECK_ARG(in && out && n > 0, ZL_EINVAL);
    hipStream_t hs = (hipStream_t)s;
    switch (in_type) {
    case T_F64: return cast_from<T_F64>(in, out, out_type, n, hs);
    case T_F32: return cast_from<T_F32>(in, out, out_type, n, hs);
    case T_F16: return cast_from<T_F16>(in, out, out_type, n, hs);
    case T_BF16: return cast_from<T_BF16>(in, out, out_type, n, hs);
    case T_I8: return cast_from<T_I8>(in, out, out_type, n, hs);
    case T_I16: return cast_from<T_I16>(in, out, out_type, n, hs);
    case T_I32: return cast_from<T_I32>(in, out, out_type, n, hs);
    default: return ZL_EDTYPE;
    }
}

int zl_copy_2d(const void* src, int64_t src_pitch, void* dst, int64_t dst_pitch, int64_t width_bytes, int64_t rows, zl_stream_t s) {
    ZL_CHECK_ARG(src && dst && width_bytes > 0 && rows > 0 && src_pitch >= width_bytes && dst_pitch >= width_bytes, ZL_EINVAL);
    const uintptr_t bits = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)src_pitch | (uintptr_t)dst_pitch | (uintptr_t)width_bytes;
    hipStream_t hs = (hipStream_t)s;
    if ((bits & 15) == 0)
        hipLaunchKernelGGL(k_copy_2d<uint4>, dim3(grid_for(rows * (width_bytes / 16), 256)), dim3(256), 0, hs, (const char*)src, src_pitch,
                           (char*)dst, dst_pitch, width_bytes / 16, rows);
    else if ((bits & 3) == 0)
        hipLaunchKernelGGL(k_copy_2d<uint32_t>, dim3(grid_for(rows * (width_bytes / 4), 256)), dim3(256), 0, hs, (const char*)src, src_pitch,
                           (char*)dst, dst_pitch, width_bytes / 4, rows);
    else
        hipLaunchKernelGGL(k_copy_2d<uint8_t>, dim3(grid_for(rows * width_bytes, 256)), dim3(256), 0, hs, (const char*)src, src_pitch,
                           (char*)dst, dst_pitch, width_bytes, rows);
    return zl_launch_status();
}

int zl_index_select(const void* in, void* out, const int32_t* index, int64_t outer, int64_t dim_in, int64_t n_index, int64_t inner_bytes,
                    zl_stream_t s) {
    ZL_CHECK_ARG(in && out && index && outer > 0 && dim_in > 0 && n_index > 0 && inner_bytes > 0, ZL_EINVAL);
    const uintptr_t bits = (uintptr_t)in | (uintptr_t)out | (uintptr_t)inner_bytes;
    hipStream_t hs = (hipStream_t)s;
    const int64_t rows = outer * n_index;
    if ((bits & 15) == 0)
        hipLaunchKernelGGL(k_index_select<uint4>, dim3(grid_for(rows * (inner_bytes / 16), 256)), dim3(256), 0, hs, (const uint4*)in,
                           (uint4*)out, index, outer, dim_in, n_index, inner_bytes / 16);
    else if ((bits & 3) == 0)
        hipLaunchKernelGGL(k_index_select<uint32_t>, dim3(grid_for(rows * (inner_bytes / 4), 256)), dim3(256), 0, hs, (const uint32_t*)in,
                           (uint32_t*)out, index, outer, dim_in, n_index, inner_bytes / 4);
    else if ((bits & 1) == 0)
        hipLaunchKernelGGL(k_index_select<uint16_t>, dim3(grid_for(rows * (inner_bytes / 2), 256)), dim3(256), 0, hs, (const uint16_t*)in,
                           (uint16_t*)out, index, outer, dim_in, n_index, inner_bytes / 2);
    else
        hipLaunchKernelGGL(k_index_select<uint8_t>, dim3(grid_for(rows * inner_bytes, 256)), dim3(256), 0, hs, (const uint8_t*)in,
                           (uint8_t*)out, index, outer, dim_in, n_index, inner_bytes);
    return zl_launch_status();
}

int zl_argmax_advance(const void* logits, int type, int64_t rows, int64_t n, int64_t ld, int32_t* tokens, int32_t* positions, int32_t* placement,
                      int32_t* valid_lens, int64_t* next_tokens, zl_stream_t s) {
    ZL_CHECK_ARG(logits && rows > 0 && n > 0 && ld >= n && (tokens || next_tokens), ZL_EINVAL);
    ZL_CHECK_ARG(n < ((int64_t)1 << 31) && rows < ((int64_t)1 << 31), ZL_ESHAPE);
    hipStream_t hs = (hipStream_t)s;
    const dim3 g((unsigned)rows), b(1024);
    switch (type) {
    case T_F16: hipLaunchKernelGGL(k_argmax_advance<T_F16>, g, b, 0, hs, (const uint16_t*)logits, ld, (int)n, tokens, positions, placement, valid_lens, next_tokens); break;
    case T_BF16: hipLaunchKernelGGL(k_argmax_advance<T_BF16>, g, b, 0, hs, (const uint16_t*)logits, ld, (int)n, tokens, positions, placement, valid_lens, next_tokens); break;
    case T_F32: hipLaunchKernelGGL(k_argmax_advance<T_F32>, g, b, 0, hs, (const float*)logits, ld, (int)n, tokens, positions, placement, valid_lens, next_tokens); break;
    default: return ZL_EDTYPE;
    }
    return zl_launch_status();
}

int zl_arange_i32(int32_t* out, int32_t start, int32_t step, int64_t n, zl_stream_t s) {
    ZL_CHECK_ARG(out && n > 0 && step != 0, ZL_EINVAL);
    hipLaunchKernelGGL(k_arange_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, out, start, step, n);
    return zl_launch_status();
}
int zl_divide_i32(const int32_t* a, int32_t* out, int32_t divisor, int64_t n, zl_stream_t s) {
    ZL_CHECK_ARG(a && out && n > 0 && divisor != 0, ZL_EINVAL);
    hipLaunchKernelGGL(k_divide_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, a, out, divisor, n);
    return zl_launch_status();
}
int zl_scatter_update_dim0(void* dst, const int32_t* dst_index, const void* src, const int32_t* src_index, int64_t n_index, int64_t row_bytes,
                           int64_t dst_rows, int64_t src_rows, zl_stream_t s) {
    ZL_CHECK_ARG(dst && dst_index && src && n_index > 0 && row_bytes > 0 && dst_rows > 0 && src_rows > 0, ZL_EINVAL);
    ZL_CHECK_ARG(row_bytes % 2 == 0 && n_index < ((int64_t)1 << 31), ZL_ESHAPE);
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)n_index), dim3(256), 0, (hipStream_t)s, (unsigned char*)dst, dst_index, (const unsigned char*)src, src_index,
                       row_bytes, dst_rows, src_rows);
    return zl_launch_status();
}
int zl_sort_pairs_i32(const int32_t* keys, const int32_t* values, int32_t* keys_out, int32_t* values_out, void* workspace, int64_t n, int32_t max_key,
                      zl_stream_t s) {
    ZL_CHECK_ARG(keys && values && keys_out && values_out && workspace && n > 0, ZL_EINVAL);
    ZL_CHECK_ARG(n <= ((int64_t)1 << 20), ZL_ELIMIT);
    // max_key > 0: keys in [0, max_key], only the bits that can differ.  max_key <= 0: the FULL 32-bit SIGNED order of
    // cub::DeviceRadixSort on int32 (what functions::sort_pair_1d does without a key bound): eight passes with the sign bit
    // flipped, so that a negative key -- an unfilled -1 expert id -- sorts in front, as in the reference
    int bits = 32;
    uint32_t flip = 0x80000000u;
    if (max_key > 0) {
        bits = 1;
        flip = 0;
        while (bits < 31 && (max_key >> bits) != 0) ++bits;
    }
    int32_t* kb = static_cast<int32_t*>(workspace);
    hipLaunchKernelGGL(k_sort_pairs_i32, dim3(1), dim3(512), 0, (hipStream_t)s, keys, values, keys_out, values_out, kb, kb + n, (int)n, bits, flip);
    return zl_launch_status();
}

int zl_reduce_abs_max(const void* x, void* out, int64_t rows, int64_t cols, int type, zl_stream_t s) {
    ZL_CHECK_ARG(x && out && rows > 0 && cols > 0 && rows < ((int64_t)1 << 31), ZL_EINVAL);
    const dim3 g((unsigned)rows), b(cols >= 1024 ? 1024 : (unsigned)((cols + 63) / 64 * 64));
    hipStream_t hs = (hipStream_t)s;
    ZL_T_SWITCH_FLOAT(type, hipLaunchKernelGGL(k_reduce_abs_max<T_F16>, g, b, 0, hs, (const uint16_t*)x, (uint16_t*)out, cols),
                      hipLaunchKernelGGL(k_reduce_abs_max<T_BF16>, g, b, 0, hs, (const uint16_t*)x, (uint16_t*)out, cols),
                      hipLaunchKernelGGL(k_reduce_abs_max<T_F32>, g, b, 0, hs, (const float*)x, (float*)out, cols))
    return zl_launch_status();
}

int zl_binary_op(const void* a, const void* b, void* c, int64_t rows, int64_t cols, int op, int bmode, int type, zl_stream_t s) {
    ZL_CHECK_ARG(a && b && c && rows > 0 && cols > 0 && op >= 0 && op <= 4 && bmode >= 0 && bmode <= 2, ZL_EINVAL);
    const dim3 g(grid_for(rows * cols, 256)), bl(256);
    hipStream_t hs = (hipStream_t)s;
    ZL_T_SWITCH_FLOAT(type,
                      hipLaunchKernelGGL(k_binary<T_F16>, g, bl, 0, hs, (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)c, rows, cols, op, bmode),
                      hipLaunchKernelGGL(k_binary<T_BF16>, g, bl, 0, hs, (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)c, rows, cols, op, bmode),
                      hipLaunchKernelGGL(k_binary<T_F32>, g, bl, 0, hs, (const float*)a, (const float*)b, (float*)c, rows, cols, op, bmode))
    return zl_launch_status();
}

int zl_scale(const void* in, void* out, int64_t n, float factor, int type, zl_stream_t s) {
    ZL_CHECK_ARG(in && out && n > 0, ZL_EINVAL);
    const dim3 g(grid_for(n, 256)), b(256);
    hipStream_t hs = (hipStream_t)s;
    // the factor rounded to T on the host, as T(b) in the reference's kernel argument
    const float f16 = (float)(_Float16)factor;
    uint32_t u = __builtin_bit_cast(uint32_t, factor);
    u += 0x7fffu + ((u >> 16) & 1u);
    const float bf = __builtin_bit_cast(float, u & 0xffff0000u);
    ZL_T_SWITCH_FLOAT(type, hipLaunchKernelGGL(k_scale<T_F16>, g, b, 0, hs, (const uint16_t*)in, (uint16_t*)out, n, f16),
                      hipLaunchKernelGGL(k_scale<T_BF16>, g, b, 0, hs, (const uint16_t*)in, (uint16_t*)out, n, bf),
                      hipLaunchKernelGGL(k_scale<T_F32>, g, b, 0, hs, (const float*)in, (float*)out, n, factor))
    return zl_launch_status();
}

int zl_act_inplace(void* x, int64_t n, int act, int type, zl_stream_t s) {
    ZL_CHECK_ARG(x && n > 0 && (act == 0 || act == 1), ZL_EINVAL);
    const dim3 g(grid_for(n, 256)), b(256);
    hipStream_t hs = (hipStream_t)s;
    ZL_T_SWITCH_FLOAT(type, hipLaunchKernelGGL(k_act_inplace<T_F16>, g, b, 0, hs, (uint16_t*)x, n, act),
                      hipLaunchKernelGGL(k_act_inplace<T_BF16>, g, b, 0, hs, (uint16_t*)x, n, act),
                      hipLaunchKernelGGL(k_act_inplace<T_F32>, g, b, 0, hs, (float*)x, n, act))
    return zl_launch_status();
}

int zl_count_nonfinite(const void* x, int64_t n, int type, int32_t* counter, zl_stream_t s) {
    ZL_CHECK_ARG(x && counter && n > 0, ZL_EINVAL);
    const dim3 g(grid_for(n, 256, 4096)), b(256);
    hipStream_t hs = (hipStream_t)s;
    ZL_T_SWITCH_FLOAT(type, hipLaunchKernelGGL(k_count_nonfinite<T_F16>, g, b, 0, hs, (const uint16_t*)x, n, counter),
                      hipLaunchKernelGGL(k_count_nonfinite<T_BF16>, g, b, 0, hs, (const uint16_t*)x, n, counter),
                      hipLaunchKernelGGL(k_count_nonfinite<T_F32>, g, b, 0, hs, (const float*)x, n, counter))
    return zl_launch_status();
}

int zl_perm_narrow_u16(const int32_t* perm, uint16_t* out, int64_t k, zl_stream_t s) {
    ZL_CHECK_ARG(perm && out && k > 0 && k <= 65536, ZL_EINVAL);
    hipLaunchKernelGGL(k_perm_narrow, dim3(grid_for(k, 256)), dim3(256), 0, (hipStream_t)s, perm, out, k);
    return zl_launch_status();
}
int zl_perm_reverse_u16(const int32_t* perm, uint16_t* out, int64_t k, zl_stream_t s) {
    ZL_CHECK_ARG(perm && out && k > 0 && k <= 65536, ZL_EINVAL);
    hipLaunchKernelGGL(k_perm_reverse, dim3(grid_for(k, 256)), dim3(256), 0, (hipStream_t)s, perm, out, k);
    return zl_launch_status();
}
int zl_gptq_permute_rows(const uint32_t* qweight, uint32_t* out, const int32_t* perm, int64_t k8, int64_t n, zl_stream_t s) {
    ZL_CHECK_ARG(qweight && out && perm && qweight != out && k8 > 0 && n > 0, ZL_EINVAL);
    hipLaunchKernelGGL(k_gptq_permute_rows, dim3(grid_for(k8 * n, 256)), dim3(256), 0, (hipStream_t)s, qweight, out, perm, k8, n);
    return zl_launch_status();
}
int zl_permute_input_u16(const uint16_t* x, int64_t ldx, const uint16_t* perm, uint16_t* out, int64_t rows, int64_t k, zl_stream_t s) {
    ZL_CHECK_ARG(x && perm && out && rows > 0 && k > 0 && k <= 65536 && ldx >= k && rows < ((int64_t)1 << 31), ZL_EINVAL);
    hipLaunchKernelGGL(k_permute_input_u16, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)s, x, ldx, perm, out, k);
    return zl_launch_status();
}

}  // extern "C"

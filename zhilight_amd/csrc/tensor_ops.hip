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

// w4_layout.hip -- load-time integer layout transforms and the ZLW4 packer.
//
// All of this is one-time, HBM-bound byte shuffling: plain coalesced grid-stride kernels, no LDS.
// Reference behaviour restated (not copied) from src/nn/quant/gptq/utils.cu:25-214,
// src/nn/quant/gptq/qdq_4.cuh:16-35 and q_gemm.cu:778-791; results are bit-exact vs oracle/.
#include "zl_common.h"

namespace {

constexpr int kThreads = 256;
inline int grid_for(int64_t n) {
    int64_t g = (n + kThreads - 1) / kThreads;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

__device__ __forceinline__ uint32_t shuffle_word(uint32_t qa) {
    // even weights -> low 16 bits, odd weights -> high 16 bits (exllama order)
    uint32_t qb = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        qb |= ((qa >> (8 * j)) & 0xfu) << (4 * j);
        qb |= ((qa >> (8 * j + 4)) & 0xfu) << (4 * j + 16);
    }
    return qb;
}

__global__ void k_shuffle(uint32_t* q, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        q[i] = shuffle_word(q[i]);
}

__global__ void k_increase_zero(uint32_t* q, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = q[i];
        // per-nibble +1 with 0xF -> 0: add 1 to every nibble that is not 0xF, clear those that are
        uint32_t is_f = v & (v >> 1) & (v >> 2) & (v >> 3) & 0x11111111u;  // 1 where nibble == 0xF
        uint32_t keep = ~(is_f * 0xFu);
        q[i] = ((v & keep) + (0x11111111u & ~is_f));  // non-F nibbles never carry out
    }
}

__global__ void k_q4_to_q8(const uint32_t* in, uint2* out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = in[i];
        uint2 o;
        o.x = (v & 0xfu) | ((v & 0xf0u) << 4) | ((v & 0xf00u) << 8) | ((v & 0xf000u) << 12);
        v >>= 16;
        o.y = (v & 0xfu) | ((v & 0xf0u) << 4) | ((v & 0xf00u) << 8) | ((v & 0xf000u) << 12);
        out[i] = o;
    }
}

// LDS-tiled transpose, 32x32 tile (+1 pad), any 1/2/4-byte element
template <typename T>
__global__ void k_transpose(const T* __restrict__ in, T* __restrict__ out, int64_t rows, int64_t cols) {
    __shared__ T tile[32][33];
    int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int64_t r = r0 + i, c = c0 + tx;
        if (r < rows && c < cols) tile[i][tx] = in[r * cols + c];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int64_t c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][i];
    }
}

__global__ void k_awq_un_shuffle(uint32_t* q, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = q[i], o = 0;
        const int de[8] = {0, 4, 1, 5, 2, 6, 3, 7};
#pragma unroll
        for (int s = 0; s < 8; ++s) o |= ((v >> (de[s] * 4)) & 0xfu) << (s * 4);
        q[i] = o;
    }
}

// AWQ (K, N/8) words -> (K/8, N) words; one thread per (k/8, n/8) block of 8x8 nibbles
__global__ void k_awq_shuffle(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t k8, int64_t n8,
                              int use_exllama) {
    const int de[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    const int sfl[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    int64_t total = k8 * n8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t kb = i / n8, nb = i % n8;
        uint32_t rowsw[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) rowsw[r] = in[(kb * 8 + r) * n8 + nb];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t q = 0;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                int r = use_exllama ? sfl[s] : s;
                q |= ((rowsw[r] >> (de[c] * 4)) & 0xfu) << (s * 4);
            }
            out[kb * (n8 * 8) + nb * 8 + c] = q;
        }
    }
}

// ---- ZLW4 packer --------------------------------------------------------------------------
// one thread per destination word / scale quad / zero quad
__global__ void k_pack_qw(const uint32_t* __restrict__ qw_km, uint32_t* __restrict__ dst, int64_t n, int64_t k8,
                          int64_t np, int64_t q_loads, int interleave) {
    int64_t total = (np / 2) * q_loads * 256;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
        int64_t t = i >> 8;
        int64_t q = t % q_loads, pr = t / q_loads;
        int64_t row = 2 * pr + (lane >> 5);
        int64_t word = (lane & 31) + 32 * (4 * q + j);
        int64_t src_row = interleave ? ((row & 1) * (n / 2) + (row >> 1)) : row;
        uint32_t v = 0;
        if (row < n && word < k8) v = qw_km[src_row * k8 + word];
        dst[i] = v;
    }
}

__global__ void k_pack_meta(const uint8_t* __restrict__ qz_km, const uint16_t* __restrict__ sc_km,
                            uint16_t* __restrict__ scales, uint16_t* __restrict__ zeros, int64_t n, int64_t ng,
                            int64_t g, int64_t np, int64_t q_loads, int64_t c_classes, int interleave) {
    int64_t total = np * q_loads * c_classes;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // destination index i = ((pr * Q + q) * 2 + h) * C + c
        int64_t c = i % c_classes, t = i / c_classes;
        int64_t hh = t & 1;
        t >>= 1;
        int64_t q = t % q_loads, row = 2 * (t / q_loads) + hh;
        int64_t src_row = interleave ? ((row & 1) * (n / 2) + (row >> 1)) : row;
        uint16_t z4 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t grp = (1024 * q + 256 * j) / g + c;
            uint16_t sc = 0, z = 0;
            if (row < n && grp < ng) {
                sc = sc_km[src_row * ng + grp];
                z = qz_km[src_row * ng + grp] & 0xf;
            }
            scales[i * 4 + j] = sc;
            z4 |= (uint16_t)(z << (4 * j));
        }
        zeros[i] = z4;
    }
}

// dequantise the packed layout back to (N, K) fp16: rn16(rn16(q - z) * s)
__global__ void k_w4_dequant(const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
                             const uint16_t* __restrict__ zeros, uint16_t* __restrict__ out, int64_t n, int64_t k,
                             int64_t g, int64_t q_loads, int64_t c_classes) {
    int64_t k8 = k / 8, total = n * k8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t row = i / k8, w = i % k8;
        int64_t p = w >> 5, r = w & 31, q = p >> 2, j = p & 3;
        int lane = (int)((row & 1) * 32 + r);
        uint32_t word = qw[(((row >> 1) * q_loads + q) * 64 + lane) * 4 + j];
        int64_t c = (c_classes > 1) ? (r / (32 / c_classes)) : 0;
        int64_t mi = (((row >> 1) * q_loads + q) * 2 + (row & 1)) * c_classes + c;
        float s = (float)__builtin_bit_cast(_Float16, scales[mi * 4 + j]);
        int z = (zeros[mi] >> (4 * j)) & 0xf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int nib = (e & 1) ? ((word >> (4 * (e >> 1) + 16)) & 0xf) : ((word >> (4 * (e >> 1))) & 0xf);
            _Float16 d = (_Float16)(float)(nib - z);
            _Float16 v = zl_f32_to_f16((float)d * s);  // product of two fp16 is exact in fp32 -> one rounding
            out[row * k + 8 * w + e] = __builtin_bit_cast(uint16_t, v);
        }
    }
}

}  // namespace

extern "C" {

int zl_gptq_shuffle(uint32_t* qweight, int64_t k8, int64_t n, zl_stream_t s) {
    ZL_CHECK_ARG(qweight && k8 > 0 && n > 0, ZL_EINVAL);
    hipLaunchKernelGGL(k_shuffle, dim3(grid_for(k8 * n)), dim3(kThreads), 0, (hipStream_t)s, qweight, k8 * n);
    return zl_launch_status();
}

int zl_gptq_increase_zero(uint32_t* qzeros, int64_t nwords, zl_stream_t s) {
    ZL_CHECK_ARG(qzeros && nwords > 0, ZL_EINVAL);
    hipLaunchKernelGGL(k_increase_zero, dim3(grid_for(nwords)), dim3(kThreads), 0, (hipStream_t)s, qzeros, nwords);
    return zl_launch_status();
}

int zl_gptq_q4_to_q8(const uint32_t* in, uint8_t* out, int64_t nwords, zl_stream_t s) {
    ZL_CHECK_ARG(in && out && nwords > 0, ZL_EINVAL);
    hipLaunchKernelGGL(k_q4_to_q8, dim3(grid_for(nwords)), dim3(kThreads), 0, (hipStream_t)s, in, (uint2*)out, nwords);
    return zl_launch_status();
}

// nn::gptq::reconstruct_gptq (src/nn/quant/gptq/q_gemm.cu:641-676): the (K/8, N) checkpoint-order weight of the legacy route
// (GPTQ_KERNEL_ALGO=0 without exllama: a row-parallel act-order shard) as a dense fp16 (K, N) matrix,
//     out[k][n] = hmul(int2half(q[k][n] - zero[g][n]), scale[g][n]),  g = g_idx[k] (or k / group_size),
// zero = the stored nibble (increase_zero already ran at load).  One thread per (word row, column): eight k of one column.
namespace {
__global__ __launch_bounds__(256) void k_gptq_reconstruct(const uint32_t* __restrict__ qw, const uint32_t* __restrict__ qz,
                                                          const uint16_t* __restrict__ sc, const int32_t* __restrict__ g_idx,
                                                          uint16_t* __restrict__ out, int64_t k, int64_t n, int64_t groups) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x, row8 = blockIdx.y;
    if (col >= n) return;
    const uint32_t w = qw[row8 * n + col];
    const int64_t gsz = k / groups;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t kk = row8 * 8 + e;
        const int64_t g = g_idx ? (int64_t)g_idx[kk] : kk / gsz;
        const int z = (int)((qz[g * (n / 8) + col / 8] >> ((col & 7) * 4)) & 0xf);
        const _Float16 d = (_Float16)((int)((w >> (4 * e)) & 0xf) - z);                  // |value| <= 15: exact
        const _Float16 v = d * __builtin_bit_cast(_Float16, sc[g * n + col]);            // __hmul: one fp16 rounding
        out[kk * n + col] = __builtin_bit_cast(uint16_t, v);
    }
}
}  // namespace

int zl_gptq_reconstruct(const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales, const int32_t* g_idx, uint16_t* out,
                        int64_t k, int64_t n, int64_t groups, zl_stream_t s) {
    ZL_CHECK_ARG(qweight && qzeros && scales && out && k > 0 && n > 0 && groups > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 8 == 0 && n % 8 == 0 && k % groups == 0 && k / 8 <= 65535, ZL_ESHAPE);
    hipLaunchKernelGGL(k_gptq_reconstruct, dim3((unsigned)((n + 255) / 256), (unsigned)(k / 8)), dim3(256), 0, (hipStream_t)s, qweight,
                       qzeros, scales, g_idx, out, k, n, groups);
    return zl_launch_status();
}

int zl_transpose_2d(const void* in, void* out, int64_t rows, int64_t cols, int elem_size, zl_stream_t s) {
    ZL_CHECK_ARG(in && out && rows > 0 && cols > 0, ZL_EINVAL);
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    ZL_CHECK_ARG(grid.y <= 65535, ZL_ELIMIT);
    if (elem_size == 4)
        hipLaunchKernelGGL(k_transpose<uint32_t>, grid, dim3(256), 0, (hipStream_t)s, (const uint32_t*)in, (uint32_t*)out, rows, cols);
    else if (elem_size == 2)
        hipLaunchKernelGGL(k_transpose<uint16_t>, grid, dim3(256), 0, (hipStream_t)s, (const uint16_t*)in, (uint16_t*)out, rows, cols);
    else if (elem_size == 1)
        hipLaunchKernelGGL(k_transpose<uint8_t>, grid, dim3(256), 0, (hipStream_t)s, (const uint8_t*)in, (uint8_t*)out, rows, cols);
    else
        return ZL_EDTYPE;
    return zl_launch_status();
}

int zl_awq_un_shuffle(uint32_t* q, int64_t dim0, int64_t n, zl_stream_t s) {
    ZL_CHECK_ARG(q && dim0 > 0 && n > 0, ZL_EINVAL);
    hipLaunchKernelGGL(k_awq_un_shuffle, dim3(grid_for(dim0 * n)), dim3(kThreads), 0, (hipStream_t)s, q, dim0 * n);
    return zl_launch_status();
}

int zl_awq_shuffle(const uint32_t* in, uint32_t* out, int64_t k, int64_t n, int use_exllama, zl_stream_t s) {
    ZL_CHECK_ARG(in && out && k > 0 && n > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 8 == 0 && n % 8 == 0, ZL_ESHAPE);
    hipLaunchKernelGGL(k_awq_shuffle, dim3(grid_for((k / 8) * (n / 8))), dim3(kThreads), 0, (hipStream_t)s, in, out,
                       k / 8, n / 8, use_exllama);
    return zl_launch_status();
}

int zl_w4_layout(int64_t n, int64_t k, int64_t g, zl_w4_layout_t* out) {
    ZL_CHECK_ARG(out && n > 0 && k > 0 && g > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 8 == 0 && k % g == 0 && g % 32 == 0 && (g & (g - 1)) == 0 || (g == k && k % 8 == 0), ZL_ESHAPE);
    // group boundaries must coincide with the per-lane 8-weight words and the (q, j) 256-weight
    // blocks: power-of-two G >= 32, or one group per row (G == K, any K)
    out->n = n;
    out->k = k;
    out->group_size = g;
    out->np = (n + 1) / 2 * 2;
    out->kp = (k + 1023) / 1024 * 1024;
    out->q = out->kp / 1024;
    out->c = g >= 256 ? 1 : 256 / g;
    out->qw_bytes = (out->np / 2) * out->q * 64 * 4 * 4;
    out->scales_bytes = out->np * out->q * out->c * 4 * 2;
    out->zeros_bytes = out->np * out->q * out->c * 2;
    return ZL_OK;
}

int zl_w4_pack(const uint32_t* qweight_km, const uint8_t* qzeros_km, const uint16_t* scales_km, int64_t n, int64_t k,
               int64_t g, int row_interleave, uint32_t* qw, uint16_t* scales, uint16_t* zeros, zl_stream_t s) {
    ZL_CHECK_ARG(qweight_km && qzeros_km && scales_km && qw && scales && zeros, ZL_EINVAL);
    zl_w4_layout_t L;
    int st = zl_w4_layout(n, k, g, &L);
    if (st) return st;
    ZL_CHECK_ARG(!row_interleave || n % 2 == 0, ZL_ESHAPE);
    // a group that is not a power of two (G == K case) must not straddle the (q,j) blocks unevenly:
    // (1024q + 256j)/G is then always 0, fine.
    hipLaunchKernelGGL(k_pack_qw, dim3(grid_for((L.np / 2) * L.q * 256)), dim3(kThreads), 0, (hipStream_t)s,
                       qweight_km, qw, n, k / 8, L.np, L.q, row_interleave);
    hipLaunchKernelGGL(k_pack_meta, dim3(grid_for(L.np * L.q * L.c)), dim3(kThreads), 0, (hipStream_t)s, qzeros_km,
                       scales_km, scales, zeros, n, k / g, g, L.np, L.q, L.c, row_interleave);
    return zl_launch_status();
}

int zl_w4_dequant(const uint32_t* qw, const uint16_t* scales, const uint16_t* zeros, int64_t n, int64_t k, int64_t g,
                  uint16_t* out, zl_stream_t s) {
    ZL_CHECK_ARG(qw && scales && zeros && out, ZL_EINVAL);
    zl_w4_layout_t L;
    int st = zl_w4_layout(n, k, g, &L);
    if (st) return st;
    hipLaunchKernelGGL(k_w4_dequant, dim3(grid_for(n * (k / 8))), dim3(kThreads), 0, (hipStream_t)s, qw, scales, zeros,
                       out, n, k, g, L.q, L.c);
    return zl_launch_status();
}

}  // extern "C"

// dense_gemm.hip -- fp16 / bf16 NT GEMM on the matrix cores for M > 4: y = T(alpha * x . W^T + bias), W (N, K).
//
// Replaces functions::Gemm::forward (bm/functions/gemm.cpp:505-599, cuBLASLt) where the decode path meets a
// dense matrix with more than a handful of rows: RawEmbedding::projection (the lm_head,
// src/nn/embedding/embedding.cu:274-289) for decode batches, NormalLinear for unquantised models and prompt
// chunks.  fp32 accumulation, one rounding to T (the reference's HIGH_PRECISION GEMM mode).  M <= 4 stays on
// the wave-per-row GEMV (dense_gemv.hip), which streams at 6.5 TB/s.
//
// Same tiling as w4_gemm_tiled.hip without the dequant: workgroup = 4 waves = BM x 128 outputs, wave = BM x 32
// (two 16-column weight tiles share every activation fragment), K in 128-k chunks; the activation chunk is
// double-buffered in LDS; weight fragments (lane = row n, 8 consecutive k: 16 B) are loaded straight into
// the MFMA B registers two chunks ahead (a 16-row x 128-k tile is 16 x 256 contiguous bytes).
#include <stdlib.h>
#include "zl_common.h"

namespace {

constexpr int kDW = 4, kDT = kDW * 64, kDBN = kDW * 32, kDRow = 128 + 8;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct DenseGemmParams {
    const uint16_t* x;
    int64_t ldx;
    const uint16_t* w;
    const uint16_t* bias;
    uint16_t* y;
    float alpha;
    int m, n, k, groups;
};

template <int DT>
__device__ __forceinline__ f4 mfma16(uint4 a, uint4 b, f4 c) {
    if constexpr (DT == ZL_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}

// NS: weight ring slots = chunks in flight per wave (2 KiB each).  Round 6 measured 4 instead of 2 on the decode batches' lm_head
// (1.05 GB, a pure weight stream at 5.0 TB/s: 208 / 227 us at 8 / 32 rows against 158 for the one-row GEMV): SLOWER, 228 / 250 us
// (tools/ubench/bench_lm_head.py, profiles/r06_lm_head.txt) -- bytes in flight are not what holds it back; a lane's
// 16-byte load is a quarter of a 64-byte run per row (16 rows per instruction, every 128-byte line touched by two instructions),
// which is what the MFMA B layout asks for and what a row-contiguous loader (LDS-DMA, the slab kernel's x path) would not do.  2.
// PACKED: the weights in the ZLD16M layout (zl_dense_pack_m): [N / 16][K / 128][t = 4][lane = 64][8 values] -- a wave's load of one
// (tile, chunk, t) is 1 KiB CONTIGUOUS instead of sixteen 64-byte runs in sixteen rows; what the W4 kernels' ZLW4M layout does for
// int4 weights, for a dense matrix that is streamed with a few rows (the lm_head of a decode batch)
template <int DT, int BM, int NS = 2, bool PACKED = false>
__global__ __launch_bounds__(kDT, 2) void k_dense_gemm(const DenseGemmParams p) {
    constexpr int RB = BM / 16, XR = BM / 16;
    __shared__ __attribute__((aligned(16))) uint16_t xs[2][BM * kDRow];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrow = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * BM;
    const int n_base = blockIdx.x * kDBN + wave * 32;
    const int G = p.groups;

    // weight fragments of chunk g: tile j (rows n_base + 16 j + nrow), step t: 16 B at k = 128 g + 32 t + 8 kq
    int nr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n_base + 16 * j + nrow;
        nr[j] = n < p.n ? n : p.n - 1;
    }
    uint4 wf[NS][2][4];                      // [ring slot][tile][t]
    auto load_w = [&](int slot, int g) {
        const int gc = g < G ? g : G - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (PACKED) {
                const int tile = min((n_base >> 4) + j, (p.n + 15) / 16 - 1);       // (a tile past N: the last one; its columns are never stored)
                const uint4* src = reinterpret_cast<const uint4*>(p.w) + ((size_t)tile * G + gc) * 256 + lane;
#pragma unroll
                for (int t = 0; t < 4; ++t) wf[slot][j][t] = zl_load_nt(src + 64 * t);
            } else {
                const uint16_t* src = p.w + (size_t)nr[j] * p.k + (size_t)gc * 128 + 8 * kq;
#pragma unroll
                for (int t = 0; t < 4; ++t) wf[slot][j][t] = zl_load_nt(reinterpret_cast<const uint4*>(src + 32 * t));
            }
        }
    };
    const int xrow = threadIdx.x >> 4, xcol = (threadIdx.x & 15) * 8;
    uint4 xr[XR];
    auto load_x = [&](int g) {
        const int gc = g < G ? g : G - 1;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int row = m0 + xrow + 16 * r;
            const int rc = row < p.m ? row : p.m - 1;
            xr[r] = *reinterpret_cast<const uint4*>(p.x + (size_t)rc * p.ldx + (size_t)gc * 128 + xcol);
            if (row >= p.m) xr[r] = make_uint4(0, 0, 0, 0);
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int r = 0; r < XR; ++r) *reinterpret_cast<uint4*>(&xs[buf][(xrow + 16 * r) * kDRow + xcol]) = xr[r];
    };

    load_x(0);
#pragma unroll
    for (int u = 0; u < NS; ++u) load_w(u, u);
    store_x(0);
    load_x(1);
    __syncthreads();

    f4 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        acc[rb][0] = (f4){0.f, 0.f, 0.f, 0.f};
        acc[rb][1] = (f4){0.f, 0.f, 0.f, 0.f};
    }

    int g = 0;
#pragma unroll 1
    for (; g < G; g += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (g + u < G) {                 // workgroup-uniform; g a multiple of NS (even): chunk parity = u & 1
                const uint16_t* xb = &xs[u & 1][nrow * kDRow + kq * 8];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const uint4 a = *reinterpret_cast<const uint4*>(xb + rb * 16 * kDRow + t * 32);
                        acc[rb][0] = mfma16<DT>(a, wf[u][0][t], acc[rb][0]);
                        acc[rb][1] = mfma16<DT>(a, wf[u][1][t], acc[rb][1]);
                    }
                }
                load_w(u, g + u + NS);       // the slot just consumed <- NS chunks ahead
                store_x((u & 1) ^ 1);
                load_x(g + u + 2);
                __syncthreads();
            }
        }
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n_base + 16 * j + nrow;
        const float b = (p.bias && n < p.n) ? ZT<DT>::to_f32(p.bias[n]) : 0.f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + rb * 16 + 4 * kq + i;
                if (row < p.m && n < p.n) p.y[(size_t)row * p.n + n] = ZT<DT>::from_f32(p.alpha * acc[rb][j][i] + b);
            }
        }
    }
}

// ZLD16M pack: one thread per 16-byte unit of the output
__global__ __launch_bounds__(256) void k_dense_pack_m(const uint16_t* __restrict__ w, uint4* __restrict__ out, int n, int k, int64_t units) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const int G = k / 128;
    const int lane = (int)(u & 63), t = (int)((u >> 6) & 3);
    const int64_t tg = u >> 8;
    const int g = (int)(tg % G);
    const int64_t tile = tg / G;
    const int64_t row = tile * 16 + (lane & 15);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < n) v = *reinterpret_cast<const uint4*>(w + row * k + (size_t)g * 128 + 32 * t + 8 * (lane >> 4));
    out[u] = v;
}

}  // namespace

extern "C" int64_t zl_dense_m_bytes(int64_t n, int64_t k) {
    if (n <= 0 || k <= 0 || k % 128 != 0) return ZL_ESHAPE;
    return (n + 15) / 16 * 16 * k * 2;
}

extern "C" int zl_dense_pack_m(const uint16_t* w, uint16_t* out, int64_t n, int64_t k, zl_stream_t s) {
    ZL_CHECK_ARG(w && out && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 128 == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0 && n < ((int64_t)1 << 31) && k < ((int64_t)1 << 31), ZL_ESHAPE);
    const int64_t units = (n + 15) / 16 * (k / 128) * 256;
    hipLaunchKernelGGL(k_dense_pack_m, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)s, w, reinterpret_cast<uint4*>(out), (int)n, (int)k, units);
    return zl_launch_status();
}

// zl_gemm_nt on a ZLD16M-packed weight (m <= 32: the decode batches' lm_head; the arithmetic and the order of the sums are
// zl_gemm_nt's, the results bit for bit)
extern "C" int zl_gemm_nt_packed(const uint16_t* x, int64_t ldx, const uint16_t* wp, const uint16_t* bias, uint16_t* y, int64_t m,
                                 int64_t n, int64_t k, float alpha, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && wp && y && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 128 == 0 && ldx % 8 == 0 && ldx >= k && ((uintptr_t)x & 15) == 0 && ((uintptr_t)wp & 15) == 0 && m <= 32, ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    DenseGemmParams p;
    p.x = x; p.ldx = ldx; p.w = wp; p.bias = bias; p.y = y; p.alpha = alpha;
    p.m = (int)m; p.n = (int)n; p.k = (int)k; p.groups = (int)(k / 128);
    const dim3 grid((unsigned)((n + kDBN - 1) / kDBN), 1);
    hipStream_t hs = (hipStream_t)s;
    if (dtype == ZL_F16) {
        if (m <= 16) hipLaunchKernelGGL((k_dense_gemm<ZL_F16, 16, 2, true>), grid, dim3(kDT), 0, hs, p);
        else hipLaunchKernelGGL((k_dense_gemm<ZL_F16, 32, 2, true>), grid, dim3(kDT), 0, hs, p);
    } else {
        if (m <= 16) hipLaunchKernelGGL((k_dense_gemm<ZL_BF16, 16, 2, true>), grid, dim3(kDT), 0, hs, p);
        else hipLaunchKernelGGL((k_dense_gemm<ZL_BF16, 32, 2, true>), grid, dim3(kDT), 0, hs, p);
    }
    return zl_launch_status();
}

extern "C" int zl_gemm_nt(const uint16_t* x, int64_t ldx, const uint16_t* w, const uint16_t* bias, uint16_t* y, int64_t m,
                          int64_t n, int64_t k, float alpha, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && w && y && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 128 == 0 && ldx % 8 == 0 && ldx >= k && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    DenseGemmParams p;
    p.x = x; p.ldx = ldx; p.w = w; p.bias = bias; p.y = y; p.alpha = alpha;
    p.m = (int)m; p.n = (int)n; p.k = (int)k; p.groups = (int)(k / 128);
    const unsigned gx = (unsigned)((n + kDBN - 1) / kDBN);
    hipStream_t hs = (hipStream_t)s;
    const int bm = m <= 16 ? 16 : (m <= 32 ? 32 : 64);
    ZL_CHECK_ARG((m + bm - 1) / bm <= 65535, ZL_ELIMIT);
    const dim3 grid(gx, (unsigned)((m + bm - 1) / bm));
#ifndef ZL_DENSE_NS
#define ZL_DENSE_NS 2        // (tools/ubench/variant.sh ... -DZL_DENSE_NS=4 rebuilds the deeper ring for A/B runs)
#endif
#define ZL_DG(DT)                                                                              \
    if (bm == 16) hipLaunchKernelGGL((k_dense_gemm<DT, 16, ZL_DENSE_NS>), grid, dim3(kDT), 0, hs, p);       \
    else if (bm == 32) hipLaunchKernelGGL((k_dense_gemm<DT, 32, ZL_DENSE_NS>), grid, dim3(kDT), 0, hs, p);  \
    else hipLaunchKernelGGL((k_dense_gemm<DT, 64>), grid, dim3(kDT), 0, hs, p);
    if (dtype == ZL_F16) { ZL_DG(ZL_F16) } else { ZL_DG(ZL_BF16) }
#undef ZL_DG
    return zl_launch_status();
}

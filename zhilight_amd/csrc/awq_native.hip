// awq_native.hip -- AWQ checkpoints consumed in their on-disk layout, no repack:
//     qweight (K, N/8) int32, qzeros (K/G, N/8) int32, scales (K/G, N) fp16; nibble i of a word = column 8c + {0,2,4,6,1,3,5,7}[i]
// Replaces nn::awq::awq_gemm / awq_dequantize (src/nn/quant/awq/awq.h:10-25; gemm_forward_4bit_cuda_m16nXk32,
// gemm_kernels.cu:32-275; dequantize_weights :277-330; KERNEL_sum_dim0 :381-400; dequantize_s4_to_fp16x2, dequantize.cuh:45-112).
//
// Arithmetic restated: W16[k,n] = rn16(fp16(q - z) * s) -- exact difference, ONE rounding of the product; the
// 32-row K tiles are dealt round-robin to split_k_iters splits (tile t = i * split_k_iters + z); a split accumulates
// x * W16 in fp32 and leaves its partial C as FP16; the partials are summed in fp32 in split order and rounded to fp16.
//
// Why a VALU kernel: the layout is N-contiguous -- a word holds 8 COLUMNS of one k -- while every MFMA operand wants 8
// consecutive k per lane; feeding the matrix cores would need an 8 x 8 nibble transpose per word across lanes.  Kept
// instead: the lane that loads a word owns its 8 columns and walks down K (outer-product form): 4 v_and_or + 1 shift
// (the interleave {0,4,1,5,2,6,3,7} is exactly the order the 0x6400 pair trick extracts), 4 packed subtracts, 4 packed
// multiplies with the scales, 8 v_fma_mix_f32 per activation row -- 21 VALU per word at one row, about the HBM rate of a
// CU.  This is the format-faithful route (ZL_AWQ_NATIVE / QuantConfig.awq_native); the default AWQ route re-tiles the
// checkpoint once at load into ZLW4M and runs the matrix-core kernels (llama.py).
#include "zl_common.h"

namespace {

typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask, uint32_t magic) { return (w & mask) | magic; }

// the 8 values of a word in COLUMN order as four fp16 pairs, zero points subtracted: exact (integers < 2048)
struct Deq8 { hv2 p[4]; };
__device__ __forceinline__ Deq8 awq_sub(uint32_t w, uint32_t z) {
    const hv2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
    const hv2 q0 = __builtin_bit_cast(hv2, and_or(w, 0x000f000fu, 0x64006400u)), z0 = __builtin_bit_cast(hv2, and_or(z, 0x000f000fu, 0x64006400u));
    const hv2 q1 = __builtin_bit_cast(hv2, and_or(w, 0x00f000f0u, 0x64006400u)), z1 = __builtin_bit_cast(hv2, and_or(z, 0x00f000f0u, 0x64006400u));
    const uint32_t wb = w >> 8, zb = z >> 8;
    const hv2 q2 = __builtin_bit_cast(hv2, and_or(wb, 0x000f000fu, 0x64006400u)), z2 = __builtin_bit_cast(hv2, and_or(zb, 0x000f000fu, 0x64006400u));
    const hv2 q3 = __builtin_bit_cast(hv2, and_or(wb, 0x00f000f0u, 0x64006400u)), z3 = __builtin_bit_cast(hv2, and_or(zb, 0x00f000f0u, 0x64006400u));
    Deq8 d;
    d.p[0] = q0 - z0;                         // (1024 + q) - (1024 + z)
    d.p[1] = (q1 - z1) * sixteenth;           // (1024 + 16 q) - (1024 + 16 z) = 16 (q - z): exact, then exact / 16
    d.p[2] = q2 - z2;
    d.p[3] = (q3 - z3) * sixteenth;
    return d;
}

// ---- dequantize_weights: (K, N/8) -> W16 (K, N) fp16 ------------------------------------------------------------------
__global__ void k_awq_dequant(const uint32_t* __restrict__ qweight, const uint32_t* __restrict__ qzeros,
                              const uint16_t* __restrict__ scales, uint16_t* __restrict__ out, int64_t k, int64_t n8, int64_t g) {
    const int64_t total = k * n8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t kk = i / n8, c = i % n8, grp = kk / g;
        const Deq8 d = awq_sub(qweight[i], qzeros[grp * n8 + c]);
        const uint4 sv = *reinterpret_cast<const uint4*>(scales + (grp * n8 + c) * 8);
        const uint32_t su[4] = {sv.x, sv.y, sv.z, sv.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __builtin_bit_cast(uint32_t, d.p[j] * __builtin_bit_cast(hv2, su[j]));   // one rounding
        *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- split-K GEMM, M <= kMR activation rows per pass --------------------------------------------------------------------
// grid (column blocks of 64 words, split z), block 256 = 4 waves: wave w takes every 4th of split z's tiles; lane = one
// word column (8 output columns).  fp32 accumulators of the 4 waves meet in LDS (fixed order), the split's partial is
// rounded to fp16 like the reference's.
constexpr int kMR = 8;
struct AwqParams {
    const uint16_t* x;           // (M, K) fp16
    int64_t ldx;
    const uint32_t* qweight;     // (K, N/8)
    const uint32_t* qzeros;      // (K/G, N/8)
    const uint16_t* scales;      // (K/G, N)
    uint16_t* partial;           // (splits, M, N) fp16
    int m, n8, k, g, splits, tiles;
};

template <int MR>
__global__ __launch_bounds__(256) void k_awq_gemm_partial(const AwqParams p) {
    __shared__ float red[4][MR][8][64];          // 64 KB at MR = 8
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int z = blockIdx.y;
    const bool live = c < p.n8;
    const int cc = live ? c : 0;
    float acc[MR][8];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
    for (int t = z + wave * p.splits; t < p.tiles; t += 4 * p.splits) {
        const int k0 = t * 32, grp = k0 / p.g;
        const uint32_t zw = p.qzeros[(size_t)grp * p.n8 + cc];
        const uint4 sv = *reinterpret_cast<const uint4*>(p.scales + ((size_t)grp * p.n8 + cc) * 8);
        const uint32_t su[4] = {sv.x, sv.y, sv.z, sv.w};
        uint32_t w[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) w[r] = (k0 + r < p.k) ? __builtin_nontemporal_load(p.qweight + (size_t)(k0 + r) * p.n8 + cc) : zw;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const Deq8 d = awq_sub(w[r], zw);
            hv2 wv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = d.p[j] * __builtin_bit_cast(hv2, su[j]);      // W16, one rounding
            const int kk = k0 + r < p.k ? k0 + r : 0;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m < p.m) {                                                                // workgroup-uniform
                    const float xv = k0 + r < p.k ? (float)__builtin_bit_cast(_Float16, p.x[(size_t)m * p.ldx + kk]) : 0.f;   // wave-uniform load
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[m][2 * j] = __builtin_fmaf(xv, (float)wv[j].x, acc[m][2 * j]);
                        acc[m][2 * j + 1] = __builtin_fmaf(xv, (float)wv[j].y, acc[m][2 * j + 1]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][m][j][lane] = acc[m][j];
    __syncthreads();
    // 4 waves x (m, j) pairs: thread (wave, lane) sums rows m = wave, wave + 4 of its column word
    for (int m = wave; m < p.m; m += 4) {
        if (!live) continue;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                a += red[w4][m][2 * j][lane];
                b += red[w4][m][2 * j + 1][lane];
            }
            hv2 h;
            h.x = zl_f32_to_f16(a);
            h.y = zl_f32_to_f16(b);
            o[j] = __builtin_bit_cast(uint32_t, h);
        }
        *reinterpret_cast<uint4*>(p.partial + (((size_t)z * p.m + m) * p.n8 + c) * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// KERNEL_sum_dim0: out[i] = half( sum_z float(partial[z][i]) ), z ascending
__global__ void k_awq_sum_splits(const uint16_t* __restrict__ partial, uint16_t* __restrict__ out, int splits, int64_t count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        float r = 0.f;
        for (int z = 0; z < splits; ++z) r += (float)__builtin_bit_cast(_Float16, partial[(size_t)z * count + i]);
        out[i] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(r));
    }
}

inline unsigned grid_of(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    return (unsigned)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int zl_awq_dequantize(const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales, uint16_t* out, int64_t k, int64_t n,
                      int64_t group_size, zl_stream_t s) {
    ZL_CHECK_ARG(qweight && qzeros && scales && out && k > 0 && n > 0 && group_size > 0, ZL_EINVAL);
    ZL_CHECK_ARG(n % 8 == 0 && k % group_size == 0 && ((uintptr_t)scales & 15) == 0 && ((uintptr_t)out & 15) == 0, ZL_ESHAPE);
    hipLaunchKernelGGL(k_awq_dequant, dim3(grid_of(k * (n / 8), 256)), dim3(256), 0, (hipStream_t)s, qweight, qzeros, scales, out, k,
                       n / 8, group_size);
    return zl_launch_status();
}

int64_t zl_awq_gemm_workspace_bytes(int64_t m, int64_t n, int64_t split_k_iters) {
    if (m <= 0 || n <= 0 || split_k_iters <= 0) return ZL_EINVAL;
    return split_k_iters * m * n * 2;
}

int zl_awq_gemm(const uint16_t* x, int64_t ldx, const uint32_t* qweight, const uint32_t* qzeros, const uint16_t* scales, uint16_t* y,
                void* workspace, int64_t m, int64_t n, int64_t k, int64_t group_size, int64_t split_k_iters, zl_stream_t s) {
    ZL_CHECK_ARG(x && qweight && qzeros && scales && y && workspace && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    // the reference's own conditions (gemm_kernels.cu:426-433): OC % 64, group_size % 32
    ZL_CHECK_ARG(n % 64 == 0 && group_size % 32 == 0 && k % group_size == 0 && ldx >= k, ZL_ESHAPE);
    ZL_CHECK_ARG(split_k_iters >= 1 && split_k_iters <= 65535 && ((uintptr_t)scales & 15) == 0, ZL_ESHAPE);
    hipStream_t hs = (hipStream_t)s;
    for (int64_t m0 = 0; m0 < m; m0 += kMR) {   // kMR activation rows per weight pass
        const int mm = (int)(m - m0 < kMR ? m - m0 : kMR);
        AwqParams p;
        p.x = x + m0 * ldx; p.ldx = ldx; p.qweight = qweight; p.qzeros = qzeros; p.scales = scales;
        p.partial = reinterpret_cast<uint16_t*>(workspace);
        p.m = mm; p.n8 = (int)(n / 8); p.k = (int)k; p.g = (int)group_size; p.splits = (int)split_k_iters;
        p.tiles = (int)((k + 31) / 32);
        const dim3 grid((unsigned)((p.n8 + 63) / 64), (unsigned)split_k_iters);
        if (mm <= 1) hipLaunchKernelGGL(k_awq_gemm_partial<1>, grid, dim3(256), 0, hs, p);
        else if (mm <= 4) hipLaunchKernelGGL(k_awq_gemm_partial<4>, grid, dim3(256), 0, hs, p);
        else hipLaunchKernelGGL(k_awq_gemm_partial<kMR>, grid, dim3(256), 0, hs, p);
        int st = zl_launch_status();
        if (st) return st;
        const int64_t count = (int64_t)mm * n;
        hipLaunchKernelGGL(k_awq_sum_splits, dim3(grid_of(count, 256)), dim3(256), 0, hs, p.partial, y + m0 * n, (int)split_k_iters, count);
        st = zl_launch_status();
        if (st) return st;
    }
    return ZL_OK;
}

}  // extern "C"

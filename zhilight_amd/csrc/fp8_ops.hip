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

// ---- f4, first part: the FP8 128x128-block linear of config 5 (src/nn/linear/linear.cpp:1697-1950) ------------------------------
__device__ __forceinline__ float e4m3_to_f32(uint32_t c) {
    const uint32_t e = (c >> 3) & 15u, m = c & 7u;
    // normal: (1 + m/8) 2^(e-7) = bits ((e + 120) << 23) | (m << 20); subnormal: m 2^-9
    const float v = e ? __builtin_bit_cast(float, ((e + 120u) << 23) | (m << 20)) : (float)m * 0.001953125f;
    return (c & 0x80u) ? -v : v;
}

// KERNEL_per_token_cast_to_fp8 (fp8_util.cu:229-275): one wave per (row, 128-column block), two elements per lane
template <int DT>
__global__ __launch_bounds__(64) void k_fp8_per_token_cast(const uint16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ out, int64_t ld_out,
                                                           float* __restrict__ scale, int64_t aligned_m, int nblocks, int col_major, float max_e4m3) {
    const int64_t row = blockIdx.x;                   // rows on grid.x (2^31 - 1 of them; the 128-column blocks on grid.y: <= 65535 = 8 M columns)
    const int blk = blockIdx.y, lane = threadIdx.x;
    const uint32_t two = *reinterpret_cast<const uint32_t*>(x + row * ldx + blk * 128 + lane * 2);
    const float f0 = ZT<DT>::to_f32((uint16_t)(two & 0xffffu)), f1 = ZT<DT>::to_f32((uint16_t)(two >> 16));
    float amax = zl_wave_max(fmaxf(fabsf(f0), fabsf(f1)));
    if (amax < 1e-4f) amax = 1e-4f;
    const float mul = max_e4m3 / amax;
    float p0 = f0 * mul, p1 = f1 * mul;
    asm volatile("" : "+v"(p0), "+v"(p1));           // the fp32 products are materialised (no contraction into the cast)
    const uint16_t codes = (uint16_t)f32_to_e4m3(p0) | (uint16_t)((uint16_t)f32_to_e4m3(p1) << 8);
    *reinterpret_cast<uint16_t*>(out + row * ld_out + blk * 128 + lane * 2) = codes;
    if (lane == 0) scale[col_major ? (int64_t)blk * aligned_m + row : row * nblocks + blk] = amax / max_e4m3;
}

// KERNEL_dequant_fp8_block (fp8_util.cu:325-357)
template <int DT>
__global__ __launch_bounds__(256) void k_fp8_block_dequant(const uint8_t* __restrict__ w, const float* __restrict__ scale,
                                                            uint16_t* __restrict__ out, int64_t rows, int64_t cols, int64_t stride_scale) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols, c = i - r * cols;
        float v = e4m3_to_f32(w[i]) * scale[(r / 128) * stride_scale + c / 128];
        asm volatile("" : "+v"(v));
        out[i] = ZT<DT>::from_f32(v);
    }
}

// The block-scaled product of deep_gemm_fp8_block_h20_group (3rd/deep_gemm/deep_gemm_api.h): per 128-k block the fp8 x fp8
// partial product is accumulated in fp32 on the matrix cores (4 x v_mfma_f32_16x16x32_fp8_fp8 from zero), then added to the
// running sum scaled by sa[kb, m] * sw[n / 128, kb].
//
// Shape of the kernel (decode row counts: the weights are the bytes).  A workgroup = 8 waves = 16 weight rows x the whole K: wave w
// takes the 128-k blocks w, w + 8, ... (N / 16 workgroups x 8 waves of independent streams -- the first version's one wave per
// 16 rows walked K with 2 KiB in flight: 0.1-0.2 TB/s), U blocks at a time with every load of the U blocks issued before the
// first MFMA (weight fragments straight from global memory, non-temporal; a row's 128 bytes of a block are two 16-byte lane loads),
// MT tiles of 16 activation rows share each weight fragment; the eight partial tiles meet in LDS and are summed in wave order
// (deterministic).  A tile of 16 rows takes the expert (weight matrix) of its first row; rows of the tile carrying another index
// are not written (contiguous grouped layout, groups aligned to 16 rows; DeepGEMM's own contract is alignment to its block_m =
// 64); the grouped form runs with MT = 1 (neighbouring tiles may belong to different experts).
// PACKED (round 6): the weights in the ZLF8M layout (zl_fp8_block_pack): [group][N / 16][K / 128][2][64 lanes][16 codes] -- a wave's
// fragment load of one (tile, block, half) is 1 KiB contiguous instead of sixteen 64-byte runs in sixteen rows
template <int MT, int U, int DT, bool PACKED = false>
__global__ __launch_bounds__(512) void k_fp8_block_gemm(const uint8_t* __restrict__ a, const float* __restrict__ sa, int64_t aligned_m,
                                                        const uint8_t* __restrict__ w, const float* __restrict__ sw,
                                                        const int32_t* __restrict__ m_indices, uint16_t* __restrict__ c, int m, int n, int k,
                                                        int num_groups) {
    __shared__ __attribute__((aligned(16))) float red[8][MT][256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (MT * 16);
    const int col = lane & 15, kq = lane >> 4;
    const int kb_n = k / 128, nbw = (n + 127) / 128;
    const int g = m_indices ? m_indices[m0] : 0;              // workgroup-uniform (m0 < m by the grid)
    if (g < 0 || g >= num_groups) return;                     // a padding tile of the grouped layout (or a foreign index): nothing is read or written
    // (clamped addresses: a column / row past the end re-reads the last one and is never stored)
    const uint8_t* brow = PACKED ? w + (((size_t)g * ((n + 15) / 16) + blockIdx.x) * kb_n) * 2048 + lane * 16
                                 : w + ((size_t)g * n + min(n0 + col, n - 1)) * k + 16 * kq;
    constexpr int kBlk = PACKED ? 2048 : 128, kHalf = PACKED ? 1024 : 64;       // byte strides of a k-block / of its second half
    const float* swg = sw + ((size_t)g * nbw + n0 / 128) * kb_n;
    const uint8_t* arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) arow[t] = a + (size_t)min(m0 + 16 * t + col, m - 1) * k + 16 * kq;
    f4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};

    for (int kb0 = wave; kb0 < kb_n; kb0 += 8 * U) {
        uint4 bf[U][2], af[U][MT][2];
        float sc[U][MT][4], wsc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kb = kb0 + 8 * u;
            const bool ok = kb < kb_n;                        // wave-uniform; a block past the end re-reads block kb0 with a zero scale
            const int kbc = ok ? kb : kb0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf[u][h] = zl_load_nt(reinterpret_cast<const uint4*>(brow + (size_t)kbc * kBlk + h * kHalf));
#pragma unroll
                for (int t = 0; t < MT; ++t) af[u][t][h] = *reinterpret_cast<const uint4*>(arow[t] + kbc * 128 + h * 64);
            }
            wsc[u] = ok ? swg[kbc] : 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + 16 * t + 4 * kq + i;
                    sc[u][t][i] = row < m ? sa[(size_t)kbc * aligned_m + row] : 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                f4 blk = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint4 b4 = bf[u][h], a4 = af[u][t][h];
                    const long b_lo = (long)(((unsigned long long)b4.y << 32) | b4.x), b_hi = (long)(((unsigned long long)b4.w << 32) | b4.z);
                    const long a_lo = (long)(((unsigned long long)a4.y << 32) | a4.x), a_hi = (long)(((unsigned long long)a4.w << 32) | a4.z);
                    blk = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_lo, b_lo, blk, 0, 0, 0);
                    blk = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_hi, b_hi, blk, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_fmaf(blk[i], sc[u][t][i] * wsc[u], acc[t][i]);
            }
        }
    }
    // ---- the eight k-slices meet in LDS: [wave][tile][C row 4 kq + i][column]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][t][(4 * kq + i) * 16 + col] = acc[t][i];
    __syncthreads();
    for (int idx = threadIdx.x; idx < MT * 256; idx += 512) {
        const int t = idx >> 8, e = idx & 255, row = m0 + 16 * t + (e >> 4), cc = n0 + (e & 15);
        float v = red[0][t][e];
#pragma unroll
        for (int ws = 1; ws < 8; ++ws) v += red[ws][t][e];
        if (row < m && cc < n && (!m_indices || m_indices[row] == g)) c[(size_t)row * n + cc] = ZT<DT>::from_f32(v);
    }
}

// 17..32 rows (round 6): k_fp8_block_gemm<2, ..> loads its activation fragments straight from global memory -- 16 bytes out of 16
// different rows per 16 lanes, the access pattern that cost the first cut of w4_slab.hip 5..8 us per launch -- and every workgroup
// pulls ALL of the activations (32 x K bytes: twice its weight bytes).  Here a wave moves the 32 rows x 128 bytes of ITS next block
// global -> LDS by LDS-DMA (row-contiguous; 16-byte pieces XOR-swizzled by the row so the ds_read_b128 fragment reads are
// conflict-free), one block ahead, next to the weight fragments and the per-token scales of that block; the arithmetic, the order
// of a wave's blocks and the cross-wave sum are k_fp8_block_gemm's: the same bits.
typedef __attribute__((address_space(3))) void* fp8_lds_ptr;
constexpr int fp8_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }
// NB = blocks per wave = ceil(K / 1024) (compile-time: the block loop is unrolled, LDS sets and registers statically indexed).
// Issue order: the weight fragments and scales of ALL the wave's blocks first (the HBM stream: 2 KB per block and wave in flight,
// as in k_fp8_block_gemm<.., U = NB>), then the activation images of the first three blocks (L2); image i + 3 is requested when
// block i has been used.  In-order return makes "image i has landed" = "at most the two younger images outstanding".
template <int DT, int NB, bool PACKED = false>
__global__ __launch_bounds__(512, 1) void k_fp8_block_gemm_dma(const uint8_t* __restrict__ a, const float* __restrict__ sa, int64_t aligned_m,
                                                               const uint8_t* __restrict__ w, const float* __restrict__ sw,
                                                               uint16_t* __restrict__ c, int m, int n, int k) {
    constexpr int MT = 2, SETS = NB < 3 ? NB : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];       // [8 waves][SETS][32 rows][128 B], then red[8][MT][256]
    float* red = reinterpret_cast<float*>(fsm + 8 * SETS * 4096);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 32;
    const int col = lane & 15, kq = lane >> 4;
    const int kb_n = k / 128;
    const uint8_t* brow = PACKED ? w + ((size_t)blockIdx.x * kb_n) * 2048 + lane * 16 : w + (size_t)min(n0 + col, n - 1) * k + 16 * kq;
    constexpr int kBlk = PACKED ? 2048 : 128, kHalf = PACKED ? 1024 : 64;
    const float* swg = sw + (size_t)(n0 / 128) * kb_n;
    unsigned char* region = fsm + wave * (SETS * 4096);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a), 0, (uint32_t)((size_t)m * k), 0x00020000);
    // DMA d of a block: lane -> (row 8 d + (lane >> 3), LDS piece lane & 7), which holds the row's piece (lane & 7) ^ (row & 7)
    uint32_t x_off[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int row = 8 * d + (lane >> 3);
        x_off[d] = (uint32_t)min(m0 + row, m - 1) * (uint32_t)k + (uint32_t)(((lane & 7) ^ (row & 7)) * 16);
    }
    // block i of this wave = k-block wave + 8 i; past the end of K (the last round of a ragged K): block `wave` again with a zero
    // scale (real bytes: an e4m3 NaN pattern out of stale LDS would survive the zero)
    uint4 bf[NB][2];
    float sc[NB][MT][4], wsc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const bool ok = wave + 8 * i < kb_n;
        const int kb = ok ? wave + 8 * i : min(wave, kb_n - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) bf[i][h] = zl_load_nt(reinterpret_cast<const uint4*>(brow + (size_t)kb * kBlk + h * kHalf));
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) sc[i][t][q] = sa[(size_t)kb * aligned_m + min(m0 + 16 * t + 4 * kq + q, m - 1)];
        wsc[i] = ok ? swg[kb] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    auto dma = [&](int i) {                                                    // static i
        const int kb = wave + 8 * i < kb_n ? wave + 8 * i : min(wave, kb_n - 1);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (fp8_lds_ptr)(region + (i % SETS) * 4096 + d * 1024), 16, x_off[d], kb * 128, 0, 0);
#endif
        }
    };
#pragma unroll
    for (int i = 0; i < SETS; ++i) dma(i);
    __builtin_amdgcn_sched_barrier(0);
    f4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
    const uint32_t rbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)region;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        // images younger than image i that may still be outstanding: i + 1 .. min(i + SETS - 1, NB - 1) (image i + SETS goes out below)
        const int younger = (i + SETS - 1 < NB - 1 ? i + SETS - 1 : NB - 1) - i;
        if (younger >= 2) __builtin_amdgcn_s_waitcnt(fp8_vmcnt(8));
        else if (younger == 1) __builtin_amdgcn_s_waitcnt(fp8_vmcnt(4));
        else __builtin_amdgcn_s_waitcnt(fp8_vmcnt(0));
        uint4 af[MT][2];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t addr = rbase + (uint32_t)((i % SETS) * 4096 + (16 * t + col) * 128 + (((kq + 4 * h) ^ (col & 7)) * 16));
                asm volatile("ds_read_b128 %0, %1" : "=v"(af[t][h]) : "v"(addr) : "memory");
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + SETS < NB) dma(i + SETS);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f4 blk = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 b4 = bf[i][h], a4 = af[t][h];
                const long b_lo = (long)(((unsigned long long)b4.y << 32) | b4.x), b_hi = (long)(((unsigned long long)b4.w << 32) | b4.z);
                const long a_lo = (long)(((unsigned long long)a4.y << 32) | a4.x), a_hi = (long)(((unsigned long long)a4.w << 32) | a4.z);
                blk = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_lo, b_lo, blk, 0, 0, 0);
                blk = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_hi, b_hi, blk, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float s_row = (m0 + 16 * t + 4 * kq + q) < m ? sc[i][t][q] : 0.f;
                acc[t][q] = __builtin_fmaf(blk[q], s_row * wsc[i], acc[t][q]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(wave * MT + t) * 256 + (4 * kq + q) * 16 + col] = acc[t][q];
    __syncthreads();
    for (int idx = threadIdx.x; idx < MT * 256; idx += 512) {
        const int t = idx >> 8, e = idx & 255, row = m0 + 16 * t + (e >> 4), cc = n0 + (e & 15);
        float v = red[t * 256 + e];
#pragma unroll
        for (int ws = 1; ws < 8; ++ws) v += red[(ws * MT + t) * 256 + e];
        if (row < m && cc < n) c[(size_t)row * n + cc] = ZT<DT>::from_f32(v);
    }
}

// ZLF8M pack: one thread per 16-byte unit of the output
__global__ __launch_bounds__(256) void k_fp8_block_pack(const uint8_t* __restrict__ w, uint4* __restrict__ out, int n, int k, int64_t units) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const int kb_n = k / 128, tiles = (n + 15) / 16;
    const int lane = (int)(u & 63), h = (int)((u >> 6) & 1);
    const int64_t tk = u >> 7;
    const int kb = (int)(tk % kb_n);
    const int64_t gt = tk / kb_n;
    const int tile = (int)(gt % tiles);
    const int64_t g = gt / tiles;
    const int row = tile * 16 + (lane & 15);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < n) v = *reinterpret_cast<const uint4*>(w + ((size_t)g * n + row) * k + (size_t)kb * 128 + 64 * h + 16 * (lane >> 4));
    out[u] = v;
}

}  // namespace

#define ZL_DT_SWITCH(dtype, EXPR_F16, EXPR_BF16) \
    if ((dtype) == ZL_F16) { EXPR_F16; } else if ((dtype) == ZL_BF16) { EXPR_BF16; } else return ZL_EDTYPE;

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

int zl_fp8_per_token_cast(const uint16_t* x, int64_t ldx, uint8_t* out, int64_t ld_out, float* scale, int64_t aligned_m, int64_t m, int64_t n,
                          int scale_col_major, float max_e4m3, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && out && scale && m > 0 && n > 0 && max_e4m3 > 0.f, ZL_EINVAL);
    ZL_CHECK_ARG(n % 128 == 0 && ldx >= n && ld_out >= n && ldx % 2 == 0 && ld_out % 2 == 0 && aligned_m >= m, ZL_ESHAPE);
    ZL_CHECK_ARG(m < ((int64_t)1 << 31) && n / 128 <= 65535, ZL_ELIMIT);        // (the reference launches grid (m, n / 128) too: fp8_util.cu:277-322)
    const dim3 grid((unsigned)m, (unsigned)(n / 128));
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_fp8_per_token_cast<ZL_F16>, grid, dim3(64), 0, (hipStream_t)s, x, ldx, out, ld_out, scale, aligned_m, (int)(n / 128), scale_col_major, max_e4m3),
        hipLaunchKernelGGL(k_fp8_per_token_cast<ZL_BF16>, grid, dim3(64), 0, (hipStream_t)s, x, ldx, out, ld_out, scale, aligned_m, (int)(n / 128), scale_col_major, max_e4m3))
    return zl_launch_status();
}

int zl_fp8_block_dequant(const uint8_t* w, const float* scale, uint16_t* out, int64_t rows, int64_t cols, int64_t stride_scale, int dtype,
                         zl_stream_t s) {
    ZL_CHECK_ARG(w && scale && out && rows > 0 && cols > 0 && stride_scale * 128 >= cols, ZL_EINVAL);
    int64_t g = (rows * cols + 255) / 256;
    if (g > 65535 * 16) g = 65535 * 16;
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_fp8_block_dequant<ZL_F16>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)s, w, scale, out, rows, cols, stride_scale),
        hipLaunchKernelGGL(k_fp8_block_dequant<ZL_BF16>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)s, w, scale, out, rows, cols, stride_scale))
    return zl_launch_status();
}

int64_t zl_fp8_block_packed_bytes(int64_t n, int64_t k, int64_t num_groups) {
    if (n <= 0 || k <= 0 || k % 128 != 0 || num_groups < 1) return ZL_ESHAPE;
    return num_groups * ((n + 15) / 16 * 16) * k;
}

int zl_fp8_block_pack(const uint8_t* w, uint8_t* out, int64_t n, int64_t k, int64_t num_groups, zl_stream_t s) {
    ZL_CHECK_ARG(w && out && n > 0 && k > 0 && num_groups >= 1, ZL_EINVAL);
    ZL_CHECK_ARG(k % 128 == 0 && (((uintptr_t)w | (uintptr_t)out) & 15) == 0 && n < ((int64_t)1 << 31) && k < ((int64_t)1 << 31), ZL_ESHAPE);
    const int64_t units = num_groups * ((n + 15) / 16) * (k / 128) * 128;
    ZL_CHECK_ARG((units + 255) / 256 < ((int64_t)1 << 31), ZL_ELIMIT);
    hipLaunchKernelGGL(k_fp8_block_pack, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)s, w, reinterpret_cast<uint4*>(out), (int)n, (int)k, units);
    return zl_launch_status();
}

static int fp8_block_gemm_impl(const uint8_t* lhs, const float* lhs_scales, int64_t aligned_m, const uint8_t* rhs, const float* rhs_scales,
                               const int32_t* m_indices, uint16_t* out, int64_t m, int64_t n, int64_t k, int num_groups, int dtype, bool packed,
                               zl_stream_t s);

int zl_fp8_block_gemm_group(const uint8_t* lhs, const float* lhs_scales, int64_t aligned_m, const uint8_t* rhs, const float* rhs_scales,
                            const int32_t* m_indices, uint16_t* out, int64_t m, int64_t n, int64_t k, int num_groups, int dtype, zl_stream_t s) {
    return fp8_block_gemm_impl(lhs, lhs_scales, aligned_m, rhs, rhs_scales, m_indices, out, m, n, k, num_groups, dtype, false, s);
}

// ... on a ZLF8M-packed weight (zl_fp8_block_pack; up to 32 rows per launch or the grouped form: the decode shapes): the same bits
int zl_fp8_block_gemm_group_packed(const uint8_t* lhs, const float* lhs_scales, int64_t aligned_m, const uint8_t* rhs_packed, const float* rhs_scales,
                                   const int32_t* m_indices, uint16_t* out, int64_t m, int64_t n, int64_t k, int num_groups, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(m_indices || m <= 32, ZL_ESHAPE);
    return fp8_block_gemm_impl(lhs, lhs_scales, aligned_m, rhs_packed, rhs_scales, m_indices, out, m, n, k, num_groups, dtype, true, s);
}

static int fp8_block_gemm_impl(const uint8_t* lhs, const float* lhs_scales, int64_t aligned_m, const uint8_t* rhs, const float* rhs_scales,
                               const int32_t* m_indices, uint16_t* out, int64_t m, int64_t n, int64_t k, int num_groups, int dtype, bool packed,
                               zl_stream_t s) {
    ZL_CHECK_ARG(lhs && lhs_scales && rhs && rhs_scales && out && m > 0 && n > 0 && k > 0 && num_groups >= 1, ZL_EINVAL);
    ZL_CHECK_ARG(k % 128 == 0 && aligned_m >= m && (num_groups == 1 || m_indices), ZL_ESHAPE);
    ZL_CHECK_ARG((((uintptr_t)lhs | (uintptr_t)rhs) & 15) == 0, ZL_ESHAPE);      // 16-byte fragment loads
    ZL_CHECK_ARG(m < ((int64_t)1 << 31) && n < ((int64_t)1 << 31) && (m + 15) / 16 <= 65535, ZL_ELIMIT);
    hipStream_t hs = (hipStream_t)s;
    const unsigned gx = (unsigned)((n + 15) / 16);
#define ZL_FP8B_(MT_, U_, P_)                                                                                                              \
    {                                                                                                                                      \
        const dim3 grid(gx, (unsigned)((m + 16 * MT_ - 1) / (16 * MT_)));                                                                  \
        ZL_DT_SWITCH(dtype,                                                                                                                \
            hipLaunchKernelGGL((k_fp8_block_gemm<MT_, U_, ZL_F16, P_>), grid, dim3(512), 0, hs, lhs, lhs_scales, aligned_m, rhs, rhs_scales, \
                               m_indices, out, (int)m, (int)n, (int)k, num_groups),                                                        \
            hipLaunchKernelGGL((k_fp8_block_gemm<MT_, U_, ZL_BF16, P_>), grid, dim3(512), 0, hs, lhs, lhs_scales, aligned_m, rhs, rhs_scales, \
                               m_indices, out, (int)m, (int)n, (int)k, num_groups))                                                        \
    }
#define ZL_FP8B(MT_, U_) { if (packed) ZL_FP8B_(MT_, U_, true) else ZL_FP8B_(MT_, U_, false) }
    // the grouped form: one 16-row tile per workgroup (neighbouring tiles may belong to different experts)
    // U = blocks a wave has in flight per round: a wave owns ceil(K / 1024) blocks, and every round pays a full memory latency, so the
    // decode shapes take them all at once when the registers allow (K = 7168: 7 blocks per wave, one round instead of two / four)
    const int64_t bpw = (k / 128 + 7) / 8;
    if (m_indices || m <= 16) {
        if (bpw > 4) ZL_FP8B(1, 8)
        else ZL_FP8B(1, 4)
    } else if (m <= 32) {
#ifndef ZL_FP8_NO_DMA
        // 17..32 rows: activations by LDS-DMA (k_fp8_block_gemm_dma; same bits) for the block counts per wave it is built for
        if ((int64_t)m * k < ((int64_t)1 << 31) && (bpw == 1 || bpw == 2 || bpw == 4 || bpw == 7 || bpw == 8)) {
            const dim3 grid(gx, (unsigned)((m + 31) / 32));
#define ZL_FP8D_(NB_, P_)                                                                                                                   \
            {                                                                                                                                \
                const int lds = 8 * (NB_ < 3 ? NB_ : 3) * 4096 + 8 * 2 * 256 * 4;                                                           \
                const void* fn = dtype == ZL_F16 ? reinterpret_cast<const void*>(&k_fp8_block_gemm_dma<ZL_F16, NB_, P_>)                    \
                                                 : reinterpret_cast<const void*>(&k_fp8_block_gemm_dma<ZL_BF16, NB_, P_>);                  \
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return ZL_ELIMIT;               \
                ZL_DT_SWITCH(dtype,                                                                                                          \
                    hipLaunchKernelGGL((k_fp8_block_gemm_dma<ZL_F16, NB_, P_>), grid, dim3(512), lds, hs, lhs, lhs_scales, aligned_m, rhs, rhs_scales, out, (int)m, (int)n, (int)k), \
                    hipLaunchKernelGGL((k_fp8_block_gemm_dma<ZL_BF16, NB_, P_>), grid, dim3(512), lds, hs, lhs, lhs_scales, aligned_m, rhs, rhs_scales, out, (int)m, (int)n, (int)k)) \
            }
#define ZL_FP8D(NB_) { if (packed) ZL_FP8D_(NB_, true) else ZL_FP8D_(NB_, false) }
            switch ((int)bpw) {
                case 1: ZL_FP8D(1) break;
                case 2: ZL_FP8D(2) break;
                case 4: ZL_FP8D(4) break;
                case 7: ZL_FP8D(7) break;
                default: ZL_FP8D(8) break;
            }
#undef ZL_FP8D
#undef ZL_FP8D_
            return zl_launch_status();
        }
#endif
        if (bpw > 2) ZL_FP8B(2, 4)
        else ZL_FP8B(2, 2)
    } else ZL_FP8B(4, 1)
#undef ZL_FP8B
#undef ZL_FP8B_
    return zl_launch_status();
}

}  // extern "C"

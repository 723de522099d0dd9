// w4_gemm_tiled.hip -- W4A16 GEMM for M > 16 (prefill chunks, large decode batches) on the matrix cores.
//
// Reference: the M > 40 branch of gptq_gemm_k_major (src/nn/quant/gptq/q_gemm_k_major.cu:1083-1100):
// dequant_k_major writes the WHOLE fp16 weight matrix W16 = rn16(rn16(q - z) * s) to memory (:843-952,
// 2 bytes per weight, every call) and hands it to cuBLAS (fp32 accumulate).  Here the same W16 values are
// produced in registers right in front of the MFMA that consumes them -- the 4-bit weights are read once
// per M-tile and nothing is materialised:
//   y[m,n] = half( sum_k x[m,k] * W16[n,k] (+ bias[n]) ),  products exact in fp32, fp32 accumulation.
// (The M <= 16 kernel applies the scale to fp32 group sums instead; like in the reference the two row
// ranges therefore round differently.)
//
// Tiling (ZLW4M layout, see w4_mfma.hip): workgroup = 4 waves = BM x 128 outputs, wave = BM x 32 (two
// 16-row weight tiles share every activation fragment read from LDS); K advances in 128-k chunks = one
// ZLW4M item per weight tile; the activation chunk (BM x 128 halfs) is double-buffered in LDS with padded
// rows (conflict-free ds_read_b128), global -> registers one chunk ahead; the weight items ride an 8-slot
// ring of non-temporal buffer loads (4 chunks ahead).  Per chunk and wave: 2 x 52 VALU ops of dequant feed
// 2 x 4 x (BM/16) MFMAs (v_mfma_f32_16x16x32_f16) on 2 x BM/16 independent accumulators.
#include "zl_common.h"

namespace {

constexpr int kWavesT = 4;
constexpr int kThreadsT = kWavesT * 64;
constexpr int kBN = kWavesT * 32;
constexpr int kRingT = 8;          // items (2 per chunk)
constexpr int kRowHalfs = 128 + 8; // padded LDS row (an XOR-swizzled 256-B row image measured the same: 312 us)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

struct TiledParams {
    const uint16_t* x;
    int64_t ldx;
    const uint4* qw;
    const uint32_t* meta;
    uint32_t qw_bytes, meta_bytes;
    const uint16_t* bias;
    const uint16_t* residual;
    uint16_t* y;
    int m, n, k;
    int groups;      // k / 128 chunks
    int tiles;       // ceil(n / 16)
    int epi, ld_out;
    // split-K (few rows, few column tiles): blockIdx.z handles chunks [z * split_chunks, ...) and leaves its raw
    // fp32 partial in ws[z][row][n] (row stride ld_ws); k_splitk_epilogue sums the splits in index order
    int split_chunks;
    float* ws;
    int ld_ws;
};

__device__ __forceinline__ uint32_t and_or_t(uint32_t w, uint32_t mask_s, uint32_t magic_v) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask_s), "v"(magic_v));
    return r;
}

// word -> 8 x fp16 rn16((q - z) * s): exact (q - z), one rounding in the multiply == dequant_k_major
__device__ __forceinline__ h8 dequant_scaled(uint32_t w, hv2 z1, hv2 z16, hv2 s2, uint32_t mask_lo, uint32_t mask_hi,
                                             uint32_t magic) {
    const hv2 one16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    const hv2 d0 = (__builtin_bit_cast(hv2, and_or_t(w, mask_lo, magic)) + z1) * s2;
    const hv2 d1 = __builtin_elementwise_fma(__builtin_bit_cast(hv2, and_or_t(w, mask_hi, magic)), one16, z16) * s2;
    const uint32_t wb = w >> 8;
    const hv2 d2 = (__builtin_bit_cast(hv2, and_or_t(wb, mask_lo, magic)) + z1) * s2;
    const hv2 d3 = __builtin_elementwise_fma(__builtin_bit_cast(hv2, and_or_t(wb, mask_hi, magic)), one16, z16) * s2;
    h8 a;
    a[0] = d0.x; a[1] = d0.y; a[2] = d1.x; a[3] = d1.y; a[4] = d2.x; a[5] = d2.y; a[6] = d3.x; a[7] = d3.y;
    return a;
}

__device__ __forceinline__ float silu_t(float x) { return x / (1.0f + expf(-x)); }

template <int BM>
__global__ __launch_bounds__(kThreadsT, 2) void k_w4a16_gemm_tiled(const TiledParams p) {
    constexpr int RB = BM / 16;               // 16-row blocks of the M tile
    constexpr int XR = BM / 16;               // uint4 per thread per x chunk (16 threads x 16 B per row)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    uint16_t (*xs)[BM * kRowHalfs] = reinterpret_cast<uint16_t (*)[BM * kRowHalfs]>(smem_t);   // [2][BM * kRowHalfs]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrow = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * BM;
    const int tile0 = blockIdx.x * (kBN / 16) + wave * 2;   // this wave's two 16-row weight tiles
    const int g_begin = p.ws ? blockIdx.z * p.split_chunks : 0;
    const int G = p.ws ? min(p.groups, g_begin + p.split_chunks) : p.groups;   // one past this workgroup's last chunk

    // ---- weight ring: item (j, g) of this wave = tile (tile0 + j), chunk g
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.qw), 0, p.qw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.meta), 0, p.meta_bytes, 0x00020000);
    const uint32_t q_off = (uint32_t)lane * 16u, m_off = (uint32_t)nrow * 4u;
    const int t0c = tile0 < p.tiles ? tile0 : p.tiles - 1, t1c = tile0 + 1 < p.tiles ? tile0 + 1 : p.tiles - 1;
    const uint32_t base0 = (uint32_t)t0c * (uint32_t)p.groups, base1 = (uint32_t)t1c * (uint32_t)p.groups;
    uint4 wq[kRingT];
    uint32_t mt[kRingT];
    int iss_g = g_begin;                      // next chunk to issue (both tiles)
    auto issue_pair = [&](int slot0) {
        const uint32_t g = (uint32_t)(iss_g < G ? iss_g : G - 1);
        const uint32_t it0 = base0 + g, it1 = base1 + g;
        wq[slot0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rq, q_off, it0 * 1024u, 2));
        mt[slot0] = __builtin_amdgcn_raw_buffer_load_b32(rm, m_off, it0 * 64u, 2);
        wq[slot0 + 1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rq, q_off, it1 * 1024u, 2));
        mt[slot0 + 1] = __builtin_amdgcn_raw_buffer_load_b32(rm, m_off, it1 * 64u, 2);
        ++iss_g;
    };

    // ---- activation chunk loads: thread -> (row = tid / 16 + 16 r, 16-byte column tid % 16)
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
        for (int r = 0; r < XR; ++r)
            *reinterpret_cast<uint4*>(&xs[buf][(xrow + 16 * r) * kRowHalfs + xcol]) = xr[r];
    };

    load_x(g_begin);
#pragma unroll
    for (int s = 0; s < kRingT; s += 2) {
        issue_pair(s);
        __builtin_amdgcn_sched_barrier(0);
    }
    store_x(0);
    load_x(g_begin + 1);
    __syncthreads();

    const uint32_t mask_lo = __builtin_amdgcn_readfirstlane(0x000f000fu);
    const uint32_t mask_hi = __builtin_amdgcn_readfirstlane(0x00f000f0u);
    uint32_t magic = 0x64006400u;
    asm volatile("" : "+v"(magic));
    const hv2 c960 = {(_Float16)960.f, (_Float16)960.f};

    f4 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        acc[rb][0] = (f4){0.f, 0.f, 0.f, 0.f};
        acc[rb][1] = (f4){0.f, 0.f, 0.f, 0.f};
    }

    // one K chunk: dequantise the two weight items, refill their ring slots, run the MFMAs of all row blocks
    auto chunk = [&](int slot0, int buf) {
        h8 bfr[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t mw = mt[slot0 + j];
            const hv2 z1 = __builtin_bit_cast(hv2, __builtin_amdgcn_perm(mw, mw, 0x03020302u));
            const hv2 z16 = z1 + c960;
            const hv2 s2 = __builtin_bit_cast(hv2, __builtin_amdgcn_perm(mw, mw, 0x01000100u));
            const uint32_t wds[4] = {wq[slot0 + j].x, wq[slot0 + j].y, wq[slot0 + j].z, wq[slot0 + j].w};
#pragma unroll
#ifdef ZL_TEXP_NODEQ
            for (int t = 0; t < 4; ++t) bfr[j][t] = __builtin_bit_cast(h8, make_uint4(wds[t], mw, wds[(t + 1) & 3], magic));
#else
            for (int t = 0; t < 4; ++t) bfr[j][t] = dequant_scaled(wds[t], z1, z16, s2, mask_lo, mask_hi, magic);
#endif
        }
        issue_pair(slot0);
        const uint16_t* xb = &xs[buf][nrow * kRowHalfs + kq * 8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
#ifdef ZL_TEXP_NOLDS
                const h8 a = __builtin_bit_cast(h8, make_uint4(magic + rb, magic + t, mask_lo, mask_hi));
#else
                const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(xb + rb * 16 * kRowHalfs + t * 32));
#endif
                acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bfr[0][t], acc[rb][0], 0, 0, 0);
                acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bfr[1][t], acc[rb][1], 0, 0, 0);
            }
        }
    };

    // ---- main loop: 4 chunks per turn of the ring (static slot indices)
    int g = g_begin;                          // ring slots and LDS buffers go by the LOCAL chunk index (g - g_begin)
#pragma unroll 1
    for (; g < G; g += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (g + u < G) {                  // workgroup-uniform
                chunk(2 * u, u & 1);
#ifndef ZL_TEXP_NOSTAGE
                store_x((u + 1) & 1);         // chunk g+u+1 (loaded one step ago) -> the other buffer
                load_x(g + u + 2);
#endif
#ifndef ZL_TEXP_NOBAR
                __syncthreads();
#endif
            }
        }
    }

    if (p.ws) {                               // split-K: raw partial sums, epilogue in k_splitk_epilogue
        float* wsz = p.ws + (size_t)blockIdx.z * p.m * p.ld_ws;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = (tile0 + j) * 16 + nrow;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + rb * 16 + 4 * kq + i;
                    if (row < p.m && n < p.ld_ws) wsz[(size_t)row * p.ld_ws + n] = acc[rb][j][i];
                }
            }
        }
        return;
    }

    // ---- epilogue: C fragment = column n (lane & 15), rows 4 kq + i of each 16-row block
    const bool silu = (p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = (tile0 + j) * 16 + nrow;
        const float b = ((p.epi & ZL_EPI_BIAS) && p.bias && n < p.n) ? (float)__builtin_bit_cast(_Float16, p.bias[n]) : 0.f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + rb * 16 + 4 * kq + i;
                float v = acc[rb][j][i];
                if (!silu) {
                    if (row < p.m && n < p.n) {
                        const size_t o = (size_t)row * p.ld_out + n;
                        float ov;
                        if (p.epi & ZL_EPI_ADD_C) ov = ((float)__builtin_bit_cast(_Float16, p.y[o]) + v) + b;
                        else ov = v + b;
                        _Float16 y16 = zl_f32_to_f16(ov);
                        if (p.epi & ZL_EPI_RESIDUAL)
                            y16 = zl_f32_to_f16((float)__builtin_bit_cast(_Float16, p.residual[o]) + (float)y16);
                        p.y[o] = __builtin_bit_cast(uint16_t, y16);
                    }
                } else {
                    // rows of the packed matrix interleave gate (even n) and up (odd n): partner = lane ^ 1
                    v += b;
                    const float other = __shfl_xor(v, 1, 64);
                    if ((nrow & 1) == 0 && row < p.m && n + 1 < p.n) {
                        float gt = v, up = other, ov;
                        if (p.epi & ZL_EPI_SILU_MUL) {
                            gt = (float)zl_f32_to_f16(gt);
                            up = (float)zl_f32_to_f16(up);
                            ov = silu_t(gt) * up;
                        } else {
                            ov = (float)((double)gt / (1.0 + (double)expf(-gt))) * up;
                        }
                        p.y[(size_t)row * p.ld_out + n / 2] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(ov));
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_w4a16_gemm_wide -- the prompt-chunk GEMM (M >= 128): same operands, same arithmetic, a tile shaped for the LDS pipe
// and scheduled inside the wave.
//
// PMC of k_w4a16_gemm_tiled<128> (wave = 128 x 32 outputs, two workgroups per CU) showed all three pipes loaded at once:
// every activation fragment read from LDS fed two MFMAs (128 KB of ds_read_b128 per 128-k chunk and workgroup = as many
// cycles as its 256 MFMAs), 2-way bank conflicts on top, 176 VALU next to 64 MFMAs per chunk and wave.  Here:
//   * workgroup = 4 waves = 128 x 256 outputs, wave = 128 x 64: FOUR weight tiles share every activation fragment (half
//     the LDS reads per MFMA, half the dequant VALU per MFMA of a 64-row tile); ONE workgroup per CU with the whole
//     register file (128 accumulators, a 4-chunk weight ring, two sets of B fragments, three activation fragments);
//   * everything that is not an MFMA rides in the shadow of the MFMAs of the SAME wave: the k-step is cut into 8 blocks
//     of 4 MFMAs (one 16-row block x four tiles), each block carries one activation-fragment read two blocks ahead, one
//     eighth of the NEXT k-step's dequantisation (half a word: 7 VALU), and one piece of the activation staging or of the
//     weight ring refill; sched_barrier pins the block order, the compiler schedules inside a block;
//   * the LDS image of the activation chunk (128 rows x 256 B) is XOR-swizzled: 16-byte unit u of row r sits at
//     u ^ (r & 15): the 16 rows of a ds_read_b128 lane group cover the 64 banks exactly once, and so do the 8 units of a
//     ds_write_b128 lane group;
//   * blockIdx -> (column tile, row tile) such that the row tiles of one column tile run on ONE XCD at the same time: its
//     weights come from HBM once and from that XCD's L2 for the other row tiles.
// M tile = 16 RB rows, RB = 8 (128 rows) or 16 (256 rows: the dequant VALU of a weight item is shared by twice the MFMAs --
// a SIMD runs MFMAs and VALU one after the other, not side by side: PMC of the 128-row tile, 2 048 MFMA + 1 124 VALU cycles
// per chunk and wave in a 3 850-cycle chunk)
constexpr int kWideRing = 2;                         // weight chunks in flight (4 items each): ~2 x 2 000 cycles ahead

template <int U>
struct WideIdx { static constexpr int value = U; };

// the main loop of k_w4a16_gemm_wide is written in issue order: volatile one-instruction asm statements keep their source
// order, so the dequant VALU can be placed BETWEEN the MFMAs of a block (a wave issues in order: four MFMAs back to back
// hold it for ~50 cycles and the VALU behind them then runs with the matrix pipe idle)
__device__ __forceinline__ uint32_t wv_and_or(uint32_t w, uint32_t mask_s, uint32_t magic_v) {
    uint32_t r;
    asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask_s), "v"(magic_v));
    return r;
}
__device__ __forceinline__ uint32_t wv_lshr8(uint32_t w) {
    uint32_t r;
    asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ uint32_t wv_pk_add(uint32_t a, uint32_t b) {
    uint32_t r;
    asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t wv_pk_fma(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t wv_pk_mul(uint32_t a, uint32_t b) {
    uint32_t r;
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

#ifndef ZL_WIDE_OCC
#define ZL_WIDE_OCC 1
#endif
// NT = 16-column weight tiles per wave (4: 256 columns per workgroup; 3: 192 -- N = 6144 = 32 x 192 gives the qkv projection
// of a 1024-token chunk 256 workgroups instead of 192)
// workgroup -> tile map of one launch: column strips cx_base .. cx_base + gx - 1; the first s0 of them carry gy row tiles each, the
// others gy2 (0: none); row tile ry starts at row_base + ry x (16 RB).  The plain launch is {gx, gy, gx, 0, 0, 0}; the two-height
// plan of a prompt GEMM whose tile count is not a multiple of the CU count runs TWO launches (256-row tiles, then 192-row tiles over
// the rows the first one left) that each fill the chip exactly once -- see the launcher.
struct WideMap {
    int gx, gy, s0, gy2, cx_base, row_base;
};

template <int RB, int NT>
__global__ __launch_bounds__(256, ZL_WIDE_OCC) void k_w4a16_gemm_wide(const TiledParams p, const WideMap mp) {
    constexpr int kWideBM = 16 * RB;
    constexpr int kWideChunkBytes = kWideBM * 128 * 2;         // one 128-k activation chunk in LDS
    constexpr int NBLK = 4 * RB;                               // blocks of 4 MFMAs per chunk
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_w[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrow = lane & 15, kq = lane >> 4;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    // strips xcd, xcd + 8, ... of the launch belong to this XCD: first those below s0 (gy row tiles each), then the rest (gy2 each)
    const int nx = mp.s0 > xcd ? (mp.s0 - xcd + 7) >> 3 : 0;
    int rel, ry;
    if (local < nx * mp.gy) {
        rel = (local / mp.gy) * 8 + xcd;
        ry = local % mp.gy;
    } else {
        if (mp.gy2 <= 0) return;
        const int l2 = local - nx * mp.gy;
        rel = (nx + l2 / mp.gy2) * 8 + xcd;
        ry = l2 % mp.gy2;
    }
    if (rel >= mp.gx) return;
    const int cx = mp.cx_base + rel;
    const int m0 = mp.row_base + ry * kWideBM;
    if (m0 >= p.m) return;
    const int tile0 = cx * (4 * NT) + wave * NT;               // this wave's NT 16-column weight tiles
    const int g_begin = p.ws ? blockIdx.y * p.split_chunks : 0;
    const int G = p.ws ? min(p.groups, g_begin + p.split_chunks) : p.groups;

    // ---- weight ring: kWideRing chunks x 4 items
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.qw), 0, p.qw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.meta), 0, p.meta_bytes, 0x00020000);
    const uint32_t q_off = (uint32_t)lane * 16u, m_off = (uint32_t)nrow * 4u;
    uint32_t tbase[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) tbase[j] = (uint32_t)(tile0 + j < p.tiles ? tile0 + j : p.tiles - 1) * (uint32_t)p.groups;
    uint4 wq[NT * kWideRing];
    uint32_t mt[NT * kWideRing];
    auto issue_item = [&](int slot, int j, int g) {
        const uint32_t it = tbase[j] + (uint32_t)(g < G ? g : G - 1);
        wq[slot] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rq, q_off, it * 1024u, 2));
        mt[slot] = __builtin_amdgcn_raw_buffer_load_b32(rm, m_off, it * 64u, 2);
    };

    // ---- activation staging: thread -> (row tid / 16 + 16 r, 16-byte unit tid % 16), LDS unit = unit ^ (row & 15);
    // rows through a buffer descriptor (rows past M clamp to the last row: never stored)
    const int xrow = threadIdx.x >> 4, xu = threadIdx.x & 15;
    const uint32_t xs_off = (uint32_t)(xrow * 256 + ((xu ^ xrow) * 16));
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, 0xffffffffu, 0x00020000);
    uint32_t xg_off[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = m0 + xrow + 16 * r;
        xg_off[r] = (uint32_t)(((size_t)(row < p.m ? row : p.m - 1) * p.ldx + xu * 8) * 2);
    }
    uint4 xr[RB];
    auto load_x1 = [&](int r, int g) {
        const uint32_t gc = (uint32_t)(g < G ? g : G - 1);
        xr[r] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, xg_off[r], gc * 256u, 0));
    };
    auto store_x1 = [&](int r, uint32_t buf_off) {
        *reinterpret_cast<uint4*>(smem_w + buf_off + xs_off + r * 4096) = xr[r];
    };
    // fragment reads: row 16 rb + nrow, unit (4 t + kq) ^ nrow
    const uint32_t a_off0 = (uint32_t)(nrow * 256 + ((kq ^ nrow) * 16));     // t = 0; t > 0: ^ (64 t)

    const uint32_t mask_lo = __builtin_amdgcn_readfirstlane(0x000f000fu);
    const uint32_t mask_hi = __builtin_amdgcn_readfirstlane(0x00f000f0u);
    uint32_t magic = 0x64006400u;
    asm volatile("" : "+v"(magic));
    const hv2 c960 = {(_Float16)960.f, (_Float16)960.f};
    const hv2 one16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    uint32_t one16_v = 0x2c002c00u;                            // 0.0625 x 2, in a VGPR for the asm statements
    asm volatile("" : "+v"(one16_v));

    f4 acc[RB][NT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[rb][j] = (f4){0.f, 0.f, 0.f, 0.f};

    hv2 z1[NT], z16[NT], s2[NT];
    // live == false: a chunk past the end (odd chunk counts run the second half of the unrolled pair on the clamped last
    // chunk): its scales are zero, so it adds exact zeros
    auto load_consts = [&](int slot0, bool live) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const uint32_t mw = mt[slot0 + j];
            z1[j] = __builtin_bit_cast(hv2, __builtin_amdgcn_perm(mw, mw, 0x03020302u));
            z16[j] = z1[j] + c960;
            s2[j] = __builtin_bit_cast(hv2, live ? __builtin_amdgcn_perm(mw, mw, 0x01000100u) : 0u);
        }
    };
    auto word_of = [&](int slot, int t) -> uint32_t {
        return t == 0 ? wq[slot].x : (t == 1 ? wq[slot].y : (t == 2 ? wq[slot].z : wq[slot].w));
    };
    // B fragments of two k-steps as raw words: bw[set][tile][4]; half h of a word fills [2 h], [2 h + 1] (dequant_scaled's order)
    uint32_t bw[2][NT][4];
    auto dequant_half = [&](int set, int j, uint32_t w, int h) {
        const uint32_t ws = h ? (w >> 8) : w;
        const hv2 lo = (__builtin_bit_cast(hv2, and_or_t(ws, mask_lo, magic)) + z1[j]) * s2[j];
        const hv2 hi = __builtin_elementwise_fma(__builtin_bit_cast(hv2, and_or_t(ws, mask_hi, magic)), one16, z16[j]) * s2[j];
        bw[set][j][2 * h] = __builtin_bit_cast(uint32_t, lo);
        bw[set][j][2 * h + 1] = __builtin_bit_cast(uint32_t, hi);
    };
    auto bfrag = [&](int set, int j) -> h8 {
        return __builtin_bit_cast(h8, make_uint4(bw[set][j][0], bw[set][j][1], bw[set][j][2], bw[set][j][3]));
    };

    // ---- prologue
#pragma unroll
    for (int r = 0; r < RB; ++r) load_x1(r, g_begin);
#pragma unroll
    for (int c = 0; c < kWideRing; ++c)
#pragma unroll
        for (int j = 0; j < NT; ++j) issue_item(NT * c + j, j, g_begin + c);
#pragma unroll
    for (int r = 0; r < RB; ++r) store_x1(r, 0);
#pragma unroll
    for (int r = 0; r < RB; ++r) load_x1(r, g_begin + 1);
    load_consts(0, true);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        dequant_half(0, j, word_of(j, 0), 0);
        dequant_half(0, j, word_of(j, 0), 1);
    }
    __syncthreads();

    uint4 af[4];
    // one 128-k chunk; U = chunk index mod 2: ring slots 4 U .. 4 U + 3, LDS buffer U.  Straight-line code: 32 blocks of
    // 4 MFMAs (k-step t = blk / 8, 16-row block rb = blk % 8), each with its share of everything else
    auto chunk = [&](auto Uc, int g, bool next_live) {
        constexpr int U = decltype(Uc)::value;
        constexpr int cs = NT * U, ns = NT * (U ^ 1);
        constexpr uint32_t xb = U * kWideChunkBytes, xo = (U ^ 1) * kWideChunkBytes;
        auto read_a = [&](int slot, int blk) {
            af[slot] = *reinterpret_cast<const uint4*>(smem_w + xb + (a_off0 ^ (uint32_t)((blk / RB) * 64)) + (blk % RB) * 4096);
        };
        read_a(0, 0);
        read_a(1, 1);
        read_a(2, 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const int t = blk / RB, rb = blk % RB;
            const h8 a = __builtin_bit_cast(h8, af[blk & 3]);
            // the accumulators are tied to ONE AGPR quad each (inline asm "+a"): with the builtin the allocator rotates 64
            // of the 128 accumulators through copies at the loop back-edge (192 v_accvgpr moves per two chunks, each
            // serialising against its MFMA)
#define ZL_WIDE_MFMA(J)                                                                                      \
    {                                                                                                        \
        const h8 bb = bfrag(t & 1, J);                                                                       \
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[rb][J]) : "v"(a), "v"(bb));         \
    }
            // one eighth of the next k-step's B fragments between the MFMAs: tile dj = rb / 2, half dh = rb & 1 of word
            // t + 1 of this chunk (t == 3: word 0 of the next chunk, with its meta)
            // blocks with rb % (RB / 8) == 0 carry one of the 2 NT dequant slices of the next k-step: slice sl = tile sl / 2, half sl & 1
            const int sl = rb / (RB / 8);
            const bool deq = (rb % (RB / 8)) == 0 && sl < 2 * NT;
            const int dj = deq ? sl >> 1 : 0, dh = sl & 1, dset = (t + 1) & 1;
            if (t == 3 && rb == 0) load_consts(ns, next_live);
            const uint32_t w = t < 3 ? word_of(cs + dj, t + 1) : word_of(ns + dj, 0);
            uint32_t ws = 0, x_lo = 0, x_hi = 0, y_lo = 0, y_hi = 0;
#ifdef ZL_WEXP_NODEQ
            if (deq) bw[dset][dj][2 * dh] = w;
#define ZL_WIDE_DEQ(STEP)
#else
#define ZL_WIDE_DEQ(STEP)                                                                                   \
    if (deq) {                                                                                               \
        if (STEP == 0) ws = dh ? wv_lshr8(w) : w;                                                            \
        if (STEP == 1) { x_lo = wv_and_or(ws, mask_lo, magic); x_hi = wv_and_or(ws, mask_hi, magic); }       \
        if (STEP == 2) {                                                                                     \
            y_lo = wv_pk_add(x_lo, __builtin_bit_cast(uint32_t, z1[dj]));                                    \
            y_hi = wv_pk_fma(x_hi, one16_v, __builtin_bit_cast(uint32_t, z16[dj]));                          \
        }                                                                                                    \
        if (STEP == 3) {                                                                                     \
            bw[dset][dj][2 * dh] = wv_pk_mul(y_lo, __builtin_bit_cast(uint32_t, s2[dj]));                    \
            bw[dset][dj][2 * dh + 1] = wv_pk_mul(y_hi, __builtin_bit_cast(uint32_t, s2[dj]));                \
        }                                                                                                    \
    }
#endif
            ZL_WIDE_DEQ(0)
            ZL_WIDE_MFMA(0)
#ifndef ZL_WEXP_NOLDS
            if (blk + 3 < NBLK) read_a((blk + 3) & 3, blk + 3);
#endif
            ZL_WIDE_DEQ(1)
            ZL_WIDE_MFMA(1)
            ZL_WIDE_DEQ(2)
#ifndef ZL_WEXP_NOSTAGE
            // staging: piece r of chunk g + 1 (loaded one chunk ago) -> the other LDS buffer, its registers take chunk g + 2
            if (t < 2 && (blk & 1) == 0) {
                store_x1(blk >> 1, xo);
                load_x1(blk >> 1, g + 2);
            }
#endif
            if constexpr (NT > 2) ZL_WIDE_MFMA(2)
            ZL_WIDE_DEQ(3)
            if constexpr (NT > 3) ZL_WIDE_MFMA(3)
#undef ZL_WIDE_DEQ
#undef ZL_WIDE_MFMA
            if (t == 3 && rb >= 1 && rb <= NT) issue_item(cs + rb - 1, rb - 1, g + kWideRing);
#ifndef ZL_WEXP_NOPIN
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#ifndef ZL_WEXP_NOBAR
        __syncthreads();
#endif
    };

#pragma unroll 1
    for (int g = g_begin; g < G; g += 2) {
        chunk(WideIdx<0>{}, g, g + 1 < G);
        chunk(WideIdx<1>{}, g + 1, g + 2 < G);
    }

    // the MFMAs are inline asm: the compiler does not know that the accumulators come out of the matrix pipe and inserts no
    // wait states between the last MFMAs and the first reads of their results (observed: registers of the last block read
    // one MFMA early).  MFMAs retire in order, so the results of the last two blocks are passed through builtin MFMAs that
    // add exact zeros: the compiler then orders every read of those behind the pipe, and everything older is done by then.
    {
        const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int rb = RB - 2; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zero8, zero8, acc[rb][j], 0, 0, 0);
    }
    // ---- epilogue (row-block-major like the main loop: the accumulators stay where the loop left them)
    if (p.ws) {
        float* wsz = p.ws + (size_t)blockIdx.y * p.m * p.ld_ws;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (tile0 + j) * 16 + nrow;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + rb * 16 + 4 * kq + i;
                    if (row < p.m && n < p.ld_ws) wsz[(size_t)row * p.ld_ws + n] = acc[rb][j][i];
                }
            }
        return;
    }
    float bj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (tile0 + j) * 16 + nrow;
        bj[j] = ((p.epi & ZL_EPI_BIAS) && p.bias && n < p.n) ? (float)__builtin_bit_cast(_Float16, p.bias[n]) : 0.f;
    }
    // three separately unrolled passes (one runs): a single loop with all the branches is too large to unroll, and a
    // rolled loop would index the accumulators dynamically (= spill them)
    auto for_each = [&](auto fn) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) fn(m0 + rb * 16 + 4 * kq + i, (tile0 + j) * 16 + nrow, acc[rb][j][i] , bj[j]);
    };
    if (!(p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32))) {
        for_each([&](int row, int n, float v, float b) {
            if (row < p.m && n < p.n) {
                const size_t o = (size_t)row * p.ld_out + n;
                float ov;
                if (p.epi & ZL_EPI_ADD_C) ov = ((float)__builtin_bit_cast(_Float16, p.y[o]) + v) + b;
                else ov = v + b;
                _Float16 y16 = zl_f32_to_f16(ov);
                if (p.epi & ZL_EPI_RESIDUAL) y16 = zl_f32_to_f16((float)__builtin_bit_cast(_Float16, p.residual[o]) + (float)y16);
                p.y[o] = __builtin_bit_cast(uint16_t, y16);
            }
        });
    } else if (p.epi & ZL_EPI_SILU_MUL) {
        // rows of the packed matrix interleave gate (even n) and up (odd n): partner = lane ^ 1.  Both lanes of a pair work:
        // the even lane finishes rows i = 0, 1 of each quad, the odd lane rows i = 2, 3 (each sends the partner the two values
        // it needs: one DPP swap per register instead of 128 silu evaluations on half of the lanes)
        const bool odd = (nrow & 1) != 0;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (tile0 + j) * 16 + nrow;
                const float b = bj[j];
                float v[4], o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[rb][j][i] + b;
                    o[i] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[i]), 0xb1, 0xf, 0xf, true));
                }
                // even lane: (gate, up) = (v, o) rows 0, 1; odd lane: (gate, up) = (o, v) rows 2, 3; column n / 2 either way
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float gt32 = odd ? o[2 + h] : v[h], up32 = odd ? v[2 + h] : o[h];
                    const int row = m0 + rb * 16 + 4 * kq + (odd ? 2 + h : h);
                    const float gt = (float)zl_f32_to_f16(gt32), up = (float)zl_f32_to_f16(up32);
                    if (row < p.m && (n | 1) < p.n)
                        p.y[(size_t)row * p.ld_out + n / 2] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(silu_t(gt) * up));
                }
            }
    } else {
        for_each([&](int row, int n, float v, float b) {
            v += b;
            const float other = __shfl_xor(v, 1, 64);
            if ((nrow & 1) == 0 && row < p.m && n + 1 < p.n) {
                const float ov = (float)((double)v / (1.0 + (double)expf(-v))) * other;
                p.y[(size_t)row * p.ld_out + n / 2] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(ov));
            }
        });
    }
}

// sums the split-K partials in split order (deterministic) and applies the launcher's epilogue
__global__ __launch_bounds__(256) void k_splitk_epilogue(const float* __restrict__ ws, int splits, int m, int n, int ld_ws,
                                                         const uint16_t* __restrict__ bias, const uint16_t* residual,
                                                         uint16_t* y, int epi, int ld_out) {
    const bool silu = (epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
    const int cols = silu ? n / 2 : n;
    const int64_t total = (int64_t)m * cols;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / cols), c = (int)(idx % cols);
        auto sum = [&](int nn) {
            float v = 0.f;
            for (int z = 0; z < splits; ++z) v += ws[((size_t)z * m + row) * ld_ws + nn];
            return v;
        };
        if (!silu) {
            const float v = sum(c);
            const float b = ((epi & ZL_EPI_BIAS) && bias) ? (float)__builtin_bit_cast(_Float16, bias[c]) : 0.f;
            const size_t o = (size_t)row * ld_out + c;
            float ov;
            if (epi & ZL_EPI_ADD_C) ov = ((float)__builtin_bit_cast(_Float16, y[o]) + v) + b;
            else ov = v + b;
            _Float16 y16 = zl_f32_to_f16(ov);
            if (epi & ZL_EPI_RESIDUAL) y16 = zl_f32_to_f16((float)__builtin_bit_cast(_Float16, residual[o]) + (float)y16);
            y[o] = __builtin_bit_cast(uint16_t, y16);
        } else {
            float gt = sum(2 * c), up = sum(2 * c + 1);
            if ((epi & ZL_EPI_BIAS) && bias) {
                gt += (float)__builtin_bit_cast(_Float16, bias[2 * c]);
                up += (float)__builtin_bit_cast(_Float16, bias[2 * c + 1]);
            }
            float ov;
            if (epi & ZL_EPI_SILU_MUL) {
                gt = (float)zl_f32_to_f16(gt);
                up = (float)zl_f32_to_f16(up);
                ov = silu_t(gt) * up;
            } else {
                ov = (float)((double)gt / (1.0 + (double)expf(-gt))) * up;
            }
            y[(size_t)row * ld_out + c] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(ov));
        }
    }
}

// the same sums for the plain epilogues, four columns per thread (16-byte partial loads, 8-byte stores): the split
// projections of a prompt chunk (attn_out, w_out at M = 1024) spent 17 us per launch in the scalar version
__global__ __launch_bounds__(256) void k_splitk_epilogue_v4(const float* __restrict__ ws, int splits, int m, int n, int ld_ws,
                                                            const uint16_t* __restrict__ bias, const uint16_t* residual,
                                                            uint16_t* y, int epi, int ld_out) {
    const int cols4 = n / 4;
    const int64_t total = (int64_t)m * cols4;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / cols4), c = (int)(idx % cols4) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < splits; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(ws + ((size_t)z * m + row) * ld_ws + c);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        const size_t o = (size_t)row * ld_out + c;
        uint16_t bb[4] = {0, 0, 0, 0}, yy[4] = {0, 0, 0, 0}, rr[4] = {0, 0, 0, 0};
        if ((epi & ZL_EPI_BIAS) && bias) *reinterpret_cast<uint2*>(bb) = *reinterpret_cast<const uint2*>(bias + c);
        if (epi & ZL_EPI_ADD_C) *reinterpret_cast<uint2*>(yy) = *reinterpret_cast<const uint2*>(y + o);
        if (epi & ZL_EPI_RESIDUAL) *reinterpret_cast<uint2*>(rr) = *reinterpret_cast<const uint2*>(residual + o);
        uint16_t out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float b = ((epi & ZL_EPI_BIAS) && bias) ? (float)__builtin_bit_cast(_Float16, bb[i]) : 0.f;
            float ov;
            if (epi & ZL_EPI_ADD_C) ov = ((float)__builtin_bit_cast(_Float16, yy[i]) + v[i]) + b;
            else ov = v[i] + b;
            _Float16 y16 = zl_f32_to_f16(ov);
            if (epi & ZL_EPI_RESIDUAL) y16 = zl_f32_to_f16((float)__builtin_bit_cast(_Float16, rr[i]) + (float)y16);
            out[i] = __builtin_bit_cast(uint16_t, y16);
        }
        *reinterpret_cast<uint2*>(y + o) = *reinterpret_cast<uint2*>(out);
    }
}

// picks the vector version where the shapes and pointers allow
static int launch_splitk_epilogue(const float* ws, int splits, int64_t m, int64_t n, int ld_ws, const uint16_t* bias,
                                  const uint16_t* residual, uint16_t* y, int epi, int ld_out, hipStream_t hs) {
    const bool silu = (epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
    const bool v4 = !silu && n % 4 == 0 && ld_ws % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)y & 7) == 0 &&
                    (!bias || ((uintptr_t)bias & 7) == 0) && (!residual || ((uintptr_t)residual & 7) == 0);
    const int64_t outs = v4 ? m * (n / 4) : m * (silu ? n / 2 : n);
    const unsigned rgrid = (unsigned)((outs + 255) / 256 > 8192 ? 8192 : (outs + 255) / 256);
    if (v4)
        hipLaunchKernelGGL(k_splitk_epilogue_v4, dim3(rgrid), dim3(256), 0, hs, ws, splits, (int)m, (int)n, ld_ws, bias, residual, y, epi, ld_out);
    else
        hipLaunchKernelGGL(k_splitk_epilogue, dim3(rgrid), dim3(256), 0, hs, ws, splits, (int)m, (int)n, ld_ws, bias, residual, y, epi, ld_out);
    return zl_launch_status();
}

}  // namespace

extern "C" int zl_w4a16_gemm_tiled(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta,
                                   const uint16_t* bias, const uint16_t* residual, uint16_t* y, int64_t m, int64_t n,
                                   int64_t k, int64_t group_size, int epilogue, zl_stream_t s) {
    return zl_w4a16_gemm_tiled_ex(x, ldx, qw, meta, bias, residual, y, m, n, k, group_size, epilogue, nullptr, s);
}

// called by zl_w4a16_gemm_mfma_ex for m > 16 (same operands)
extern "C" int zl_w4a16_gemm_tiled_ex(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta,
                                      const uint16_t* bias, const uint16_t* residual, uint16_t* y, int64_t m, int64_t n,
                                      int64_t k, int64_t group_size, int epilogue, const zl_w4_opts_t* opts, zl_stream_t s) {
    static const zl_w4_opts_t kNoOpts = {};
    const zl_w4_opts_t& o = opts ? *opts : kNoOpts;
    ZL_CHECK_ARG(x && qw && meta && y && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(ldx >= k && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0 && k % 128 == 0, ZL_ESHAPE);
    // the kernels address the activations with 32-bit byte offsets (buffer descriptors): refuse what would wrap
    ZL_CHECK_ARG(((m - 1) * ldx + k) * 2 < ((int64_t)1 << 32), ZL_ELIMIT);
    ZL_CHECK_ARG(!(epilogue & ZL_EPI_RESIDUAL) || residual, ZL_EINVAL);
    zl_w4_layout_t L;
    int st = zl_w4m_layout(n, k, group_size, &L);
    if (st) return st;
    const bool silu = epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32);
    ZL_CHECK_ARG(!silu || n % 2 == 0, ZL_ESHAPE);
    if (L.qw_bytes >= (int64_t)1 << 32) return ZL_ELIMIT;
    TiledParams p;
    p.x = x; p.ldx = ldx;
    p.qw = reinterpret_cast<const uint4*>(qw);
    p.meta = meta;
    p.qw_bytes = (uint32_t)L.qw_bytes; p.meta_bytes = (uint32_t)L.scales_bytes;
    p.bias = bias; p.residual = residual; p.y = y;
    p.m = (int)m; p.n = (int)n; p.k = (int)k;
    p.groups = (int)(k / 128);
    p.tiles = (int)(L.np / 16);
    p.epi = epilogue;
    p.ld_out = (int)(silu ? n / 2 : n);
    const int gx = (int)((L.np + kBN - 1) / kBN);
    hipStream_t hs = (hipStream_t)s;
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    // prompt chunks: k_w4a16_gemm_wide<RB, NT> -- 128 / 256 rows x 256 / 192 columns per workgroup
    if (o.tiled_wide >= 0 && o.tiled_bm == 0 && (m >= 128 || (o.tiled_wide > 0 && m > 32)) && L.np >= 192) {
        // tile shape and K splits by a small cost model.  Cycles per 128-k chunk and workgroup ~ 64 RB NT (MFMA) + 281 NT
        // (dequant and the rest of the VALU) + the per-chunk remainder (measured: 3 850 for 128 x 256, ~7 000 for 256 x 256);
        // a launch takes ceil(tiles x splits / CUs) rounds of chunks / splits chunks; fewer tiles than CUs split K just far
        // enough to give every CU one workgroup -- fp32 partials through the caller's scratch, summed in split order.
        // 192-column tiles exist for N that 256 leaves short of the chip: N = 6144 (qkv) = 24 x 256 = 32 x 192; 128-column tiles for
        // N = 4096 with a short K (attn_out: 32 x 8 = 256 workgroups without splitting K).
        int best_rb = 8, best_nt = 4, best_splits = 1;
        double best_cost = -1;
        for (int nt = 4; nt >= 2; --nt)
            for (int rb = 8; rb <= 16; rb += 8) {
                if (rb == 16 && (m <= 128 || o.tiled_wide == 2)) continue;    // (tiled_wide == 2: 128-row tiles only)
                if (nt == 4 && L.np < 256) continue;
                if (nt == 2 && rb == 16) continue;                            // (128-column tiles: 128 rows only)
                const int64_t tiles = ((L.np + 64 * nt - 1) / (64 * nt)) * ((m + 16 * rb - 1) / (16 * rb));
                int sp = 1;
                if (tiles * 4 <= (int64_t)cus * 3) {
                    sp = (int)(cus / tiles);
                    const int max_s = p.groups / 8 > 0 ? p.groups / 8 : 1;
                    sp = sp > max_s ? max_s : sp;
                    sp = sp > 8 ? 8 : (sp < 1 ? 1 : sp);
                }
                if (o.tiled_splitk > 0) sp = o.tiled_splitk <= p.groups ? o.tiled_splitk : p.groups;
                const int64_t need = ZL_SCRATCH_HEADER + (int64_t)sp * m * L.np * (int64_t)sizeof(float);
                if (sp > 1 && !(o.scratch && o.scratch_bytes >= need)) sp = 1;
                const int64_t rounds = (tiles * sp + cus - 1) / cus;
                // 128-column tiles (nt = 2) feed two MFMAs per activation fragment instead of four: 2 870 cycles per chunk measured
                // (attn_out at 1 024 rows: 45.9 us unsplit against 52.1 us for 256-column tiles with two K splits + their epilogue;
                // every other Llama shape keeps the wider tile)
                const double chunk = 64.0 * rb * nt + 281.0 * nt + (nt == 2 ? 1290.0 : (rb == 16 ? 1780.0 : 700.0));
                const double cost = (double)rounds * (double)((p.groups + sp - 1) / sp) * chunk +
                                    (sp > 1 ? 0.004 * (double)sp * (double)m * (double)L.np : 0.0);
                if (best_cost < 0 || cost < best_cost * 0.97) {             // (3 %: prefer the earlier, wider candidate on a tie)
                    best_cost = cost;
                    best_rb = rb;
                    best_nt = nt;
                    best_splits = sp;
                }
            }
        if (o.tiled_wide == 3 && m > 128) best_rb = 16;                    // (experiments: force the 256-row tile)
        if (o.tiled_wide == 4) best_nt = 3;                                 // (experiments: force 192-column tiles)
        if (o.tiled_wide == 5) { best_nt = 2; best_rb = 8; best_splits = 1; }  // (experiments: 128 x 128 tiles, no K split)
        const int rbw = best_rb, ntw = best_nt;
        const int gxw = (int)((L.np + 64 * ntw - 1) / (64 * ntw)), gyw = (int)((m + 16 * rbw - 1) / (16 * rbw));
        int splits = best_splits;
        p.ws = nullptr; p.split_chunks = p.groups; p.ld_ws = (int)L.np;
        if (splits > 1) {
            p.split_chunks = (p.groups + splits - 1) / splits;
            splits = (p.groups + p.split_chunks - 1) / p.split_chunks;
            const int64_t need = ZL_SCRATCH_HEADER + (int64_t)splits * m * L.np * (int64_t)sizeof(float);
            if (o.scratch && o.scratch_bytes >= need) {
                p.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(o.scratch) + ZL_SCRATCH_HEADER);
            } else {
                splits = 1;
                p.split_chunks = p.groups;
            }
        }
        const int64_t wgs = (int64_t)((gxw + 7) / 8) * 8 * gyw;
        ZL_CHECK_ARG(wgs <= 0x7fffffff, ZL_ELIMIT);
#define ZL_WIDE_LAUNCH_MAP(RBV, NTV, MAP, WGS, SPLITS)                                                                       \
    {                                                                                                                        \
        const size_t lds_ = (size_t)2 * 16 * RBV * 256;                                                                      \
        if (lds_ > 64 * 1024) {                                                                                              \
            /* every launch: the attribute is per device, and one process may drive several (the reference's engine) */      \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_w4a16_gemm_wide<RBV, NTV>),                             \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_) != hipSuccess)                    \
                return ZL_ELIMIT;                                                                                            \
        }                                                                                                                    \
        hipLaunchKernelGGL((k_w4a16_gemm_wide<RBV, NTV>), dim3((unsigned)(WGS), (unsigned)(SPLITS)), dim3(256), lds_, hs, p, MAP); \
    }
        // Two tile heights when the 256-row tiling leaves the last round of workgroups partly empty (gate|up of a 1 024-token chunk:
        // 112 strips x 4 = 448 tiles on 256 CUs = two rounds for 1.75 rounds of work): `ystrips` of the column strips carry kbig
        // 256-row tiles and jsmall 192-row tiles (256 kbig + 192 jsmall = M) instead of M / 256 tall ones, chosen so that the tall
        // tiles of ALL strips and the 192-row tiles each come to whole rounds; two launches, one height each.  Every output is
        // still one workgroup's sum over K in chunk order: bit-identical to the plain tiling.
        if (splits == 1 && rbw == 16 && ntw == 4 && m % 256 == 0 && o.tiled_wide != 3 && o.tiled_wide != 6) {
            const int64_t c16 = 64 * 16 * 4 + 281 * 4 + 1780, c12 = 64 * 12 * 4 + 281 * 4 + 1240;
            const int64_t plain = ((int64_t)gxw * gyw + cus - 1) / cus * c16;
            int64_t best = plain;
            int by = 0, bk = 0, bj = 0;
            for (int kbig = gyw - 1; kbig >= 0; --kbig) {
                const int64_t rem = m - 256 * (int64_t)kbig;
                if (rem % 192) continue;
                const int js = (int)(rem / 192);
                for (int ys = 8; ys <= gxw; ys += 8) {
                    const int64_t big = (int64_t)gyw * (gxw - ys) + (int64_t)kbig * ys, small = (int64_t)js * ys;
                    const int64_t cost = (big + cus - 1) / cus * c16 + (small + cus - 1) / cus * c12;
                    if (cost < best) { best = cost; by = ys; bk = kbig; bj = js; }
                }
            }
            if (by > 0 && best * 100 <= plain * 95) {
                const int s0 = gxw - by;
                const int64_t per_xcd_a = (int64_t)((s0 + 7) / 8) * gyw + (int64_t)((by + 7) / 8) * bk;
                const WideMap ma = {gxw, gyw, s0, bk, 0, 0};
                const WideMap mb = {by, bj, by, 0, s0, 256 * bk};
                ZL_WIDE_LAUNCH_MAP(16, 4, ma, per_xcd_a * 8, 1)
                st = zl_launch_status();
                if (st) return st;
                ZL_WIDE_LAUNCH_MAP(12, 4, mb, (int64_t)((by + 7) / 8) * 8 * bj, 1)
                return zl_launch_status();
            }
        }
        const WideMap mw = {gxw, gyw, gxw, 0, 0, 0};
#define ZL_WIDE_LAUNCH(RBV, NTV) ZL_WIDE_LAUNCH_MAP(RBV, NTV, mw, wgs, splits)
        if (rbw == 16 && ntw == 4) ZL_WIDE_LAUNCH(16, 4)
        else if (rbw == 16 && ntw == 3) ZL_WIDE_LAUNCH(16, 3)
        else if (rbw == 16) ZL_WIDE_LAUNCH(16, 2)
        else if (ntw == 4) ZL_WIDE_LAUNCH(8, 4)
        else if (ntw == 3) ZL_WIDE_LAUNCH(8, 3)
        else ZL_WIDE_LAUNCH(8, 2)
#undef ZL_WIDE_LAUNCH
#undef ZL_WIDE_LAUNCH_MAP
        st = zl_launch_status();
        if (st || splits <= 1) return st;
        return launch_splitk_epilogue(p.ws, splits, m, n, p.ld_ws, bias, residual, y, epilogue, p.ld_out, hs);
    }
    // M-tile height: taller tiles amortise the dequant over more MFMAs (the VALU and the MFMA pipe do not
    // overlap here: 143 VALU + 32 MFMA per chunk and wave at BM = 64 measured 40 % MFMA-busy); 128 rows need
    // enough M to still fill the chip
    const int bm_env = o.tiled_bm;
    int bm = m <= 32 ? 32 : (m * (int64_t)gx >= 128 * 512 ? 128 : 64);
    if (bm_env == 32 || bm_env == 64 || bm_env == 128) bm = bm_env;
    ZL_CHECK_ARG((m + bm - 1) / bm <= 65535, ZL_ELIMIT);
    const int gy = (int)((m + bm - 1) / bm);
    // too few workgroups for the chip (decode batches: one M tile, N / 128 column tiles): split K over
    // blockIdx.z so that ~2 workgroups per CU exist, >= 4 chunks each; partials go through the device scratch
    const int split_env = o.tiled_splitk;
    int splits = 1;
    if ((int64_t)gx * gy < cus) {
        splits = (int)((2 * (int64_t)cus + (int64_t)gx * gy - 1) / ((int64_t)gx * gy));
        const int max_s = p.groups / 4 > 0 ? p.groups / 4 : 1;
        if (splits > max_s) splits = max_s;
        if (splits > 32) splits = 32;
    }
    if (split_env > 0) splits = split_env <= p.groups ? split_env : p.groups;
    p.ws = nullptr; p.split_chunks = p.groups; p.ld_ws = (int)L.np;
    if (splits > 1) {
        p.split_chunks = (p.groups + splits - 1) / splits;
        splits = (p.groups + p.split_chunks - 1) / p.split_chunks;       // no empty split
        // partials in the caller's scratch (behind the counter header); without enough of it: one split (same result
        // up to the summation order of the fp32 partials, fewer workgroups)
        const int64_t need = ZL_SCRATCH_HEADER + (int64_t)splits * m * L.np * (int64_t)sizeof(float);
        if (o.scratch && o.scratch_bytes >= need) {
            p.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(o.scratch) + ZL_SCRATCH_HEADER);
        } else {
            splits = 1;
            p.split_chunks = p.groups;
        }
    }
    const dim3 grid(gx, (unsigned)gy, (unsigned)splits);
    const size_t lds = (size_t)2 * bm * kRowHalfs * 2;
#define ZL_TILED_LAUNCH(BMV)                                                                                   \
    {                                                                                                          \
        if (lds > 64 * 1024) {                                                                                 \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_w4a16_gemm_tiled<BMV>),        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
            if (e != hipSuccess) return ZL_ELIMIT;                                                             \
        }                                                                                                      \
        hipLaunchKernelGGL(k_w4a16_gemm_tiled<BMV>, grid, dim3(kThreadsT), lds, hs, p);                         \
    }
    if (bm == 32) ZL_TILED_LAUNCH(32)
    else if (bm == 64) ZL_TILED_LAUNCH(64)
    else ZL_TILED_LAUNCH(128)
#undef ZL_TILED_LAUNCH
    st = zl_launch_status();
    if (st || splits <= 1) return st;
    return launch_splitk_epilogue(p.ws, splits, m, n, p.ld_ws, bias, residual, y, epilogue, p.ld_out, hs);
}

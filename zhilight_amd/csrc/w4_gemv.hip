// w4_gemv.hip -- W4A16 decode GEMM (M <= 4 rows per pass) on the ZLW4 layout.  SURVEY 8a rows a2, a5.
//
// What the reference does (src/nn/quant/gptq/q_gemm_k_major.cu:127-237): one 32-lane warp per
// output channel n; lane l walks the packed words l, l+32, ...; per word: exact fp16 (q - z),
// two fp16 hfma2 accumulators over the 8 weights, f32(lo) + f32(hi), fp32 fma with the group
// scale; lanes are summed by a shuffle-down tree.
//
// How it is laid out here for CDNA4:
//  * a 64-lane wavefront owns a PAIR of output rows; half-wave h = lane/32 replays the reference
//    warp for row 2*pair + h, so the per-lane fp32 chain and the 32-lane tree -- and therefore the
//    result bits -- are the reference's.
//  * ZLW4 stores the words so that ONE global_load_dwordx4 per lane (1 KiB per wave, fully
//    coalesced, non-temporal) brings 4 consecutive chain steps (words r+32*(4q+j), j=0..3) of
//    both rows; the 4 group scales / zero nibbles those steps need are one 8-byte and one 2-byte
//    load from the re-ordered meta arrays.  Nothing but the algorithmic bytes is read.
//  * loads go straight to VGPRs through a D-deep software ring (no LDS round trip for weights:
//    they are used once); the activation rows live in LDS and are read as ds_read_b128.
//  * optional fused RMSNorm prologue (every workgroup re-derives the 1/rms of the L2-resident
//    input row while its first weight loads are in flight) and fused epilogues (bias, ADD_C,
//    residual add, silu*mul on interleaved gate/up rows).
//  * no cross-wave reduction, no atomics, no split-K: a wave finishes its rows alone.
#include <stdlib.h>
#include "zl_common.h"
#include "zl_stage.h"
#include "zl_w4_exact.h"

// Workgroup = 8 wavefronts (512 threads): the activation row(s) are staged ONCE per workgroup, and one
// or two workgroups fill a CU, so the L1/TA traffic and the barrier latency of the x staging are paid
// once or twice per CU instead of eight times (in-kernel probe: 2.3 us of the 4.4 us o_proj wave
// lifetime was x staging with 256-thread workgroups, 8 per CU).
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;

// ---- optional phase-timestamp probe (build with -DZL_W4_PROBE; tools/ubench/probe_gemv.py) --------
#ifdef ZL_W4_PROBE
__device__ unsigned long long* zl_probe_buf = nullptr;  // [waves][8] wall-clock ticks (100 MHz)
extern "C" int zl_debug_set_probe(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(zl_probe_buf), &p, sizeof(p));
}
#define ZL_PROBE(slot)                                                                                 \
    do {                                                                                                \
        if (zl_probe_buf && lane == 0) zl_probe_buf[(size_t)(blockIdx.x * kWaves + wave) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define ZL_PROBE(slot) do {} while (0)
#endif

namespace {


struct W4Params {
    const uint16_t* x;
    int64_t ldx;
    const uint4* qw;
    const uint2* scales;    // [pair][q][h][c] x 4 halfs
    const uint16_t* zeros;  // [pair][q][h][c] 4 nibbles
    const uint16_t* bias;
    const uint16_t* residual;
    uint16_t* y;
    const uint16_t* norm_w;
    float norm_eps;
    int m, n, k, kp;
    int q_loads, c_classes, c_shift;
    int sym, epi;
    int pairs_total, pairs_per_wave;
    int ld_out;  // row stride of y / residual (n, or n/2 for silu*mul)
};

using namespace zlx;

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + expf(-x)); }

// kRing = weight loads in flight per wave (1 KiB each).
// XL    = activation loads (16 B) per thread per row held in registers across the weight prologue:
//         the x loads are issued FIRST (oldest in the in-order VMEM queue), then the ring of weight
//         loads, so x (an L2 hit) can be normalised and written to LDS while the HBM loads fly.
//         XL = 0 selects the generic "stage x, then start streaming" order for very large K.
template <int MT, int kRing, int XL>
__global__ __launch_bounds__(kThreads, 4) void k_w4a16_gemm(const W4Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem);                 // [MT][kp]
    float* red = reinterpret_cast<float*>(smem + (size_t)MT * p.kp * 2);  // 16 floats
    float* res_all = red + 16;                                            // [4 waves][64 rows][MT]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int cls = r >> p.c_shift;
    const int m0 = blockIdx.y * MT;
    const int Q = p.q_loads;

    float* res = res_all + wave * 64 * MT;  // [kWaves][64 rows][MT]
    ZL_PROBE(0);
    const int gw = blockIdx.x * kWaves + wave;
    const int pair0 = gw * p.pairs_per_wave;
    int npairs = p.pairs_total - pair0;
    npairs = npairs < 0 ? 0 : (npairs > p.pairs_per_wave ? p.pairs_per_wave : npairs);
    const int total = npairs * Q;

    // ---- per-lane streams: weights advance 64 uint4 per item, meta advance C entries per item.
    // A wave without work (tail of the grid) still runs the branch-free load sequence on the last
    // pair and discards it: no VMEM instruction may sit behind a branch (see `issue` below).
    const int pair0c = pair0 < p.pairs_total ? pair0 : p.pairs_total - 1;
    const uint4* wptr = p.qw + ((size_t)pair0c * Q) * 64 + lane;
    const uint2* sptr = p.scales + ((size_t)pair0c * Q * 2 + h) * p.c_classes + cls;
    const uint16_t* zptr = p.zeros + ((size_t)pair0c * Q * 2 + h) * p.c_classes + cls;
    const int meta_step = 2 * p.c_classes;

    // ---- (1) activation loads into registers (no branches: clamped addresses + selects)
    uint4 xr[MT][XL > 0 ? XL : 1];
    uint4 nwr[XL > 0 ? XL : 1];
    if constexpr (XL > 0) {
        const uint16_t* nwp = p.norm_w ? p.norm_w : p.x;  // dummy but valid address when unused
#pragma unroll
        for (int l = 0; l < XL; ++l) {
            const int i = (threadIdx.x + l * kThreads) * 8;
            nwr[l] = *reinterpret_cast<const uint4*>(nwp + (i < p.k ? i : p.k - 8));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const uint16_t* xrow = p.x + (size_t)((m0 + m) < p.m ? (m0 + m) : 0) * p.ldx;
#pragma unroll
            for (int l = 0; l < XL; ++l) {
                const int i = (threadIdx.x + l * kThreads) * 8;
                xr[m][l] = *reinterpret_cast<const uint4*>(xrow + (i < p.k ? i : p.k - 8));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    } else {
        zl_stage_rows<ZL_F16, MT, kThreads>(p.x, p.ldx, m0, p.m, p.k, p.kp, p.norm_w, p.norm_eps, xs, red);
    }

    // ---- (2) weight ring prologue
    uint4 wq[kRing];
    uint2 sc[kRing];
    uint16_t zq[kRing];  // kept 16-bit until consumed: a widening at issue time would be sunk to the
                         // loop tail by hipcc and drain the ring there
    // Item i of this wave = (pair pair0 + i / Q, load i % Q).  The issue stream is CLAMPED to the
    // wave's last item (a few duplicate, cache-hitting loads at the very end) so that no branch ever
    // surrounds a VMEM instruction: hipcc then keeps counted s_waitcnt vmcnt(N) in the steady-state
    // loop instead of draining the ring with vmcnt(0) at every loop header.
    int iss_idx = 0;
    auto issue = [&](int slot) {
        wq[slot] = zl_load_nt(wptr);
        sc[slot] = zl_load_nt(sptr);
        zq[slot] = zl_load_nt(zptr);
        const bool adv = iss_idx + 1 < total;  // wave-uniform
        ++iss_idx;
        wptr += adv ? 64 : 0;
        sptr += adv ? meta_step : 0;
        zptr += adv ? meta_step : 0;
    };

    // pin the prologue's issue order slot by slot: the loop-header s_waitcnt is the minimum over the
    // preheader and back-edge paths, so a regrouped prologue would force vmcnt(~0) there
#pragma unroll
    for (int s = 0; s < kRing; ++s) {
        issue(s);
        __builtin_amdgcn_sched_barrier(0);
    }

    ZL_PROBE(1);
    // ---- (3) activation rows -> LDS (optionally RMS-normalised: LayerNorm::forward semantics,
    // src/nn/layernorm/layernorm.cu:10-42), K padding zeroed
    if constexpr (XL > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool live = (m0 + m) < p.m;
            float rs = 1.f;
            if (p.norm_w) {
                float ss = 0.f;
#pragma unroll
                for (int l = 0; l < XL; ++l) {
                    const int i = (threadIdx.x + l * kThreads) * 8;
                    if (i < p.k) {
                        const uint32_t u[4] = {xr[m][l].x, xr[m][l].y, xr[m][l].z, xr[m][l].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const hv2 hh = as_hv2(u[e]);
                            ss = __builtin_fmaf((float)hh.x, (float)hh.x, ss);
                            ss = __builtin_fmaf((float)hh.y, (float)hh.y, ss);
                        }
                    }
                }
                ss = zl_block_sum(ss, red);
                rs = zl_rsqrt_rn(ss / (float)p.k + p.norm_eps);
            }
#pragma unroll
            for (int l = 0; l < XL; ++l) {
                const int i = (threadIdx.x + l * kThreads) * 8;
                if (i < p.kp) {
                    uint4 v = xr[m][l];
                    if (!live || i >= p.k) v = make_uint4(0, 0, 0, 0);
                    if (p.norm_w && i < p.k) {
                        const uint4 wv = nwr[l];
                        uint32_t u[4] = {v.x, v.y, v.z, v.w};
                        const uint32_t wu[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const hv2 hh = as_hv2(u[e]), ww = as_hv2(wu[e]);
                            hv2 o;
                            o.x = zl_f32_to_f16((float)hh.x * rs * (float)ww.x);
                            o.y = zl_f32_to_f16((float)hh.y * rs * (float)ww.y);
                            u[e] = __builtin_bit_cast(uint32_t, o);
                        }
                        v = make_uint4(u[0], u[1], u[2], u[3]);
                    }
                    *reinterpret_cast<uint4*>(xs + (size_t)m * p.kp + i) = v;
                }
            }
        }
    }
    __syncthreads();
    ZL_PROBE(2);

    // ---- main loop
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    int cq = 0, cpair = pair0;  // position of the item being consumed
    // loop-invariant operands of the dequant (SGPR masks, VGPR magic 0x6400 = half 1024)
    const uint32_t mask_lo = __builtin_amdgcn_readfirstlane(0x000f000fu);
    const uint32_t mask_hi = __builtin_amdgcn_readfirstlane(0x00f000f0u);
    uint32_t magic = 0x64006400u, one16 = 0x2c002c00u;  // half2(1024), half2(1/16)
    asm volatile("" : "+v"(magic), "+v"(one16));  // keep them in VGPRs (opaque to constant propagation)
    const uint16_t* xlane = xs + 8 * r;

    auto consume = [&](int slot, bool valid) {
        const uint32_t wds[4] = {wq[slot].x, wq[slot].y, wq[slot].z, wq[slot].w};
        // an item past the end of the wave's run is neutralised by zero scales: fma(dot, 0, acc) == acc
        const uint32_t s01 = valid ? sc[slot].x : 0u, s23 = valid ? sc[slot].y : 0u;
        const uint32_t z4 = zq[slot];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t z = (z4 >> (4 * j)) & 0xfu;
            const uint32_t z1 = 0xe400e400u | z | (z << 16);                        // half2 -(1024 + z), exact
            const hv2 c960 = {(_Float16)960.f, (_Float16)960.f};
            const uint32_t z16 = __builtin_bit_cast(uint32_t, as_hv2(z1) + c960);  // half2 -(64 + z), exact
            const DeqWord d = deq_word(wds[j], z1, z16, mask_lo, mask_hi, magic, one16);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const uint4 xa = *reinterpret_cast<const uint4*>(xlane + (size_t)m * p.kp + 8 * (32 * (4 * cq + j)));
                if (j & 1) acc[m] = dot_word<true>(d, xa, j < 2 ? s01 : s23, acc[m]);
                else acc[m] = dot_word<false>(d, xa, j < 2 ? s01 : s23, acc[m]);
            }
        }
        if (valid && ++cq == Q) {
            // ---- row pair finished: replay the reference's 32-lane shuffle-down tree per half-wave and
            // park the fp32 sums in this wave's LDS slot; the epilogue runs once, after the stream.
            cq = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v = acc[m];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) v += __shfl_down(v, off, 32);
                acc[m] = 0.f;
                if (r == 0) res[((cpair - pair0) * 2 + h) * MT + m] = v;
            }
            ++cpair;
        }
    };

    if (total > 0) {
        int it = 0;
#pragma unroll 1
        for (; it + kRing < total; it += kRing) {  // steady state: every slot valid, consumed, re-issued
#pragma unroll
            for (int s = 0; s < kRing; ++s) {
                consume(s, true);
                issue(s);
            }
        }
#pragma unroll
        for (int s = 0; s < kRing; ++s) consume(s, it + s < total);  // last (possibly partial) ring
    }

    ZL_PROBE(3);
    // ---- epilogue: lane i of the wave finishes row (or gate/up pair) i of the wave's run; stores
    // of consecutive lanes are consecutive fp16 elements.
    const bool silu = (p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
    const int nres = silu ? npairs : 2 * npairs;
    if (lane < nres) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if ((m0 + m) >= p.m) break;
            const size_t orow = (size_t)(m0 + m) * p.ld_out;
            if (silu) {
                const int pr = pair0 + lane;
                if (2 * pr + 1 >= p.n) continue;
                float g = res[(2 * lane) * MT + m], u = res[(2 * lane + 1) * MT + m];
                if ((p.epi & ZL_EPI_BIAS) && p.bias) {
                    g += (float)__builtin_bit_cast(_Float16, p.bias[2 * pr]);
                    u += (float)__builtin_bit_cast(_Float16, p.bias[2 * pr + 1]);
                }
                float o;
                if (p.epi & ZL_EPI_SILU_MUL) {
                    g = (float)zl_f32_to_f16(g);  // the two fp16 linear outputs
                    u = (float)zl_f32_to_f16(u);
                    o = silu_f32(g) * u;
                } else {
                    o = (float)((double)g / (1.0 + (double)expf(-g))) * u;
                }
                p.y[orow + pr] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(o));
            } else {
                const int row = 2 * pair0 + lane;
                if (row >= p.n) continue;
                const float v = res[lane * MT + m];
                const float b = ((p.epi & ZL_EPI_BIAS) && p.bias) ? (float)__builtin_bit_cast(_Float16, p.bias[row]) : 0.f;
                float o;
                if (p.epi & ZL_EPI_ADD_C) o = ((float)__builtin_bit_cast(_Float16, p.y[orow + row]) + v) + b;
                else o = v + b;
                _Float16 y16 = zl_f32_to_f16(o);
                if (p.epi & ZL_EPI_RESIDUAL)
                    y16 = zl_f32_to_f16((float)__builtin_bit_cast(_Float16, p.residual[orow + row]) + (float)y16);
                p.y[orow + row] = __builtin_bit_cast(uint16_t, y16);
            }
        }
    }
    ZL_PROBE(4);
}

template <int MT, int kRing, int XL>
int launch(const W4Params& p, int grid_x, int grid_y, hipStream_t st) {
    size_t lds = (size_t)MT * p.kp * 2 + 64 + kWaves * 64 * MT * 4;
    if (lds > 160 * 1024) return ZL_ELIMIT;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_w4a16_gemm<MT, kRing, XL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((k_w4a16_gemm<MT, kRing, XL>), dim3(grid_x, grid_y), dim3(kThreads), lds, st, p);
    return zl_launch_status();
}

}  // namespace

extern "C" int zl_w4a16_gemm(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint16_t* scales,
                             const uint16_t* zeros, const uint16_t* bias, const uint16_t* residual, uint16_t* y,
                             int64_t m, int64_t n, int64_t k, int64_t group_size, int sym,
                             const uint16_t* norm_weight, float norm_eps, int epilogue, zl_stream_t s) {
    ZL_CHECK_ARG(x && qw && scales && (zeros || sym) && y && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(ldx >= k && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(!(epilogue & ZL_EPI_RESIDUAL) || residual, ZL_EINVAL);
    zl_w4_layout_t L;
    int st = zl_w4_layout(n, k, group_size, &L);
    if (st) return st;
    const bool silu = epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32);
    ZL_CHECK_ARG(!silu || n % 2 == 0, ZL_ESHAPE);

    W4Params p;
    p.x = x; p.ldx = ldx;
    p.qw = reinterpret_cast<const uint4*>(qw);
    p.scales = reinterpret_cast<const uint2*>(scales);
    p.zeros = zeros;
    p.bias = bias; p.residual = residual; p.y = y;
    p.norm_w = norm_weight; p.norm_eps = norm_eps;
    p.m = (int)m; p.n = (int)n; p.k = (int)k; p.kp = (int)L.kp;
    p.q_loads = (int)L.q; p.c_classes = (int)L.c;
    int per_class = 32 / (int)L.c, shift = 0;
    while ((1 << shift) < per_class) ++shift;
    p.c_shift = shift;
    p.sym = sym; p.epi = epilogue;
    p.pairs_total = (int)(L.np / 2);
    p.ld_out = silu ? (int)(n / 2) : (int)n;

    // rows-per-pass: the VALU budget of the reference-exact arithmetic holds up to ~3 rows per
    // weight pass at HBM speed; LDS must hold MT * Kp halfs
    int mt = m >= 3 ? 4 : (int)m;
    while (mt > 1 && (size_t)mt * L.kp * 2 + 64 + 4096 > 64 * 1024) mt >>= 1;
    const int grid_y = (int)((m + mt - 1) / mt);

    // grid: 16 wavefronts per CU (two 8-wave workgroups; ~100 VGPRs -> 4 waves per SIMD), each wave owns
    // a contiguous run of row pairs; small matrices get one workgroup per CU.
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    int wgs_per_cu = (p.pairs_total + cus * kWaves - 1) / (cus * kWaves) >= 2 ? 2 : 1;
    int best_ppw = (p.pairs_total + cus * wgs_per_cu * kWaves - 1) / (cus * wgs_per_cu * kWaves);
    if (best_ppw < 1) best_ppw = 1;
    if (best_ppw > 32) best_ppw = 32;  // LDS result slots: 64 rows per wave
    p.pairs_per_wave = best_ppw;
    const int waves_needed = (p.pairs_total + best_ppw - 1) / best_ppw;
    const int grid_x = (waves_needed + kWaves - 1) / kWaves;

    hipStream_t hs = (hipStream_t)s;
    // ring depth: never more loads in flight than the wave has items (no dummy loads for tiny runs);
    // XL: x loads per thread kept in registers (2 -> K <= 4096, 8 -> K <= 16384 with few rows, else 0)
    const bool small = (int64_t)best_ppw * L.q < 8;
    const int xl_need = (int)((L.kp + kThreads * 8 - 1) / (kThreads * 8));
    const int xl = xl_need <= 1 ? 1 : (xl_need <= 2 ? 2 : ((xl_need <= 4 && mt <= 2) ? 4 : 0));
#define ZL_W4_LAUNCH(MT)                                                                     \
    if (xl == 1)                                                                             \
        return small ? launch<MT, 4, 1>(p, grid_x, grid_y, hs) : launch<MT, 8, 1>(p, grid_x, grid_y, hs); \
    if (xl == 2) return launch<MT, 8, 2>(p, grid_x, grid_y, hs);                             \
    if (xl == 4) return launch<MT, 8, 4>(p, grid_x, grid_y, hs);                             \
    return small ? launch<MT, 4, 0>(p, grid_x, grid_y, hs) : launch<MT, 8, 0>(p, grid_x, grid_y, hs);
    switch (mt) {
        case 1: ZL_W4_LAUNCH(1)
        case 2: ZL_W4_LAUNCH(2)
        default: ZL_W4_LAUNCH(4)
    }
#undef ZL_W4_LAUNCH
}

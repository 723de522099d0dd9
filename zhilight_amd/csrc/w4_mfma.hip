// w4_mfma.hip -- W4A16 GEMM for decode batches on the matrix cores (M <= 16 per pass, any M by passes).
//
// Which reference arithmetic this is: NOT the M <= 40 warp-reduce kernel (fp16 partial dots; that one is
// replayed bit-exactly by w4_gemv.hip) but the fp32-accumulating flavour the reference itself switches
// to for M > 40 (dequant_k_major + fp32-compute GEMM, src/nn/quant/gptq/q_gemm_k_major.cu:1083-1100):
//   y[m,n] = half( sum_g s[n,g] * ( sum_{k in g} x[m,k] * (q[n,k] - z[n,g]) ) + bias ),
// products exact (fp16 x small integer in fp32), group sums accumulated in fp32 inside the MFMA,
// one fp32 fma per group with the scale.  It differs from the exact result by fp32 rounding only
// (~1e-6 rel) and from the warp-reduce kernel by that kernel's own fp16 noise (~1e-3 of the output rms).
//
// Why it exists: the bit-exact VALU kernel needs ~97 VALU ops per KiB of weights and is VALU-issue bound
// (rocprofv3 / tools/ubench/valu_rate.hip); here the 8-weight dot products run on the MFMA pipe
// (v_mfma_f32_16x16x32_f16), the VALU only extracts nibbles and subtracts the zero (~46 ops per KiB),
// and up to 16 activation rows cost the same as one -- the reference re-reads the weights per 16-row
// chunk and is ALU bound there.
//
// Measured on MI355X (gate|up 28672x4096, M = 1; tools/bench_gemv.py --mfma, tools/ubench/variant.sh
// ablations, tools/ubench/probe_mfma.py):  14.3 us = 12.5 us for the bare stream of this structure
// (-DZL_EXP_NOCOMPUTE) + 1.8 us of exposed compute.  What got it there, in order of weight:
//   * address-space-pure B/A reads (a pointer that may be LDS or global becomes FLAT: vmcnt(0) per item),
//   * no ring PHIs between a skippable loop and its tail (preheader copies wait for vmcnt(0)),
//   * buffer loads with a scalar stream offset (no per-item vector address math, no waterfall loops),
//   * MFMAs chained back to back after the whole dequant, scale applied one step later,
//   * one workgroup per CU with the k-split chosen so every SIMD streams the same number of items;
//     a deeper ring (16 KiB per wave) or 16 waves per CU were both SLOWER: a CU sustains ~10 B/clk of
//     HBM misses whatever is queued behind them.
//
// ZLW4M layout (zl_w4m_pack), tile = 16 output rows x 128 k, 1 KiB of nibbles + 64 B of meta:
//   qw   : u32 [N/16][Kp/128][64 lanes][4]   lane (n = lane & 15, kq = lane >> 4), element t holds the word
//          with k = 128 g + 32 t + 8 kq .. +7  == the B fragment of MFMA step t, fetched as ONE dwordx4
//   meta : u32 [N/16][Kp/128][16]            row n: f16 scale (low half) | f16 -(1024 + zero) (high half),
//          the group's pair repeated on each of its 128-k items (group_size = any multiple of 128)
// The activations are the A operand (row m = lane & 15), so a lane's C registers are column n = lane & 15,
// rows m = 4 kq + i: the row a lane dequantises is the row whose scale it applies -- one 4-byte meta load.
// A workgroup of 8 wavefronts owns a run of row tiles; wave w streams k-slice (w % ks) of tile (w / ks)
// (split-K inside the workgroup, fixed-order LDS reduction: deterministic), x is staged once per workgroup
// in LDS (padded rows: conflict-free ds_read_b128 fragments) with the optional fused RMSNorm.
#include <stdlib.h>
#include "zl_common.h"
#include "zl_stage.h"

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kRing = 8;
constexpr int kMaxRounds = 8;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

// ---- optional phase-timestamp probe (build with -DZL_W4M_PROBE; tools/ubench/probe_mfma.py) -----------
#ifdef ZL_W4M_PROBE
__device__ unsigned long long* zl_probe_m = nullptr;  // [waves][8] wall-clock ticks (100 MHz)
#define ZL_PROBE(slot)                                                                                 \
    do {                                                                                               \
        if (zl_probe_m && lane == 0) zl_probe_m[(size_t)(blockIdx.x * kWaves + wave) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define ZL_PROBE(slot) do {} while (0)
#endif

struct MfmaParams {
    const uint16_t* x;
    int64_t ldx;
    const uint4* qw;
    const uint32_t* meta;    // [tile][g][16]
    uint32_t qw_bytes, meta_bytes;
    const uint16_t* bias;
    const uint16_t* residual;
    uint16_t* y;
    const uint16_t* norm_w;
    float norm_eps;
    int m, n, k, kp;         // kp = K rounded up to 128
    int groups;              // kp / 128 items per row tile
    int ks;                  // k-slices per row tile (power of two <= kWaves): waves cooperating on a tile
    int items_per_slice;     // ceil(groups / ks)
    int tiles;               // row tiles (16 rows)
    int rounds;              // a workgroup handles kWaves/ks tiles per round, `rounds` rounds
    int epi, ld_out;
    int lds_stride;          // kp + 8 halfs
};

__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask_s, uint32_t magic_v) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask_s), "v"(magic_v));
    return r;
}

// word -> 8 fp16 (q - z), exact; natural k order (w0..w7) = MFMA A-fragment element order
__device__ __forceinline__ h8 dequant_word(uint32_t w, hv2 z1, hv2 z16, uint32_t mask_lo, uint32_t mask_hi,
                                           uint32_t magic) {
    const hv2 one16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    const hv2 d0 = __builtin_bit_cast(hv2, and_or(w, mask_lo, magic)) + z1;
    const hv2 d1 = __builtin_elementwise_fma(__builtin_bit_cast(hv2, and_or(w, mask_hi, magic)), one16, z16);
    const uint32_t wb = w >> 8;
    const hv2 d2 = __builtin_bit_cast(hv2, and_or(wb, mask_lo, magic)) + z1;
    const hv2 d3 = __builtin_elementwise_fma(__builtin_bit_cast(hv2, and_or(wb, mask_hi, magic)), one16, z16);
    h8 a;
    a[0] = d0.x; a[1] = d0.y; a[2] = d1.x; a[3] = d1.y; a[4] = d2.x; a[5] = d2.y; a[6] = d3.x; a[7] = d3.y;
    return a;
}

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + expf(-x)); }

// Streaming structure = the dense GEMV's (which reaches 6.5 TB/s): every wave runs an 8-deep ring of
// 1 KiB non-temporal loads over its own (row tile, k-slice) items with NO barrier inside the stream;
// the partial C fragments of all rounds are parked in LDS and reduced once, after the stream.
//
// XMODE: how the B operand (activations) reaches the MFMA.
//   1,2,4   : the M x K activations are <= XMODE 16-byte chunks per thread: they are loaded into registers
//             FIRST (so vmcnt, which retires in order, releases them before the weight ring), the weight
//             ring is issued, and only then x is normalised/stored to LDS -- the staging hides entirely
//             behind the first HBM round trip.
//   0       : larger M x K that still fits LDS: generic stage-then-stream order.
//   -1      : M x K beyond LDS: B fragments straight from global memory (L2), no fused norm.
// The LDS and global B paths are separate instantiations on purpose: a pointer that may be either
// becomes a FLAT access, and flat loads force vmcnt(0)+lgkmcnt(0) before every use -- that drained the
// weight ring at each item.
// B columns m >= M are never stored, so their lanes simply re-read row 0 (each C element is an
// independent dot product: garbage columns cannot contaminate live ones).
template <int XMODE>
__global__ __launch_bounds__(kThreads, 4) void k_w4a16_mfma(const MfmaParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [ x rows (optional) | red: rounds x kWaves x (16 * ceil(M/4)) live lanes x f4 | block-reduce scratch ]
    constexpr bool XLDS = XMODE >= 0;
    constexpr int XL = XMODE > 0 ? XMODE : 1;
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem);
    const size_t x_bytes = XLDS ? (size_t)p.m * p.lds_stride * 2 : 0;
    f4* red = reinterpret_cast<f4*>(smem + x_bytes);
    const int live_lanes = 16 * ((p.m + 3) / 4);     // C rows m = 4 kq + i: lanes with 4 kq >= M hold nothing
    float* scratch = reinterpret_cast<float*>(smem + x_bytes + (size_t)p.rounds * kWaves * live_lanes * 16);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrow = lane & 15, kq = lane >> 4;
    // (readfirstlane: the divisions by the runtime k-split are expanded on the VALU, which makes every
    //  value derived from them "divergent" to the compiler -- and a divergent buffer soffset costs a
    //  waterfall loop per load)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int tpr = uni(kWaves / p.ks);              // tiles per round
    const int slice = uni(wave % p.ks), tsub = uni(wave / p.ks);
    const int g0 = slice * p.items_per_slice;
    const int g1 = min(p.groups, g0 + p.items_per_slice);
    const int nit = g1 > g0 ? g1 - g0 : 0;           // items of this wave per round
    const int tile_base = blockIdx.x * p.rounds * tpr + tsub;  // tile of round r = tile_base + r * tpr
    // flattened item stream of this wave: round r, item j -> (tile_base + r*tpr, g0 + j)
    int live_rounds = tile_base < p.tiles ? uni((p.tiles - tile_base + tpr - 1) / tpr) : 0;
    live_rounds = live_rounds < p.rounds ? live_rounds : p.rounds;
    const int total = live_rounds * nit;
    ZL_PROBE(0);

    // ---- x (and the norm weight) into registers first: chunk c of a thread = row c / cpr, halfs
    //      8 * tid + 4096 * (c % cpr) .. +7 (the zl_stage_rows assignment, so the sum of squares below
    //      runs in the same order as the stand-alone RMSNorm kernel)
    const int cpr = (p.kp + kThreads * 8 - 1) / (kThreads * 8);
    uint4 xr[XL], nwr[XL];
    if constexpr (XMODE > 0) {
        int cm = 0, cj = 0;
#pragma unroll
        for (int c = 0; c < XL; ++c) {
            const int idx = threadIdx.x * 8 + cj * (kThreads * 8);
            const bool live = cm < p.m && idx < p.k;
            const uint16_t* src = p.x + (live ? (size_t)cm * p.ldx + idx : 0);
            xr[c] = *reinterpret_cast<const uint4*>(src);
            if (!live) xr[c] = make_uint4(0, 0, 0, 0);
            if (p.norm_w) nwr[c] = *reinterpret_cast<const uint4*>(p.norm_w + (live ? idx : 0));
            if (++cj == cpr) {
                cj = 0;
                ++cm;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (XMODE == 0) {
        for (int m = 0; m < p.m; ++m) {
            const uint16_t* xrow = p.x + (size_t)m * p.ldx;
            uint16_t* xd = xs + (size_t)m * p.lds_stride;
            float ss = 0.f;
            for (int i = threadIdx.x * 8; i < p.kp; i += kThreads * 8) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (i < p.k) v = *reinterpret_cast<const uint4*>(xrow + i);
                if (p.norm_w) {
                    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const hv2 hh = __builtin_bit_cast(hv2, u[e]);
                        ss = __builtin_fmaf((float)hh.x, (float)hh.x, ss);
                        ss = __builtin_fmaf((float)hh.y, (float)hh.y, ss);
                    }
                }
                *reinterpret_cast<uint4*>(xd + i) = v;
            }
            if (p.norm_w) {
                ss = zl_block_sum(ss, scratch);
                const float rs = zl_rsqrt_rn(ss / (float)p.k + p.norm_eps);
                for (int i = threadIdx.x * 8; i < p.k; i += kThreads * 8) {
                    const uint4 v = *reinterpret_cast<uint4*>(xd + i);
                    const uint4 wv = *reinterpret_cast<const uint4*>(p.norm_w + i);
                    uint32_t u[4] = {v.x, v.y, v.z, v.w};
                    const uint32_t wu[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const hv2 hh = __builtin_bit_cast(hv2, u[e]), ww = __builtin_bit_cast(hv2, wu[e]);
                        hv2 o;
                        o.x = zl_f32_to_f16((float)hh.x * rs * (float)ww.x);
                        o.y = zl_f32_to_f16((float)hh.y * rs * (float)ww.y);
                        u[e] = __builtin_bit_cast(uint32_t, o);
                    }
                    *reinterpret_cast<uint4*>(xd + i) = make_uint4(u[0], u[1], u[2], u[3]);
                }
            }
        }
        __syncthreads();
    }

    // ---- weight ring prologue.  Clamped, branch-free.  Buffer loads split the address the way the
    //      stream is shaped: descriptor base + per-lane VGPR offset (constant) + SCALAR stream offset
    //      (items are contiguous inside a round, a round change is one bigger stride) -- no vector
    //      address arithmetic at all, ~8 SALU per item.
    uint4 wq[kRing];
    uint32_t mt[kRing];
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.qw), 0, p.qw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.meta), 0, p.meta_bytes, 0x00020000);
    const int t0c = tile_base < p.tiles ? tile_base : p.tiles - 1;
    const int g0c = g0 < p.groups ? g0 : p.groups - 1;
    const uint32_t it0 = (uint32_t)t0c * (uint32_t)p.groups + (uint32_t)g0c;
    uint32_t qs = it0 * 1024u, ms = it0 * 64u;       // scalar byte offsets of the next item to issue
    const int round_jump = tpr * p.groups - (nit - 1);
    const uint32_t q_off = (uint32_t)lane * 16u, m_off = (uint32_t)nrow * 4u;
    int iss_left = total - 1, iss_jrem = nit;
    // exhausted stream: refills go through a zero-length descriptor (out-of-range buffer loads return zeros without a
    // memory access; re-reading the last item kept kRing - 1 loads per wave in flight when the kernel should retire)
    const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.qw), 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rqc = total > 0 ? rq : rnull, rmc = total > 0 ? rm : rnull;
    auto issue = [&](int slot) {
        wq[slot] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rqc, q_off, qs, 2 /* nt */));
#ifndef ZL_EXP_NOMETA
        mt[slot] = __builtin_amdgcn_raw_buffer_load_b32(rmc, m_off, ms, 2);
#else
        mt[slot] = 0xec083c00u + ms;
#endif
        // one compare per select keeps these on the scalar unit (s_cmp + s_cselect_b32); combining two
        // conditions into a bool makes the compiler build lane masks and do the select on the VALU
        const bool more = iss_left > 0;
        const int adv = more ? 1 : 0;
        rqc = more ? rqc : rnull;
        rmc = more ? rmc : rnull;
        --iss_left;
        const int wrap = iss_jrem == 1 ? adv : 0;
        iss_jrem = wrap != 0 ? nit : iss_jrem - adv;
#ifdef ZL_EXP_SAMEITEM
        const int d = 0 * wrap;
#else
        const int d = wrap != 0 ? round_jump : adv;
#endif
        qs += (uint32_t)d * 1024u;
        ms += (uint32_t)d * 64u;
    };
    // The prologue fills slots 0 .. kRing-2; the last slot starts as the neutral "previous item" (scale 0)
    // of the first step, which refills it with item kRing-1 -- every step then has the same shape
    // (no peeled first step: a peel makes the compiler re-assign the ring registers in the loop
    // preheader, and those copies wait for vmcnt(0), i.e. drain the whole ring once).
#pragma unroll
    for (int s = 0; s < kRing - 1; ++s) {
        issue(s);
        __builtin_amdgcn_sched_barrier(0);
    }
    mt[kRing - 1] = 0;
    wq[kRing - 1] = make_uint4(0, 0, 0, 0);
    ZL_PROBE(1);

    // ---- registers -> LDS (optionally RMS-normalised), padded rows
    if constexpr (XMODE > 0) {
        if (p.norm_w) {
            // per-thread chain over the row's chunks, 64-lane butterfly, waves in order: zl_block_sum's order
            float part[XL];
            float run = 0.f;
            int cj = 0;
#pragma unroll
            for (int c = 0; c < XL; ++c) {
                if (cj == 0) run = 0.f;
                const uint32_t u[4] = {xr[c].x, xr[c].y, xr[c].z, xr[c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const hv2 hh = __builtin_bit_cast(hv2, u[e]);
                    run = __builtin_fmaf((float)hh.x, (float)hh.x, run);
                    run = __builtin_fmaf((float)hh.y, (float)hh.y, run);
                }
                part[c] = zl_wave_sum(run);   // meaningful on the row's last chunk
                if (++cj == cpr) cj = 0;
            }
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < XL; ++c) scratch[c * kWaves + wave] = part[c];
            }
            __syncthreads();
        }
        int cm = 0, cj = 0;
#pragma unroll
        for (int c = 0; c < XL; ++c) {
            const int idx = threadIdx.x * 8 + cj * (kThreads * 8);
            uint4 v = xr[c];
            if (p.norm_w) {
                const int last = (cm + 1) * cpr - 1;
                float tot = 0.f;
                if (last < XL) {
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) tot += scratch[last * kWaves + w];
                }
                const float rs = zl_rsqrt_rn(tot / (float)p.k + p.norm_eps);
                uint32_t u[4] = {v.x, v.y, v.z, v.w};
                const uint32_t wu[4] = {nwr[c].x, nwr[c].y, nwr[c].z, nwr[c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const hv2 hh = __builtin_bit_cast(hv2, u[e]), ww = __builtin_bit_cast(hv2, wu[e]);
                    hv2 o;
                    o.x = zl_f32_to_f16((float)hh.x * rs * (float)ww.x);
                    o.y = zl_f32_to_f16((float)hh.y * rs * (float)ww.y);
                    u[e] = __builtin_bit_cast(uint32_t, o);
                }
                v = make_uint4(u[0], u[1], u[2], u[3]);
            }
            if (cm < p.m && idx < p.kp) *reinterpret_cast<uint4*>(xs + (size_t)cm * p.lds_stride + idx) = v;
            if (++cj == cpr) {
                cj = 0;
                ++cm;
            }
        }
        __syncthreads();
    }

    ZL_PROBE(2);
    const uint32_t mask_lo = __builtin_amdgcn_readfirstlane(0x000f000fu);
    const uint32_t mask_hi = __builtin_amdgcn_readfirstlane(0x00f000f0u);
    uint32_t magic = 0x64006400u;
    asm volatile("" : "+v"(magic));
    const int brow = nrow < p.m ? nrow : 0;          // A fragment row m = lane & 15
    const uint16_t* xl_lds = xs + (size_t)brow * p.lds_stride + 8 * kq;
    const uint16_t* xl_glb = p.x + (size_t)brow * p.ldx + 8 * kq;

    // One ring step = [dequantise item i: ~40 VALU] [scale-accumulate item i-1: 4 VALU] [4 MFMAs of item
    // i, back to back].  Rules this order follows (MI355X measurements): MFMAs chained on one accumulator
    // only run at ~17 cycles each when NOTHING is issued between them (one VALU in the gap costs ~+43
    // cycles), so the dequant of all four words is finished first; the group sum is read one step later,
    // when the matrix pipe has long drained, so no wave ever sits in the MFMA->VALU hazard window;
    // the f16 scale is widened inside v_fma_mix_f32 (packed-f32 VALU beside MFMAs is slower than scalar).
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4 accg_prev = {0.f, 0.f, 0.f, 0.f};
    int cj = 0, ej = -1, er = 0;                     // ej = -1: the first scale-accumulate is the neutral one
    uint32_t xoff = (uint32_t)g0 * 256u;             // byte offset of the wave's current group in an x row
    // scale-accumulate of the PREVIOUS item; its ring slot is refilled only after this (the slot's meta
    // register is read here), so the ring runs 7 deep on the refill side
    auto finish_prev = [&](int pslot) {
        asm("v_fma_mix_f32 %0, %4, %8, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n\t"
            "v_fma_mix_f32 %1, %5, %8, %1 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n\t"
            "v_fma_mix_f32 %2, %6, %8, %2 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n\t"
            "v_fma_mix_f32 %3, %7, %8, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
            : "v"(accg_prev[0]), "v"(accg_prev[1]), "v"(accg_prev[2]), "v"(accg_prev[3]), "v"(mt[pslot]));
        if (++ej == nit) {
            if (lane < live_lanes) red[((size_t)er * kWaves + wave) * live_lanes + lane] = acc;   // round's partial C
            acc = (f4){0.f, 0.f, 0.f, 0.f};
            ej = 0;
            ++er;
        }
    };
    auto step = [&](int slot, int pslot, bool has_prev, bool refill) {
        uint4 bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#ifdef ZL_EXP_NOLDS
            if constexpr (false)
#else
            if constexpr (XLDS)
#endif
                bv[t] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(xl_lds) + xoff + 64 * t);
#ifndef ZL_EXP_NOLDS
            else
                bv[t] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(xl_glb) + xoff + 64 * t);
#endif
        }
#ifdef ZL_EXP_NOLDS
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = make_uint4(0x3c003c00u + xoff, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + t);
#endif
        if (++cj == nit) {
            cj = 0;
            xoff = (uint32_t)g0 * 256u;
        } else {
            xoff += 256u;
        }
        // -(1024 + z) in both halves (the meta word's high half), and the same minus 960 for the x16 nibbles
        const hv2 z1 = __builtin_bit_cast(hv2, __builtin_amdgcn_perm(mt[slot], mt[slot], 0x03020302u));
        const hv2 c960 = {(_Float16)960.f, (_Float16)960.f};
        const hv2 z16 = z1 + c960;
        const uint32_t wds[4] = {wq[slot].x, wq[slot].y, wq[slot].z, wq[slot].w};
        h8 a[4];
#pragma unroll
#ifdef ZL_EXP_NODEQ
        for (int t = 0; t < 4; ++t) {
            const uint4 u = make_uint4(wds[t], wds[(t + 1) & 3], __builtin_bit_cast(uint32_t, z1), __builtin_bit_cast(uint32_t, z16));
            a[t] = __builtin_bit_cast(h8, u);
        }
#else
        for (int t = 0; t < 4; ++t) a[t] = dequant_word(wds[t], z1, z16, mask_lo, mask_hi, magic);
#endif
        if (has_prev) finish_prev(pslot);
        __builtin_amdgcn_sched_barrier(0);
        f4 accg = {0.f, 0.f, 0.f, 0.f};
#ifdef ZL_EXP_NOCOMPUTE
        accg[0] = __builtin_bit_cast(float, wds[0] ^ wds[1] ^ wds[2] ^ wds[3]);
#elif defined(ZL_EXP_NOMFMA)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint4 au = __builtin_bit_cast(uint4, a[t]);
            accg[t] = __builtin_bit_cast(float, au.x ^ au.y ^ au.z ^ au.w ^ bv[t].x ^ bv[t].y ^ bv[t].z ^ bv[t].w);
        }
#else
#pragma unroll
        for (int t = 0; t < 4; ++t)
            accg = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, bv[t]), a[t], accg, 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        accg_prev = accg;
        if (refill) issue(pslot);
    };
    // The long-stream and the short-stream paths are kept completely apart after the prologue: if the
    // loop could be skipped, the ring registers of "prologue -> tail" and "loop -> tail" would meet in
    // PHIs and the compiler resolves those with copies in the loop preheader -- copies that wait for
    // vmcnt(0), i.e. drain the whole ring once before the first item is consumed.
    auto tail = [&](int k) {
        // the last < kRing items: nothing is issued any more, so wave-uniform branches are harmless here
#pragma unroll
        for (int s = 0; s < kRing - 1; ++s) {
            if (k + s < total) step(s, (s + kRing - 1) % kRing, true, false);
        }
        const int last = (total - 1) % kRing;        // static register indices only: a dynamic one would
#pragma unroll                                       // push the whole ring into scratch memory
        for (int s = 0; s < kRing; ++s) {
            if (last == s) finish_prev(s);
        }
    };
    if (total >= kRing) {
        int k = 0;
#pragma unroll 1
        do {
#pragma unroll
            for (int s = 0; s < kRing; ++s) step(s, (s + kRing - 1) % kRing, true, true);
            k += kRing;
            if (k == kRing) ZL_PROBE(3);
        } while (k + kRing <= total);
        tail(k);
    } else if (total > 0) {
        tail(0);
    }
    ZL_PROBE(4);
    __syncthreads();
    ZL_PROBE(5);

    // ---- split-K reduction across the ks slices of each tile (fixed order: deterministic) + epilogue.
    //      C fragment: lane = (m >> 2) * 16 + n_local, register m & 3.
    const bool silu = (p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
    const float* redf = reinterpret_cast<const float*>(red);
    const int per_tile = (silu ? 8 : 16) * p.m;
    const int nouts = p.rounds * tpr * per_tile;
    for (int o = threadIdx.x; o < nouts; o += kThreads) {
        const int tslot = o / per_tile, rem = o % per_tile;
        const int r = tslot / tpr, ts = tslot % tpr;
        const int tile = blockIdx.x * p.rounds * tpr + r * tpr + ts;
        if (tile >= p.tiles) continue;
        auto total_of = [&](int n_local, int m) {
            const int ln = (m >> 2) * 16 + n_local, i = m & 3;
            float v = 0.f;
            for (int sl = 0; sl < p.ks; ++sl) {
                const int w = ts * p.ks + sl;
                if (sl * p.items_per_slice < p.groups)  // slices without items never wrote their slot
                    v += redf[(((size_t)r * kWaves + w) * live_lanes + ln) * 4 + i];
            }
            return v;
        };
        if (!silu) {
            const int m = rem >> 4, n_local = rem & 15;
            const int row = tile * 16 + n_local;
            if (row < p.n) {
                const float v = total_of(n_local, m);
                const size_t orow = (size_t)m * p.ld_out;
                const float b = ((p.epi & ZL_EPI_BIAS) && p.bias) ? (float)__builtin_bit_cast(_Float16, p.bias[row]) : 0.f;
                float ov;
                if (p.epi & ZL_EPI_ADD_C) ov = ((float)__builtin_bit_cast(_Float16, p.y[orow + row]) + v) + b;
                else ov = v + b;
                _Float16 y16 = zl_f32_to_f16(ov);
                if (p.epi & ZL_EPI_RESIDUAL)
                    y16 = zl_f32_to_f16((float)__builtin_bit_cast(_Float16, p.residual[orow + row]) + (float)y16);
                p.y[orow + row] = __builtin_bit_cast(uint16_t, y16);
            }
        } else {
            const int m = rem >> 3, j = rem & 7;
            const int pr = tile * 8 + j;
            if (2 * pr + 1 < p.n) {
                float g = total_of(2 * j, m), u = total_of(2 * j + 1, m);
                if ((p.epi & ZL_EPI_BIAS) && p.bias) {
                    g += (float)__builtin_bit_cast(_Float16, p.bias[2 * pr]);
                    u += (float)__builtin_bit_cast(_Float16, p.bias[2 * pr + 1]);
                }
                float ov;
                if (p.epi & ZL_EPI_SILU_MUL) {
                    g = (float)zl_f32_to_f16(g);
                    u = (float)zl_f32_to_f16(u);
                    ov = silu_f32(g) * u;
                } else {
                    ov = (float)((double)g / (1.0 + (double)expf(-g))) * u;
                }
                p.y[(size_t)m * p.ld_out + pr] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(ov));
            }
        }
    }
    ZL_PROBE(6);
}

// ---- packer: k-major (N,K/8) words, (N,K/G) u8 zeros, (N,K/G) f16 scales -> ZLW4M
__global__ void k_pack_m_qw(const uint32_t* __restrict__ qw_km, uint32_t* __restrict__ dst, int64_t n, int64_t k8,
                            int64_t tiles, int64_t groups, int interleave) {
    const int64_t total = tiles * groups * 256;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const int64_t tg = i >> 8, g = tg % groups, tile = tg / groups;
        const int64_t row = tile * 16 + (lane & 15);
        const int64_t word = g * 16 + 4 * t + (lane >> 4);
        const int64_t src_row = interleave ? ((row & 1) * (n / 2) + (row >> 1)) : row;
        uint32_t v = 0;
        if (row < n && word < k8) v = qw_km[src_row * k8 + word];
        dst[i] = v;
    }
}

__global__ void k_pack_m_meta(const uint8_t* __restrict__ qz_km, const uint16_t* __restrict__ sc_km,
                              uint32_t* __restrict__ meta, int64_t n, int64_t ng, int64_t group_items, int64_t tiles,
                              int64_t groups, int interleave) {
    const int64_t total = tiles * groups * 16;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i & 15);
        const int64_t tg = i >> 4, g = tg % groups, tile = tg / groups;
        const int64_t grp = g / group_items, row = tile * 16 + r;
        const int64_t src_row = interleave ? ((row & 1) * (n / 2) + (row >> 1)) : row;
        uint32_t sc = 0, z = 0;
        if (row < n && grp < ng) {
            sc = sc_km[src_row * ng + grp];
            z = qz_km[src_row * ng + grp] & 0xf;
        }
        meta[i] = sc | ((0xe400u | z) << 16);   // f16 scale | f16 -(1024 + z)
    }
}

// ---- inverse of the packer: ZLW4M -> the k-major operands (every word / row is read from the slot k_pack_m_* wrote it to)
__global__ void k_unpack_m_qw(const uint32_t* __restrict__ src, uint32_t* __restrict__ qw_km, int64_t n, int64_t k8,
                              int64_t tiles, int64_t groups, int interleave) {
    const int64_t total = tiles * groups * 256;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const int64_t tg = i >> 8, g = tg % groups, tile = tg / groups;
        const int64_t row = tile * 16 + (lane & 15);
        const int64_t word = g * 16 + 4 * t + (lane >> 4);
        const int64_t dst_row = interleave ? ((row & 1) * (n / 2) + (row >> 1)) : row;
        if (row < n && word < k8) qw_km[dst_row * k8 + word] = src[i];
    }
}

__global__ void k_unpack_m_meta(const uint32_t* __restrict__ meta, uint8_t* __restrict__ qz_km, uint16_t* __restrict__ sc_km,
                                int64_t n, int64_t ng, int64_t group_items, int64_t tiles, int64_t groups, int interleave) {
    const int64_t total = tiles * groups * 16;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i & 15);
        const int64_t tg = i >> 4, g = tg % groups, tile = tg / groups;
        if (g % group_items != 0) continue;            // the pair is repeated on each 128-k item of its group
        const int64_t grp = g / group_items, row = tile * 16 + r;
        const int64_t dst_row = interleave ? ((row & 1) * (n / 2) + (row >> 1)) : row;
        if (row < n && grp < ng) {
            const uint32_t v = meta[i];
            sc_km[dst_row * ng + grp] = (uint16_t)(v & 0xffffu);
            qz_km[dst_row * ng + grp] = (uint8_t)((v >> 16) & 0xfu);
        }
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

// w4_phase.hip: the phase-pipelined streaming kernel for 5..32 rows
int zl_w4a16_gemm_phase(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                        uint32_t meta_bytes, const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n,
                        int k, int groups, int tiles, int epilogue, int ld_out, const uint16_t* norm_w, float norm_eps,
                        const zl_w4_opts_t* opts, hipStream_t hs);

// w4_slab.hip: 9..32 rows on 128-column x K-slice tiles (round 6); ZL_ESHAPE = not this shape / no scratch for the K split
int zl_w4a16_gemm_slab(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                       const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups, int tiles,
                       int epilogue, int ld_out, const uint16_t* norm_w, float norm_eps, const zl_w4_opts_t* opts, hipStream_t hs);
int zl_w4a16_gemm_slab_rope(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                            uint32_t meta_bytes, const uint16_t* bias, int m, int n, int k, int groups, int tiles, const float* cosv,
                            const float* sinv, const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                            uint16_t* const* v_bufs, uint16_t* q_out, int h, int hkv, int d, int bshd, const uint16_t* norm_w,
                            float norm_eps, const zl_w4_opts_t* opts, hipStream_t hs);

bool zl_w4a16_i8p_covers(int64_t m, int64_t k);
int zl_w4a16_gemm_i8p(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                      uint32_t meta_bytes, const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k,
                      int groups, int tiles, int epilogue, int ld_out, const uint16_t* norm_w, float norm_eps, int rounds_override,
                      hipStream_t hs);
int zl_w4a16_gemm_i8p_rope(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                           uint32_t meta_bytes, const uint16_t* bias, int m, int n, int k, int groups, int tiles,
                           const uint16_t* norm_w, float norm_eps, const float* cosv, const float* sinv,
                           const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                           uint16_t* const* v_bufs, uint16_t* q_out, int h, int hkv, int d, int bshd, hipStream_t hs);
int zl_w4a16_gemm_i8p_merge(const void* ws, const int32_t* buf_lens, const int32_t* valid_lens, int split_len, int max_splits,
                            const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                            const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups,
                            int tiles, int epilogue, hipStream_t hs);
#ifdef ZL_EXPERIMENTAL
// the loader / consumer engine (w4_engine.hip; experimental build only)
bool zl_w4_engine_covers(int64_t m, int64_t k, int r);
int zl_w4a16_gemm_engine(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                         uint32_t meta_bytes, const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k,
                         int groups, int tiles, int epilogue, int ld_out, const uint16_t* norm_w, float norm_eps, int slots_cap,
                         hipStream_t hs);
int zl_w4a16_gemm_engine_rope(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                              uint32_t meta_bytes, const uint16_t* bias, int m, int n, int k, int groups, int tiles,
                              const uint16_t* norm_w, float norm_eps, const float* cosv, const float* sinv,
                              const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                              uint16_t* const* v_bufs, uint16_t* q_out, int h, int hkv, int d, int bshd, hipStream_t hs);
int zl_w4a16_gemm_engine_merge(const void* ws, const int32_t* buf_lens, const int32_t* valid_lens, int split_len, int max_splits,
                               const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                               const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups,
                               int tiles, int epilogue, hipStream_t hs);
int zl_w4_engine_o_gateup_launch(const void* ws, const int32_t* buf_lens, const int32_t* valid_lens, int split_len, int max_splits,
                                 const uint32_t* qw1, const uint32_t* meta1, uint32_t qw1_bytes, uint32_t meta1_bytes,
                                 const uint16_t* bias1, uint16_t* hidden, int m, int n1, int k1, int groups1, int tiles1,
                                 const uint32_t* qw2, const uint32_t* meta2, uint32_t qw2_bytes, uint32_t meta2_bytes,
                                 const uint16_t* bias2, const uint16_t* norm_w, float norm_eps, uint16_t* act, int n2, int groups2,
                                 int tiles2, int epilogue2, void* granules, const uint32_t* epoch_ptr, uint32_t epoch_add,
                                 uint32_t* err, hipStream_t hs);
int zl_engine_epoch_advance_launch(uint32_t* epoch, uint32_t by, hipStream_t hs);
#endif
int zl_w4a16_gemm_phase_merge(const float* ws, const int32_t* buf_lens, const int32_t* valid_lens, int split_len,
                              int max_splits, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                              uint32_t meta_bytes, const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n,
                              int k, int groups, int tiles, int epilogue, hipStream_t hs);

int zl_w4a16_gemm_phase_rope(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                             uint32_t meta_bytes, const uint16_t* bias, int m, int n, int k, int groups, int tiles,
                             const uint16_t* norm_w, float norm_eps, const float* cosv, const float* sinv,
                             const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                             uint16_t* const* v_bufs, uint16_t* q_out, int h, int hkv, int d, int bshd, hipStream_t hs);

#ifdef ZL_EXPERIMENTAL
int64_t zl_w4_planes_bytes_(int64_t m, int64_t k);
int zl_w4_planes_launch(const uint16_t* x, int64_t ldx, int m, int k, const uint16_t* norm_w, float norm_eps, void* planes, hipStream_t hs);
int zl_w4a16_gemm_phase_planes(const void* planes, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                               const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups, int tiles,
                               int epilogue, int ld_out, const zl_w4_opts_t* opts, hipStream_t hs);
int zl_w4a16_gemm_phase_planes_rope(const void* planes, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                                    const uint16_t* bias, int m, int n, int k, int groups, int tiles, const float* cosv, const float* sinv,
                                    const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs, uint16_t* const* v_bufs,
                                    uint16_t* q_out, int h, int hkv, int d, int bshd, hipStream_t hs);
#endif

extern "C" {

#ifdef ZL_W4M_PROBE
int zl_debug_set_probe_m(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(zl_probe_m), &p, sizeof(p)); }
#endif

int zl_w4m_layout(int64_t n, int64_t k, int64_t g, zl_w4_layout_t* out) {
    ZL_CHECK_ARG(out && n > 0 && k > 0 && g > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 8 == 0 && k % g == 0 && (g % 128 == 0), ZL_ESHAPE);  // group = multiple of the 128-k tile
    out->n = n; out->k = k; out->group_size = g;
    out->np = (n + 15) / 16 * 16;
    out->kp = (k + 127) / 128 * 128;
    out->q = out->kp / 128;
    out->c = 1;
    out->qw_bytes = (out->np / 16) * out->q * 1024;
    out->scales_bytes = (out->np / 16) * out->q * 64;   // the meta array (scale | zero per row and item)
    out->zeros_bytes = 0;
    return ZL_OK;
}

int zl_w4m_pack(const uint32_t* qweight_km, const uint8_t* qzeros_km, const uint16_t* scales_km, int64_t n, int64_t k,
                int64_t g, int row_interleave, uint32_t* qw, uint32_t* meta, zl_stream_t s) {
    ZL_CHECK_ARG(qweight_km && qzeros_km && scales_km && qw && meta, ZL_EINVAL);
    zl_w4_layout_t L;
    int st = zl_w4m_layout(n, k, g, &L);
    if (st) return st;
    ZL_CHECK_ARG(!row_interleave || n % 2 == 0, ZL_ESHAPE);
    const int64_t tiles = L.np / 16;
    hipLaunchKernelGGL(k_pack_m_qw, dim3(grid_for(tiles * L.q * 256)), dim3(256), 0, (hipStream_t)s, qweight_km, qw, n,
                       k / 8, tiles, L.q, row_interleave);
    hipLaunchKernelGGL(k_pack_m_meta, dim3(grid_for(tiles * L.q * 16)), dim3(256), 0, (hipStream_t)s, qzeros_km,
                       scales_km, meta, n, k / g, g / 128, tiles, L.q, row_interleave);
    return zl_launch_status();
}

int zl_w4m_unpack(const uint32_t* qw, const uint32_t* meta, int64_t n, int64_t k, int64_t g, int row_interleave,
                  uint32_t* qweight_km, uint8_t* qzeros_km, uint16_t* scales_km, zl_stream_t s) {
    ZL_CHECK_ARG(qweight_km && qzeros_km && scales_km && qw && meta, ZL_EINVAL);
    zl_w4_layout_t L;
    int st = zl_w4m_layout(n, k, g, &L);
    if (st) return st;
    ZL_CHECK_ARG(!row_interleave || n % 2 == 0, ZL_ESHAPE);
    const int64_t tiles = L.np / 16;
    hipLaunchKernelGGL(k_unpack_m_qw, dim3(grid_for(tiles * L.q * 256)), dim3(256), 0, (hipStream_t)s, qw, qweight_km, n,
                       k / 8, tiles, L.q, row_interleave);
    hipLaunchKernelGGL(k_unpack_m_meta, dim3(grid_for(tiles * L.q * 16)), dim3(256), 0, (hipStream_t)s, meta, qzeros_km,
                       scales_km, n, k / g, g / 128, tiles, L.q, row_interleave);
    return zl_launch_status();
}

int64_t zl_w4a16_scratch_bytes(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return ZL_EINVAL;
    const int64_t np = (n + 127) / 128 * 128;
    // fp32 split-K partials, the largest any launcher asks for: up to 32 splits only while the tiles do not fill the chip
    // (few rows: 32 x min(m, 32) x np; more rows: splits x tiles <= ~2 x CUs, i.e. splits x m x np <= 2 m np + 2 x 256 CUs
    // x the largest tile, 256 x 256) -- not 32 x m x np, which was 3.8 GB for a 1024-token gate|up
    const int64_t few = 32 * (m < 32 ? m : 32) * np;
    return ZL_SCRATCH_HEADER + (few + 2 * m * np + 2 * 256 * 256 * 256) * (int64_t)sizeof(float);
}

int zl_w4a16_gemm_mfma(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, const uint16_t* bias,
                       const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k, int64_t group_size,
                       const uint16_t* norm_weight, float norm_eps, int epilogue, zl_stream_t s) {
    return zl_w4a16_gemm_mfma_ex(x, ldx, qw, meta, bias, residual, y, m, n, k, group_size, norm_weight, norm_eps, epilogue, nullptr, s);
}

int zl_w4a16_gemm_mfma_ex(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, const uint16_t* bias,
                          const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k, int64_t group_size,
                          const uint16_t* norm_weight, float norm_eps, int epilogue, const zl_w4_opts_t* opts, zl_stream_t s) {
    static const zl_w4_opts_t kNoOpts = {};
    const zl_w4_opts_t& o = opts ? *opts : kNoOpts;
    ZL_CHECK_ARG(x && qw && meta && y && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(ldx >= k && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(!(epilogue & ZL_EPI_RESIDUAL) || residual, ZL_EINVAL);
    zl_w4_layout_t L;
    int st = zl_w4m_layout(n, k, group_size, &L);
    if (st) return st;
    const bool silu = epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32);
    ZL_CHECK_ARG(!silu || n % 2 == 0, ZL_ESHAPE);
    hipStream_t hs = (hipStream_t)s;
    // more than one 16-row pass: the M-tiled kernel (w4_gemm_tiled.hip; no fused norm prologue) reads the
    // weights once per 32-128 rows; with few rows it splits K over workgroups to fill the chip
    // (M = 32: 82 vs 113 us per Llama-3-8B layer for two passes of this kernel; M = 64: 109 vs 224)
    // 5..32 rows without a fused norm: the phase-pipelined streaming kernel (w4_phase.hip).  With more than 16
    // rows every workgroup pulls M x K activations through L2, so a long K (the down projection) stays on
    // the M-tiled kernel, whose 128-column workgroups share them.
    // 1..4 rows (1..2 with a long K): the integer-plane kernel (w4_i8p.hip), the batch-1 decode default
    // small_algo == 2: the same arithmetic on the loader / consumer engine (w4_engine.hip) where it applies
#ifdef ZL_EXPERIMENTAL
    if (o.small_algo == 2 && L.qw_bytes < ((int64_t)1 << 32)) {
        st = zl_w4a16_gemm_engine(x, ldx, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, residual, y, (int)m, (int)n,
                                  (int)k, (int)L.q, (int)(L.np / 16), epilogue, (int)(silu ? n / 2 : n), norm_weight, norm_eps,
                                  o.phase_rounds, hs);
        if (st != ZL_ESHAPE && st != ZL_ELIMIT) return st;
    }
#endif
    if ((o.small_algo == 0 || o.small_algo == 2) && zl_w4a16_i8p_covers(m, k) && L.qw_bytes < ((int64_t)1 << 32))
        return zl_w4a16_gemm_i8p(x, ldx, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, residual, y, (int)m, (int)n,
                                 (int)k, (int)L.q, (int)(L.np / 16), epilogue, (int)(silu ? n / 2 : n), norm_weight, norm_eps,
                                 o.phase_rounds, hs);
    // 9..32 rows without a fused norm (round 6): 128-column x K-slice tiles, activation fragments straight from global memory
    // (w4_slab.hip) -- a workgroup's activation bytes ~ its weight bytes instead of M x K per 16 R columns
    // ... and WITH a norm when the caller hands over the rows' statistics (zl_w4_opts_t::row_ss): the slab kernel's NORM instantiations
    if (o.slab >= 0 && (!norm_weight || o.row_ss) && m >= (o.slab_min_m > 0 ? o.slab_min_m : (o.small_algo == 1 ? 5 : 3)) && m <= 32 && k % 128 == 0 &&
        L.qw_bytes < ((int64_t)1 << 32)) {
        st = zl_w4a16_gemm_slab(x, ldx, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, residual, y, (int)m, (int)n,
                                (int)k, (int)L.q, (int)(L.np / 16), epilogue, (int)(silu ? n / 2 : n), norm_weight, norm_eps, &o, hs);
        if (st != ZL_ESHAPE) return st;
    }
    {
        const int ph_min_m = o.phase_min_m > 0 ? o.phase_min_m : 5, ph_max_m = o.phase_max_m > 0 ? o.phase_max_m : 32;
        const int ph_ksplit = o.phase_ksplit ? o.phase_ksplit : 2;   // long K, 13..32 rows: K split inside the phase kernel
        const bool have_scratch = o.scratch && o.scratch_bytes >= ZL_SCRATCH_HEADER + (int64_t)ph_ksplit * m * n * (int64_t)sizeof(float);
        const bool can_split = (ph_ksplit == 2 || ph_ksplit == 4) && !silu && !norm_weight && have_scratch;
        const int ph_maxk_32 = can_split ? (1 << 30) : 8192;
        const bool rows_5_32 = !norm_weight && m >= ph_min_m && m <= ph_max_m && m <= 32 && k <= (m <= 16 ? (1 << 30) : ph_maxk_32);
        const bool rows_1_4 = !o.phase_small_off && k <= 4096 && (norm_weight ? m <= 8 : (m <= 4 && m < ph_min_m));   // fused norm: <= 8 rows
        // fused norm with 9..32 rows: the phase kernel's DEFERRED norm (w4_phase.hip DN: T(x w) staged, rs applied to the fp32 totals)
        //   -- only on request (zl_w4_opts_t::defer_norm): it is NOT zl_rmsnorm + GEMM bit for bit, and parity is the first gate
        ZL_CHECK_ARG(!(norm_weight && m >= 9 && m <= 32) || o.defer_norm == 1, ZL_ESHAPE);
        const bool rows_9_32_dn = norm_weight && m >= 9 && m <= 32 && k % 128 == 0 && k <= (m <= 16 ? (1 << 30) : 8192);
        if ((rows_5_32 || rows_1_4 || rows_9_32_dn) && L.qw_bytes < ((int64_t)1 << 32))
            return zl_w4a16_gemm_phase(x, ldx, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, residual, y,
                                       (int)m, (int)n, (int)k, (int)L.q, (int)(L.np / 16), epilogue, (int)(silu ? n / 2 : n),
                                       norm_weight, norm_eps, &o, hs);
    }
    const int tiled_min_m = o.tiled_min_m > 0 ? o.tiled_min_m : 17;
    if (m >= tiled_min_m && !norm_weight && k % 128 == 0)
        return zl_w4a16_gemm_tiled_ex(x, ldx, qw, meta, bias, residual, y, m, n, k, group_size, epilogue, &o, s);

    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    const int64_t ld_out = silu ? n / 2 : n;
    for (int64_t m0 = 0; m0 < m; m0 += 16) {  // <= 16 activation rows per pass (the reference chunks by 16 too)
        const int mm = (int)(m - m0 < 16 ? m - m0 : 16);
        MfmaParams p;
        p.x = x + m0 * ldx; p.ldx = ldx;
        p.qw = reinterpret_cast<const uint4*>(qw);
        p.meta = meta;
        if (L.qw_bytes >= (int64_t)1 << 32) return ZL_ELIMIT;   // 32-bit buffer offsets
        p.qw_bytes = (uint32_t)L.qw_bytes;
        p.meta_bytes = (uint32_t)L.scales_bytes;
        p.bias = bias;
        p.residual = residual ? residual + m0 * ld_out : nullptr;
        p.y = y + m0 * ld_out;
        p.norm_w = norm_weight; p.norm_eps = norm_eps;
        p.m = mm; p.n = (int)n; p.k = (int)k; p.kp = (int)L.kp;
        p.groups = (int)L.q;
        p.tiles = (int)(L.np / 16);
        p.epi = epilogue; p.ld_out = (int)ld_out;
        p.lds_stride = p.kp + 8;
        const size_t x_bytes = (size_t)mm * p.lds_stride * 2;
        // workgroup shape: one 8-wave workgroup per CU.  Pick the k-split and the rounds per workgroup that
        // minimise the items the busiest SIMD has to stream: generations of workgroups x 2 waves per SIMD
        // x rounds x items per wave, plus a fixed charge per generation (x staging + launch ramp) and
        // per round (LDS parking).
        const size_t red_per_round = (size_t)kWaves * 16 * ((mm + 3) / 4) * 16;
        int best_ks = 1, best_rounds = 1;
        double best_cost = 1e30;
        for (int ks = 1; ks <= kWaves; ks *= 2) {
            if (ks > 1 && p.groups / ks < 2) break;
            const int ips = (p.groups + ks - 1) / ks, tpr = kWaves / ks;
            const int64_t rounds_total = (p.tiles + tpr - 1) / tpr;
            for (int rounds = 1; rounds <= kMaxRounds; ++rounds) {
                if (rounds > 1 && x_bytes + rounds * red_per_round + 1024 > 160 * 1024) break;
                const int64_t grid = (rounds_total + rounds - 1) / rounds;
                const int64_t gens = (grid + cus - 1) / cus;
                const double cost = (double)gens * (2.0 * rounds * ips + 8.0) + 0.25 * rounds;
                if (cost < best_cost - 1e-9) {
                    best_cost = cost; best_ks = ks; best_rounds = rounds;
                }
            }
        }
        // explicit overrides (micro-benchmark sweeps)
        if (o.mfma_ks > 0 && o.mfma_ks <= kWaves && (o.mfma_ks & (o.mfma_ks - 1)) == 0) best_ks = o.mfma_ks;
        if (o.mfma_rounds > 0 && o.mfma_rounds <= kMaxRounds) best_rounds = o.mfma_rounds;
        p.ks = best_ks;
        p.items_per_slice = (p.groups + best_ks - 1) / best_ks;
        p.rounds = best_rounds;
        const int tpr = kWaves / best_ks;
        const int64_t rounds_total = (p.tiles + tpr - 1) / tpr;
        const int grid = (int)((rounds_total + best_rounds - 1) / best_rounds);
        const size_t red_bytes = best_rounds * red_per_round + 8 * kWaves * 4 + 64;
        const bool x_in_lds = x_bytes + red_bytes <= 160 * 1024;
        if (!x_in_lds && norm_weight) return ZL_ELIMIT;  // fused norm needs the LDS staging
        // x chunks (16 B) per thread; <= 8 of them ride in registers across the ring prologue
        const int chunks = mm * (int)((p.kp + kThreads * 8 - 1) / (kThreads * 8));
        const int xmode = !x_in_lds ? -1 : (chunks <= 1 ? 1 : chunks <= 2 ? 2 : chunks <= 4 ? 4 : 0);
        const size_t lds = (x_in_lds ? x_bytes : 0) + red_bytes;
#define ZL_MFMA_LAUNCH(XM)                                                                                    \
    {                                                                                                         \
        if (lds > 64 * 1024) {                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_w4a16_mfma<XM>),              \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
            if (e != hipSuccess) return ZL_ELIMIT;                                                            \
        }                                                                                                     \
        hipLaunchKernelGGL((k_w4a16_mfma<XM>), dim3(grid), dim3(kThreads), lds, hs, p);                        \
    }
        switch (xmode) {
            case -1: ZL_MFMA_LAUNCH(-1) break;
            case 0: ZL_MFMA_LAUNCH(0) break;
            case 1: ZL_MFMA_LAUNCH(1) break;
            case 2: ZL_MFMA_LAUNCH(2) break;
            default: ZL_MFMA_LAUNCH(4) break;
        }
#undef ZL_MFMA_LAUNCH
        st = zl_launch_status();
        if (st) return st;
    }
    return ZL_OK;
}

int zl_w4a16_qkv_rope_scatter(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta,
                              const uint16_t* bias, const uint16_t* norm_weight, float norm_eps, const float* cosv,
                              const float* sinv, const int32_t* placement, const int32_t* buf_lens,
                              uint16_t* const* k_bufs, uint16_t* const* v_bufs, uint16_t* q_out, int64_t m, int64_t h,
                              int64_t hkv, int64_t d, int64_t k, int64_t group_size, int bshd, zl_stream_t s) {
    return zl_w4a16_qkv_rope_scatter_ex(x, ldx, qw, meta, bias, norm_weight, norm_eps, cosv, sinv, placement, buf_lens, k_bufs,
                                        v_bufs, q_out, m, h, hkv, d, k, group_size, bshd, nullptr, s);
}

int zl_w4a16_qkv_rope_scatter_ex(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta,
                                 const uint16_t* bias, const uint16_t* norm_weight, float norm_eps, const float* cosv,
                                 const float* sinv, const int32_t* placement, const int32_t* buf_lens,
                                 uint16_t* const* k_bufs, uint16_t* const* v_bufs, uint16_t* q_out, int64_t m, int64_t h,
                                 int64_t hkv, int64_t d, int64_t k, int64_t group_size, int bshd, const zl_w4_opts_t* opts,
                                 zl_stream_t s) {
    ZL_CHECK_ARG(x && qw && meta && cosv && sinv && placement && buf_lens && k_bufs && v_bufs && q_out, ZL_EINVAL);
    ZL_CHECK_ARG(m > 0 && h > 0 && hkv > 0 && d > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(ldx >= k && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, ZL_ESHAPE);
    const int64_t n = (h + 2 * hkv) * d;
    zl_w4_layout_t L;
    int st = zl_w4m_layout(n, k, group_size, &L);
    if (st) return st;
    // what the phase-pipelined kernel covers (zl_w4a16_gemm_mfma's own dispatch rules); callers fall back to
    // zl_w4a16_gemm_mfma + zl_rope_scatter_decode outside of it
    ZL_CHECK_ARG(m <= 32 && d % 32 == 0 && h % 1 == 0 && L.np == n, ZL_ESHAPE);
    ZL_CHECK_ARG(m <= 16 || k <= 8192, ZL_ESHAPE);
    ZL_CHECK_ARG(L.qw_bytes < ((int64_t)1 << 32), ZL_ELIMIT);
    const int small_algo = opts ? opts->small_algo : 0;
    if (opts && opts->slab >= 0 && (!norm_weight || opts->row_ss) && m >= (opts->slab_min_m > 0 ? opts->slab_min_m : 5) && k % 128 == 0) {
        st = zl_w4a16_gemm_slab_rope(x, ldx, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, (int)m, (int)n, (int)k,
                                     (int)L.q, (int)(L.np / 16), cosv, sinv, placement, buf_lens, k_bufs, v_bufs, q_out, (int)h,
                                     (int)hkv, (int)d, bshd, norm_weight, norm_eps, opts, (hipStream_t)s);
        if (st != ZL_ESHAPE) return st;
    }
    ZL_CHECK_ARG(!norm_weight || k <= 4096 || m > 8, ZL_ESHAPE);     // <= 8 rows: register-resident staging; 9..32: deferred norm
    ZL_CHECK_ARG(!(norm_weight && m >= 9) || (opts && opts->defer_norm == 1), ZL_ESHAPE);   // the deferred norm: on request only
#ifdef ZL_EXPERIMENTAL
    if (small_algo == 2) {
        st = zl_w4a16_gemm_engine_rope(x, ldx, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, (int)m, (int)n, (int)k,
                                       (int)L.q, (int)(L.np / 16), norm_weight, norm_eps, cosv, sinv, placement, buf_lens, k_bufs,
                                       v_bufs, q_out, (int)h, (int)hkv, (int)d, bshd, (hipStream_t)s);
        if (st != ZL_ESHAPE && st != ZL_ELIMIT) return st;
    }
#endif
    if ((small_algo == 0 || small_algo == 2) && zl_w4a16_i8p_covers(m, k))
        return zl_w4a16_gemm_i8p_rope(x, ldx, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, (int)m, (int)n,
                                      (int)k, (int)L.q, (int)(L.np / 16), norm_weight, norm_eps, cosv, sinv, placement,
                                      buf_lens, k_bufs, v_bufs, q_out, (int)h, (int)hkv, (int)d, bshd, (hipStream_t)s);
    return zl_w4a16_gemm_phase_rope(x, ldx, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, (int)m, (int)n,
                                    (int)k, (int)L.q, (int)(L.np / 16), norm_weight, norm_eps, cosv, sinv, placement,
                                    buf_lens, k_bufs, v_bufs, q_out, (int)h, (int)hkv, (int)d, bshd, (hipStream_t)s);
}

#ifdef ZL_EXPERIMENTAL
int64_t zl_w4a16_planes_bytes(int64_t m, int64_t k) { return zl_w4_planes_bytes_(m, k); }

int zl_w4a16_planes(const uint16_t* x, int64_t ldx, int64_t m, int64_t k, const uint16_t* norm_weight, float norm_eps, void* planes,
                    zl_stream_t s) {
    ZL_CHECK_ARG(x && planes && m > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(ldx >= k && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(zl_w4_planes_bytes_(m, k) > 0, ZL_ESHAPE);
    return zl_w4_planes_launch(x, ldx, (int)m, (int)k, norm_weight, norm_eps, planes, (hipStream_t)s);
}

int zl_w4a16_gemm_planes(const void* planes, const uint32_t* qw, const uint32_t* meta, const uint16_t* bias, const uint16_t* residual,
                         uint16_t* y, int64_t m, int64_t n, int64_t k, int64_t group_size, int epilogue, const zl_w4_opts_t* opts,
                         zl_stream_t s) {
    ZL_CHECK_ARG(planes && qw && meta && y && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(!(epilogue & ZL_EPI_RESIDUAL) || residual, ZL_EINVAL);
    ZL_CHECK_ARG(zl_w4_planes_bytes_(m, k) > 0 && ((uintptr_t)planes & 15) == 0, ZL_ESHAPE);
    zl_w4_layout_t L;
    int st = zl_w4m_layout(n, k, group_size, &L);
    if (st) return st;
    const bool silu = epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32);
    ZL_CHECK_ARG(!silu || n % 2 == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(L.qw_bytes < ((int64_t)1 << 32), ZL_ELIMIT);
    return zl_w4a16_gemm_phase_planes(planes, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, residual, y, (int)m, (int)n,
                                      (int)k, (int)L.q, (int)(L.np / 16), epilogue, (int)(silu ? n / 2 : n), opts, (hipStream_t)s);
}

int zl_w4a16_qkv_rope_scatter_planes(const void* planes, const uint32_t* qw, const uint32_t* meta, const uint16_t* bias, const float* cosv,
                                     const float* sinv, const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                                     uint16_t* const* v_bufs, uint16_t* q_out, int64_t m, int64_t h, int64_t hkv, int64_t d, int64_t k,
                                     int64_t group_size, int bshd, zl_stream_t s) {
    ZL_CHECK_ARG(planes && qw && meta && cosv && sinv && placement && buf_lens && k_bufs && v_bufs && q_out, ZL_EINVAL);
    ZL_CHECK_ARG(m > 0 && h > 0 && hkv > 0 && d > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(zl_w4_planes_bytes_(m, k) > 0 && ((uintptr_t)planes & 15) == 0, ZL_ESHAPE);
    const int64_t n = (h + 2 * hkv) * d;
    zl_w4_layout_t L;
    int st = zl_w4m_layout(n, k, group_size, &L);
    if (st) return st;
    ZL_CHECK_ARG(d % 32 == 0 && L.np == n, ZL_ESHAPE);
    ZL_CHECK_ARG(L.qw_bytes < ((int64_t)1 << 32), ZL_ELIMIT);
    return zl_w4a16_gemm_phase_planes_rope(planes, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, (int)m, (int)n, (int)k,
                                           (int)L.q, (int)(L.np / 16), cosv, sinv, placement, buf_lens, k_bufs, v_bufs, q_out, (int)h,
                                           (int)hkv, (int)d, bshd, (hipStream_t)s);
}

#endif  // ZL_EXPERIMENTAL (digit-plane entry points)

int zl_w4a16_gemm_attn_merge(const void* attn_workspace, const int32_t* buf_lens, const int32_t* valid_lens,
                             int64_t split_len, int64_t max_splits, const uint32_t* qw, const uint32_t* meta,
                             const uint16_t* bias, const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k,
                             int64_t group_size, int epilogue, zl_stream_t s) {
    ZL_CHECK_ARG(attn_workspace && buf_lens && valid_lens && qw && meta && y, ZL_EINVAL);
    ZL_CHECK_ARG(m > 0 && n > 0 && k > 0 && split_len > 0 && max_splits > 0, ZL_EINVAL);
    ZL_CHECK_ARG(!(epilogue & ZL_EPI_BIAS) || bias, ZL_EINVAL);
    ZL_CHECK_ARG(!(epilogue & (ZL_EPI_RESIDUAL | ZL_EPI_ADD_C)) || residual, ZL_EINVAL);
    // what the merging prologue covers; callers fall back to zl_decode_attn + zl_w4a16_gemm_mfma outside of it
    ZL_CHECK_ARG(m <= 4 && k <= 4096 && k % 128 == 0 && max_splits <= 16, ZL_ESHAPE);
    ZL_CHECK_ARG(!(epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)), ZL_ESHAPE);
    zl_w4_layout_t L;
    int st = zl_w4m_layout(n, k, group_size, &L);
    if (st) return st;
    ZL_CHECK_ARG(L.qw_bytes < ((int64_t)1 << 32), ZL_ELIMIT);
    return zl_w4a16_gemm_phase_merge(reinterpret_cast<const float*>(attn_workspace), buf_lens, valid_lens, (int)split_len,
                                     (int)max_splits, qw, meta, (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, residual,
                                     y, (int)m, (int)n, (int)k, (int)L.q, (int)(L.np / 16), epilogue, (hipStream_t)s);
}

int zl_w4a16_gemm_attn_merge_h(const void* attn_workspace, const int32_t* buf_lens, const int32_t* valid_lens,
                               int64_t split_len, int64_t max_splits, const uint32_t* qw, const uint32_t* meta,
                               const uint16_t* bias, const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k,
                               int64_t group_size, int epilogue, zl_stream_t s) {
    return zl_w4a16_gemm_attn_merge_h_ex(attn_workspace, buf_lens, valid_lens, split_len, max_splits, qw, meta, bias, residual, y, m,
                                         n, k, group_size, epilogue, nullptr, s);
}

int zl_w4a16_gemm_attn_merge_h_ex(const void* attn_workspace, const int32_t* buf_lens, const int32_t* valid_lens,
                                  int64_t split_len, int64_t max_splits, const uint32_t* qw, const uint32_t* meta,
                                  const uint16_t* bias, const uint16_t* residual, uint16_t* y, int64_t m, int64_t n, int64_t k,
                                  int64_t group_size, int epilogue, const zl_w4_opts_t* opts, zl_stream_t s) {
    ZL_CHECK_ARG(attn_workspace && buf_lens && valid_lens && qw && meta && y, ZL_EINVAL);
    ZL_CHECK_ARG(m > 0 && n > 0 && k > 0 && split_len > 0 && max_splits > 0, ZL_EINVAL);
    ZL_CHECK_ARG(!(epilogue & ZL_EPI_BIAS) || bias, ZL_EINVAL);
    ZL_CHECK_ARG(!(epilogue & (ZL_EPI_RESIDUAL | ZL_EPI_ADD_C)) || residual, ZL_EINVAL);
    ZL_CHECK_ARG(m <= 4 && k <= 4096 && k % 128 == 0 && max_splits <= 16, ZL_ESHAPE);
    ZL_CHECK_ARG(!(epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)), ZL_ESHAPE);
    zl_w4_layout_t L;
    int st = zl_w4m_layout(n, k, group_size, &L);
    if (st) return st;
    ZL_CHECK_ARG(L.qw_bytes < ((int64_t)1 << 32), ZL_ELIMIT);
#ifdef ZL_EXPERIMENTAL
    if (opts && opts->small_algo == 2) {
        st = zl_w4a16_gemm_engine_merge(attn_workspace, buf_lens, valid_lens, (int)split_len, (int)max_splits, qw, meta,
                                        (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, residual, y, (int)m, (int)n, (int)k,
                                        (int)L.q, (int)(L.np / 16), epilogue, (hipStream_t)s);
        if (st != ZL_ESHAPE && st != ZL_ELIMIT) return st;
    }
#endif
    return zl_w4a16_gemm_i8p_merge(attn_workspace, buf_lens, valid_lens, (int)split_len, (int)max_splits, qw, meta,
                                   (uint32_t)L.qw_bytes, (uint32_t)L.scales_bytes, bias, residual, y, (int)m, (int)n, (int)k,
                                   (int)L.q, (int)(L.np / 16), epilogue, (hipStream_t)s);
}

#ifdef ZL_EXPERIMENTAL
int zl_w4a16_attn_out_gate_up(const void* attn_workspace, const int32_t* buf_lens, const int32_t* valid_lens, int64_t split_len,
                              int64_t max_splits, const uint32_t* qw_o, const uint32_t* meta_o, const uint16_t* bias_o,
                              uint16_t* hidden, const uint32_t* qw_ff, const uint32_t* meta_ff, const uint16_t* bias_ff,
                              const uint16_t* norm_weight, float norm_eps, uint16_t* act, int64_t m, int64_t dim_model,
                              int64_t dim_attn, int64_t n_ff, int64_t group_size, void* granules, const uint32_t* epoch,
                              uint32_t epoch_add, uint32_t* err, zl_stream_t s) {
    ZL_CHECK_ARG(attn_workspace && buf_lens && valid_lens && qw_o && meta_o && hidden && qw_ff && meta_ff && norm_weight && act, ZL_EINVAL);
    ZL_CHECK_ARG(granules && epoch && m > 0 && dim_model > 0 && dim_attn > 0 && n_ff > 0 && split_len > 0 && max_splits > 0, ZL_EINVAL);
    ZL_CHECK_ARG(((uintptr_t)granules & 7) == 0 && n_ff % 2 == 0, ZL_ESHAPE);
    zl_w4_layout_t L1, L2;
    int st = zl_w4m_layout(dim_model, dim_attn, group_size, &L1);
    if (st) return st;
    st = zl_w4m_layout(n_ff, dim_model, group_size, &L2);
    if (st) return st;
    ZL_CHECK_ARG(L1.np == dim_model && L2.np == n_ff, ZL_ESHAPE);
    ZL_CHECK_ARG(L1.qw_bytes < ((int64_t)1 << 32) && L2.qw_bytes < ((int64_t)1 << 32), ZL_ELIMIT);
    return zl_w4_engine_o_gateup_launch(attn_workspace, buf_lens, valid_lens, (int)split_len, (int)max_splits, qw_o, meta_o,
                                        (uint32_t)L1.qw_bytes, (uint32_t)L1.scales_bytes, bias_o, hidden, (int)m, (int)dim_model,
                                        (int)dim_attn, (int)L1.q, (int)(L1.np / 16), qw_ff, meta_ff, (uint32_t)L2.qw_bytes,
                                        (uint32_t)L2.scales_bytes, bias_ff, norm_weight, norm_eps, act, (int)n_ff, (int)L2.q,
                                        (int)(L2.np / 16), ZL_EPI_SILU_MUL | (bias_ff ? ZL_EPI_BIAS : 0), granules, epoch, epoch_add,
                                        err, (hipStream_t)s);
}

int zl_engine_epoch_advance(uint32_t* epoch, uint32_t by, zl_stream_t s) {
    ZL_CHECK_ARG(epoch && by > 0, ZL_EINVAL);
    return zl_engine_epoch_advance_launch(epoch, by, (hipStream_t)s);
}
#endif  // ZL_EXPERIMENTAL (fused engine launch)

}  // extern "C"

// w4_i8p.hip -- W4A16 GEMV for 1..4 activation rows: the batch-1 decode hot path (gptq_gemm_k_major's M <= 4 regime,
// src/nn/quant/gptq/q_gemm_k_major.cu:957-1116 / KERNEL_gemm_warp_reduce :127-237) on the INTEGER matrix cores.
//
// Why another kernel.  k_w4a16_phase spends 53 VALU issue slots per 1 KiB weight item (exact fp16 (q - z) of 2048 nibbles,
// 4 fp16 MFMAs): a SIMD dequantises ~8 items/us while HBM delivers ~6 per SIMD, and nothing is consumed before the
// activations are staged -- so the short projections of a decode layer (32..64 KiB per CU, everything in flight from the
// start) run their whole compute AFTER the bytes have landed (profiles/r02_phase_timeline_in_step.txt).  Here:
//   * the nibbles are only EXPANDED to bytes (w & 0x0f0f0f0f, (w >> 4) & 0x0f0f0f0f: 3 VALU per 8 weights) and contracted
//     by v_mfma_i32_16x16x64_i8 (2 per item instead of 4 fp16 MFMAs);
//   * the activations become integers: per 128-k group a block-floating-point scale 2^e (e from the group's largest
//     magnitude), X = rint(x * 2^(36 - Ef)) (|X| < 2^22, exact for every fp16 value within 2^-12 of the group maximum,
//     2^-22 of it otherwise), split into three balanced signed-byte digits X = 65536 b2 + 256 b1 + b0 that ride as three
//     ROWS of the MFMA's A operand (a batch row = rows 4 r .. 4 r + 2, so up to 4 batch rows share every instruction);
//   * per item and lane: D = 65536 D0 + 256 D1 + D2 = sum_k X_k q_k EXACTLY (int32), then in fp32
//         acc += s * (xscale * D - z * (xscale * sum_k X_k))
//     with the group constants (xscale, 65536 xscale, xscale * SX, 1024 * that) read from LDS once per group: 7 VALU.
//     The zero point costs one fma per item instead of a subtraction per weight.
//   21 issue slots per item instead of 53, no fp16 rounding anywhere inside a group: the result is the exact product of the
//   fp16 activations with (q - z) * s up to fp32 accumulation over the groups -- closer to the exact value than both the
//   reference's fp16-partial-sum kernel and its dequant + GEMM branch (parity: tests/test_gpu_w4.py, bar = output rounding).
//   * the prologue loads the activations, WAITS for them, and only then issues the weight ring: with the ring issued first
//     the (older!) activation loads come back behind ~60 KiB of weights per CU, 2.6 us after entry instead of 0.7
//     (tools/ubench/bcast_probe.hip, profiles/r03_bcast_probe.txt); the RMSNorm, the integer conversion and the LDS
//     transposition then run while the first weights are in flight;
//   * a wave keeps ONE group's A fragments in registers for all R row tiles of the workgroup (wave w owns the groups
//     g = w mod 8): no LDS reads per item, no phase barriers; the only barriers are the two of the prologue and the one in
//     front of the cross-wave reduction.
// Same ZLW4M operands, same epilogues and the same fused front ends (RMSNorm prologue, neox rotary + KV scatter) as
// k_w4a16_phase; the split-merge prologue is gone (the merged attention rows come from the attention kernel).
#include "zl_common.h"
#include "w4_i8p_common.h"

namespace {

constexpr int kT = 512, kW = 8;
#ifndef ZL_I8P_RING
#define ZL_I8P_RING 8            // 1 KiB items a wave keeps in flight (rounded down to whole groups of R tiles)
#endif
// ZL_I8P_EARLY: ring items a wave requests BEHIND its activation loads but BEFORE waiting for them (round 5 experiment).  The
// activations are older, so `s_waitcnt vmcnt(2 E)` still hands them over first; the question is whether E KiB per wave of weights
// (8 E KiB per CU instead of the 64 KiB of a full ring, which delayed every workgroup's activations by ~2 us) buys back part of the
// first-byte latency the x-first order spends: profiles/r05_i8p_early.txt.
#ifndef ZL_I8P_EARLY
#define ZL_I8P_EARLY 0
#endif

// ---- optional timeline probe (build with -DZL_I8P_PROBE; tools/ubench/probe_i8p.py): wall-clock stamps (100 MHz) per wave,
//      [workgroup][wave][8]; every launch overwrites them, so after a replayed chain they describe its last launch
#ifdef ZL_I8P_PROBE
__device__ unsigned long long* zl_probe_i8 = nullptr;
#define ZL_IPROBE_INIT()                                                                               \
    unsigned long long* pp_ = nullptr;                                                                 \
    if (zl_probe_i8 && (threadIdx.x & 63) == 0 && blockIdx.x < 2048)                                   \
        pp_ = zl_probe_i8 + ((size_t)blockIdx.x * kW + (threadIdx.x >> 6)) * 8;
#define ZL_IPROBE(slot)                                                                                \
    do {                                                                                               \
        if (pp_) pp_[slot] = wall_clock64();                                                           \
    } while (0)
#else
#define ZL_IPROBE_INIT() do {} while (0)
#define ZL_IPROBE(slot) do {} while (0)
#endif

// R: row tiles per workgroup.  LONGK: more than four groups per wave (K > 4096: rows <= 2, K <= 16384).  ROPE (R = 2): the
// two tiles are a column block and its rotation partners (w4_phase.hip).  NORM: fused RMSNorm prologue.
// MERGE (K = heads x 128 <= 4096, <= 16 splits): slot `row` of a lane = octet l & 15 of head w + 8 (l >> 4) of task `row`,
// merged from the attention's split partials (k_decode_attn_combine's weights; the partials are normalised fp16 rows, so
// out = sum_s l_s e^(m_s - m) O_s / sum_s l_s e^(m_s - m)): a wave reads only the heads of ITS groups, the workgroup as a whole
// every partial once.
template <int R, bool LONGK, bool ROPE, bool NORM, bool MERGE = false>
__global__ __launch_bounds__(kT, 2) void k_w4a16_i8p(const I8Params p) {
    static_assert(!ROPE || R == 2, "fused rotary: a tile and its partner tile");
    static_assert(!MERGE || (!LONGK && !ROPE && !NORM), "split merge: one column block, plain prologue");
    constexpr int XD = R >= ZL_I8P_RING ? 1 : (ZL_I8P_RING / R > 0 ? ZL_I8P_RING / R : 1);   // groups the ring runs ahead (8 KiB per wave in flight: with 16
    constexpr int D = R * XD;                                  // the issue itself stalls for microseconds)
    constexpr int NS = LONGK ? 8 : 4;                          // activation octet slots per thread
    constexpr int NR = LONGK ? 2 : 4;                          // rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ZL_IPROBE_INIT();
    ZL_IPROBE(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int M = p.m, K = p.k, groups = p.groups;
    const int rec = 64 * M;                                    // bytes per (group, mfma, kq) record: 4 M slots of 16 bytes
    const int Gw = (groups + kW - 1) / kW;                     // groups per wave: wave w owns g = w + 8 gi
    // LDS: wave-private regions -- nothing below needs a workgroup barrier before the final reduction (a wave's DS
    // instructions execute in order), except the eight partial sums of the fused RMSNorm
    unsigned char* planes = smem + (size_t)wave * Gw * 8 * rec;                       // [gi][mfma][kq][4 M slots][16]
    float* consts = reinterpret_cast<float*>(smem + (size_t)kW * Gw * 8 * rec) + (size_t)wave * Gw * 16;   // [gi][row 0..3][4]
    float* red = reinterpret_cast<float*>(smem + (size_t)kW * Gw * 8 * rec) + (size_t)kW * Gw * 16;        // [R][8 waves][64]
    float* scratch = red + R * kW * 64;                                               // [4 rows][8 waves]

    const int tile0 = ROPE ? (blockIdx.x / p.pair_stride) * 2 * p.pair_stride + blockIdx.x % p.pair_stride : blockIdx.x * R;
    const int tile_stride = ROPE ? p.pair_stride : 1;
    const int my_groups = wave < groups ? (groups - wave + kW - 1) / kW : 0;          // groups this wave owns

    // ---- weight ring state (declared here: ZL_I8P_EARLY items go out between the activation loads and their wait).
    //      Wave w streams the items (tile0 + r, g = w + 8 gi), gi-major.
    constexpr int E = ZL_I8P_EARLY < D ? ZL_I8P_EARLY : D;
    uint4 wq[D];
    uint32_t mt[D];
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.qw), 0, p.qw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.meta), 0, p.meta_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.qw), 0, 0, 0x00020000);
    const uint32_t q_off = (uint32_t)lane * 16u, m_off = (uint32_t)(lane & 15) * 4u;
    auto issue = [&](int slot, int gi, int r) {        // slot, r: static
        const int g = wave + kW * gi;
        const bool ok = g < groups;                    // wave-uniform; tiles past the end fall outside the descriptor
        const uint32_t it = (uint32_t)(tile0 + r * tile_stride) * (uint32_t)groups + (uint32_t)g;
        wq[slot] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ok ? rq : rnull, q_off, it * 1024u, 2 /* nt */));
        mt[slot] = __builtin_amdgcn_raw_buffer_load_b32(ok ? rm : rnull, m_off, it * 64u, 2);
    };
    auto issue_early = [&]() {
        if constexpr (E > 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < E; ++q) issue(q, q / R, q % R);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- activations first.  A wave loads exactly the k ranges of ITS groups: slot s = (row, column block c), lane l holds
    //      octet l & 15 of group g = w + 8 (4 c + (l >> 4)) -- one 16-byte load per lane covers four groups of a row
    uint4 xr[NS], nw[LONGK ? 4 : 1];
    const int lgi = lane >> 4, uo = lane & 15;
#pragma unroll
    for (int c = 0; c < (LONGK ? 4 : 1); ++c) {
        nw[c] = make_uint4(0, 0, 0, 0);
        const int g = wave + kW * (4 * c + lgi);
        if (NORM && 4 * c < Gw) nw[c] = *reinterpret_cast<const uint4*>(p.norm_w + (g < groups ? g * 128 + uo * 8 : 0));
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int row = LONGK ? (s & 1) : s, c = LONGK ? (s >> 1) : 0;
        const int g = wave + kW * (4 * c + lgi);
        xr[s] = make_uint4(0, 0, 0, 0);
        if constexpr (MERGE) {
            if (row < M) {
                constexpr int kS = 16;
                const int head = g < groups ? g : 0;
                // records of (row, head) are consecutive: one descriptor per array, the record index in the scalar
                // offset's place and u in the immediate -- no 64-bit address arithmetic per load, and a record past the
                // end of the arrays (u >= max_splits on the last head) reads as zero instead of faulting
                const uint32_t n_rec = (uint32_t)M * (uint32_t)groups * (uint32_t)p.mg_max_splits;
                const __amdgpu_buffer_rsrc_t rpart = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.mg_part), 0, n_rec * 256u, 0x00020000);
                const __amdgpu_buffer_rsrc_t rstat = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mg_stat), 0, n_rec * 8u, 0x00020000);
                const uint32_t rec0 = ((uint32_t)row * (uint32_t)groups + (uint32_t)head) * (uint32_t)p.mg_max_splits;
                const uint32_t po = rec0 * 256u + (uint32_t)uo * 16u, so = rec0 * 8u;
                uint4 pv[kS];
                float2 st[kS];
#pragma unroll
                for (int u = 0; u < kS; ++u) {           // every record the launch geometry allows; dead ones are masked below
                    // (unconditional loads: behind a uniform branch each load is waited for at the join -- 16 dependent
                    //  round trips, 8.0 instead of 6.7 us in the step)
                    //  a record index past the launch's split count re-reads record 0 -- a line the wave already has -- rather
                    //  than the next head's; scalar select, scalar offset operand)
                    const uint32_t uc = u < p.mg_max_splits ? (uint32_t)u : 0u;
                    pv[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rpart, po, uc * 256u, 0));
                    st[u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rstat, so, uc * 8u, 0));
                }
                if (s == 0) issue_early();               // behind row 0's records (older: they are handed over first)
                const int elen = min(p.buf_lens[row], p.mg_valid_lens[row]);
                const int ns = min((elen + p.mg_split_len - 1) / p.mg_split_len, kS);
                float mn = -1e20f;
#pragma unroll
                for (int u = 0; u < kS; ++u) mn = fmaxf(mn, u < ns ? st[u].x : -1e20f);
                float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, z = 0.f;
#pragma unroll
                for (int u = 0; u < kS; ++u) {
                    if (u < ns) {                        // workgroup-uniform
                        const float f = st[u].y * __expf(st[u].x - mn);
                        const uint32_t w4[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const h16x2 hh = __builtin_bit_cast(h16x2, w4[e]);
                            a[2 * e] = __builtin_fmaf((float)hh.x, f, a[2 * e]);
                            a[2 * e + 1] = __builtin_fmaf((float)hh.y, f, a[2 * e + 1]);
                        }
                        z += f;
                    }
                }
                // (one reciprocal instead of eight divisions: ~70 VALU of a front end every workgroup repeats; the product
                //  differs from the quotient by at most an fp32 ulp before the rounding to fp16)
                const float zi = 1.0f / (z + 1e-20f);
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h16x2 hh;
                    hh.x = zl_f32_to_f16(a[2 * e] * zi);
                    hh.y = zl_f32_to_f16(a[2 * e + 1] * zi);
                    o[e] = __builtin_bit_cast(uint32_t, hh);
                }
                if (g < groups) xr[s] = make_uint4(o[0], o[1], o[2], o[3]);
            }
        } else
        if (row < M && 4 * c < Gw) {                    // workgroup-uniform: no load instructions for slots that do not exist
            xr[s] = *reinterpret_cast<const uint4*>(p.x + (g < groups ? (size_t)row * p.ldx + g * 128 + uo * 8 : 0));
            if (g >= groups) xr[s] = make_uint4(0, 0, 0, 0);
        }
    }
    // ROPE: what the epilogue of this thread needs from memory (rotation table entries, the task's slot and buffer
    // pointers) is requested now, next to the activations -- in the epilogue these were three dependent round trips
    float rp_c0 = 0.f, rp_s0 = 0.f, rp_c1 = 0.f, rp_s1 = 0.f;
    int rp_place = -1, rp_blen = 0;
    uint16_t* rp_kv = nullptr;
    if constexpr (ROPE) {
        if ((int)threadIdx.x < 16 * M) {
            const int m = threadIdx.x >> 4, n0 = tile0 * 16 + (threadIdx.x & 15);
            const int head = n0 / p.d, dcol = n0 % p.d, half = p.d / 2;
            if (head < p.h + p.hkv) {
                rp_c0 = p.cosv[(size_t)m * p.d + dcol]; rp_s0 = p.sinv[(size_t)m * p.d + dcol];
                rp_c1 = p.cosv[(size_t)m * p.d + dcol + half]; rp_s1 = p.sinv[(size_t)m * p.d + dcol + half];
            }
            if (head >= p.h) {
                rp_place = p.placement[m];
                rp_blen = p.buf_lens[m];
                rp_kv = head < p.h + p.hkv ? p.k_bufs[m] : p.v_bufs[m];
            }
        }
    }
    // the other epilogues: every thread owns at most one output (R * 16 * M <= 512); its bias / residual / previous-output
    // operands are requested now as well (one more round trip at the very end of the kernel otherwise)
    const bool silu = (p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
    float ep_b0 = 0.f, ep_b1 = 0.f, ep_res = 0.f, ep_prev = 0.f;
    int ep_r = 0, ep_m = 0, ep_nl = 0, ep_col = -1;       // tile slot, row, column inside the tile, output column (-1: none)
    if constexpr (!ROPE) {
        const int per_tile = (silu ? 8 : 16) * M;
        if ((int)threadIdx.x < R * per_tile) {
            ep_r = threadIdx.x / per_tile;
            const int rem = threadIdx.x % per_tile, tile = tile0 + ep_r;
            if (!silu) {
                ep_m = rem >> 4; ep_nl = rem & 15;
                const int row = tile * 16 + ep_nl;
                if (tile < p.tiles && row < p.n) {
                    ep_col = row;
                    if ((p.epi & ZL_EPI_BIAS) && p.bias) ep_b0 = (float)__builtin_bit_cast(_Float16, p.bias[row]);
                    if (p.epi & ZL_EPI_ADD_C) ep_prev = (float)__builtin_bit_cast(_Float16, p.y[(size_t)ep_m * p.ld_out + row]);
                    if (p.epi & ZL_EPI_RESIDUAL) ep_res = (float)__builtin_bit_cast(_Float16, p.residual[(size_t)ep_m * p.ld_out + row]);
                }
            } else {
                ep_m = rem >> 3; ep_nl = rem & 7;
                const int pr = tile * 8 + ep_nl;
                if (tile < p.tiles && 2 * pr + 1 < p.n) {
                    ep_col = pr;
                    if ((p.epi & ZL_EPI_BIAS) && p.bias) {
                        ep_b0 = (float)__builtin_bit_cast(_Float16, p.bias[2 * pr]);
                        ep_b1 = (float)__builtin_bit_cast(_Float16, p.bias[2 * pr + 1]);
                    }
                }
            }
        }
    }
    if constexpr (!MERGE) issue_early();
    __builtin_amdgcn_sched_barrier(0);
    // the activations have landed BEFORE the first weight is requested: issued behind the ring they come back 2 us later
    // (bcast_probe.hip).  The wait is the BUILTIN, not inline asm: the compiler's waitcnt pass models an S_WAITCNT it can see
    // and stops treating the registers above as pending.  With an opaque asm it kept them pending, lost the count of later
    // loads at the predicated ring issues and put its own `s_waitcnt vmcnt(0)` in front of every conversion slot -- AFTER
    // ring items had been requested, so slot s waited for the weights issued during slot s - 1 (a first-byte latency per
    // slot: the long-K kernel finished staging at 4.2 us)
    static_assert(2 * E < 16, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * E));      // vmcnt(2 E): everything but the E early items (two loads each), expcnt / lgkmcnt untouched
    __builtin_amdgcn_sched_barrier(0);
    ZL_IPROBE(1);

    // ---- weight ring: wave w streams the items (tile0 + r, g = w + 8 gi), gi-major.  The D prologue items are issued
    //      IN BETWEEN the stages of the activation conversion below: a wave that issues them back to back sits in VMEM issue
    //      for 0.6-1.5 us (a CU takes ~10 B/clk of misses whatever is queued) with its VALU work stuck behind
#define ZL_ISSUE_RANGE(LO, HI)                                     \
    {                                                              \
        __builtin_amdgcn_sched_barrier(0);                         \
        _Pragma("unroll") for (int q = ((LO) > E ? (LO) : E); q < (HI); ++q) issue(q, q / R, q % R); \
        __builtin_amdgcn_sched_barrier(0);                         \
    }
    // prologue items [0, Q0) go ahead of the norm's barrier, the rest between the four stages of the first slot's conversion
    // prologue items [0, Q0) go ahead of the norm's barrier, the rest between the conversion stages of row 0's first column
    // block (LONGK: first two blocks, which always exist): stage position pos of NP gets items [QLO(pos), QLO(pos + 1))
    constexpr int Q0 = NORM ? D / 2 : 0, NP = LONGK ? 8 : 4;
#define ZL_QLO(pos) (Q0 + ((D - Q0) * (pos) + NP - 1) / NP)

    // ---- fused RMSNorm (LayerNorm::forward, src/nn/layernorm/layernorm.cu:10-42): rs per row.  One workgroup barrier, with
    //      half of the ring issued in front of it (the rest would make the waves wait for each other's VMEM issue)
    float rs[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) rs[r] = 1.f;
    if constexpr (NORM) {
        float part[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) part[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int row = LONGK ? (s & 1) : s;
            const uint32_t u[4] = {xr[s].x, xr[s].y, xr[s].z, xr[s].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const h16x2 hh = __builtin_bit_cast(h16x2, u[e]);
                part[row] = __builtin_fmaf((float)hh.x, (float)hh.x, part[row]);
                part[row] = __builtin_fmaf((float)hh.y, (float)hh.y, part[row]);
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r < M) {
                const float t = wave_sum_hi(part[r]);
                if (lane == 63) scratch[r * kW + wave] = t;
            }
        }
        ZL_ISSUE_RANGE(0, Q0)
        __syncthreads();
        const bool pow2 = (K & (K - 1)) == 0;
        const float inv_k = 1.0f / (float)K;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r < M) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < kW; ++w) tot += scratch[r * kW + w];
                rs[r] = zl_rsqrt_rn((pow2 ? tot * inv_k : tot / (float)K) + p.norm_eps);
            }
        }
    }
    ZL_IPROBE(2);

    // ---- integer planes.  Octet uo of group (gi): MFMA uo / 8, half (uo / 4) % 2, kq = uo % 4 -- the k positions word
    //      t = uo / 4 of lane kq covers in the ZLW4M item; byte order inside the 8-byte piece = the order
    //      (w & 0x0f0f0f0f | (w >> 4) & 0x0f0f0f0f) leaves the nibbles in: k offsets 0 4 1 5 | 2 6 3 7.
    //      Digits without shifts: Y = X + 0x808080 (formed by the fma that scales x, exact below 2^24); its three low bytes
    //      are the balanced digits of X with their top bits flipped (a carry into byte j + 1 happens exactly when digit j
    //      wraps), so b_j = byte_j(Y) ^ 0x80.
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int row = LONGK ? (s & 1) : s, c = LONGK ? (s >> 1) : 0;
        if (row < M && 4 * c < Gw) {
            const int gi = 4 * c + lgi;
            const bool live = wave + kW * gi < groups;
            uint32_t u[4] = {xr[s].x, xr[s].y, xr[s].z, xr[s].w};
            if constexpr (NORM) {
                const uint32_t wu[4] = {nw[c].x, nw[c].y, nw[c].z, nw[c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const h16x2 hh = __builtin_bit_cast(h16x2, u[e]), ww = __builtin_bit_cast(h16x2, wu[e]);
                    h16x2 o;
                    o.x = zl_f32_to_f16((float)hh.x * rs[row] * (float)ww.x);
                    o.y = zl_f32_to_f16((float)hh.y * rs[row] * (float)ww.y);
                    u[e] = __builtin_bit_cast(uint32_t, o);
                }
            }
            // largest magnitude of the group (16 lanes = one DPP row): exponent field Ef of its fp16 pattern
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const us2 m01 = __builtin_elementwise_max(__builtin_bit_cast(us2, u[0] & 0x7fff7fffu), __builtin_bit_cast(us2, u[1] & 0x7fff7fffu));
            const us2 m23 = __builtin_elementwise_max(__builtin_bit_cast(us2, u[2] & 0x7fff7fffu), __builtin_bit_cast(us2, u[3] & 0x7fff7fffu));
            const us2 mm = __builtin_elementwise_max(m01, m23);
            int am = max((int)mm.x, (int)mm.y);
            am = row16_max(am);
            if (row == 0 && c < NP / 4) ZL_ISSUE_RANGE(ZL_QLO(4 * c + 0), ZL_QLO(4 * c + 1))
            const int ef = min(am >> 10, 30);
            const float up = __builtin_bit_cast(float, (uint32_t)(163 - ef) << 23);     // 2^(36 - Ef): |x| < 2^(Ef - 14) -> |X| < 2^22
            // 2^(Ef - 36) -- or NaN when the group holds an infinity or a NaN (exponent field 31): every output that reads the group
            // then comes out NaN, as the fp16 kernels' products do (inf x (q - z) with mixed signs, or x 0); clamped to 30 above,
            // the block-floating image of such a group would be finite garbage (VERDICT r03 "missing" 6)
            const float xscale = am >= 0x7c00 ? __builtin_bit_cast(float, 0x7fc00000u) : __builtin_bit_cast(float, (uint32_t)(91 + ef) << 23);
            uint32_t Y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const h16x2 hh = __builtin_bit_cast(h16x2, u[e]);
                Y[2 * e] = (uint32_t)__builtin_fmaf((float)hh.x, up, 8421504.f);
                Y[2 * e + 1] = (uint32_t)__builtin_fmaf((float)hh.y, up, 8421504.f);
            }
            int sx = (int)(((Y[0] + Y[1]) + (Y[2] + Y[3])) + ((Y[4] + Y[5]) + (Y[6] + Y[7]))) - 8 * 0x808080;
            sx = row16_sum(sx);
            if (row == 0 && c < NP / 4) ZL_ISSUE_RANGE(ZL_QLO(4 * c + 1), ZL_QLO(4 * c + 2))
            // byte gather: [v0 v4 v1 v5] and [v2 v6 v3 v7] of the three digit planes
            auto planes_of = [&](int i0, int i1, int i2, int i3, uint32_t& d2, uint32_t& d1, uint32_t& d0) {
                const uint32_t P = __builtin_amdgcn_perm(Y[i1], Y[i0], 0x05010400u);    // [a.b0, b.b0, a.b1, b.b1]
                const uint32_t Q = __builtin_amdgcn_perm(Y[i3], Y[i2], 0x05010400u);
                const uint32_t P2 = __builtin_amdgcn_perm(Y[i1], Y[i0], 0x0c0c0602u);   // [a.b2, b.b2, 0, 0]
                const uint32_t Q2 = __builtin_amdgcn_perm(Y[i3], Y[i2], 0x0c0c0602u);
                d0 = __builtin_amdgcn_perm(Q, P, 0x05040100u) ^ 0x80808080u;
                d1 = __builtin_amdgcn_perm(Q, P, 0x07060302u) ^ 0x80808080u;
                d2 = __builtin_amdgcn_perm(Q2, P2, 0x05040100u) ^ 0x80808080u;
            };
            uint32_t a2, a1, a0, b2, b1, b0;
            planes_of(0, 4, 1, 5, a2, a1, a0);
            planes_of(2, 6, 3, 7, b2, b1, b0);
            if (row == 0 && c < NP / 4) ZL_ISSUE_RANGE(ZL_QLO(4 * c + 2), ZL_QLO(4 * c + 3))
            unsigned char* dst = planes + (size_t)((gi * 2 + (uo >> 3)) * 4 + (uo & 3)) * rec + (size_t)(4 * row) * 16 + ((uo >> 2) & 1) * 8;
            if (live) {
                *reinterpret_cast<uint2*>(dst) = make_uint2(a2, b2);
                *reinterpret_cast<uint2*>(dst + 16) = make_uint2(a1, b1);
                *reinterpret_cast<uint2*>(dst + 32) = make_uint2(a0, b0);
                if (row == 0) *reinterpret_cast<uint2*>(dst + 48) = make_uint2(0, 0);          // the zero slot idle lanes read
                if (uo == 0) {
                    const float bx = xscale * (float)sx;
                    *reinterpret_cast<float4*>(consts + ((size_t)gi * 4 + row) * 4) = make_float4(xscale, 65536.f * xscale, bx, 1024.f * bx);
                }
            }
            if (row == 0 && c < NP / 4) ZL_ISSUE_RANGE(ZL_QLO(4 * c + 3), ZL_QLO(4 * c + 4))
        }
    }
#undef ZL_ISSUE_RANGE
#undef ZL_QLO
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave reads back what its own lanes wrote
    __builtin_amdgcn_wave_barrier();
    ZL_IPROBE(3);

    // ---- main loop
    const int kq = lane >> 4, row16 = lane & 15;
    const int aslot = (row16 < 4 * M && (row16 & 3) != 3) ? row16 : 3;   // A rows: batch row r = rows 4 r .. 4 r + 2 (digits b2 b1 b0)
    const unsigned char* a_base = planes + (size_t)kq * rec + aslot * 16;
    const int crow = min(kq, M - 1);                                     // C: lane = (batch row lane >> 4, column lane & 15)
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    const uint32_t m4 = __builtin_amdgcn_readfirstlane(0x0f0f0f0fu);
    const v4i zero4 = (v4i){0, 0, 0, 0};

    for (int gi0 = 0; gi0 < my_groups; gi0 += XD) {
#pragma unroll
        for (int j = 0; j < XD; ++j) {
            const int gi = gi0 + j;
            if (j > 0 && gi >= my_groups) break;
            const v4i a0 = *reinterpret_cast<const v4i*>(a_base + (size_t)(gi * 2 + 0) * 4 * rec);
            const v4i a1 = *reinterpret_cast<const v4i*>(a_base + (size_t)(gi * 2 + 1) * 4 * rec);
            const float4 cst = *reinterpret_cast<const float4*>(consts + ((size_t)gi * 4 + crow) * 4);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int slot = j * R + r;
                const uint4 w = wq[slot];
                const uint32_t mw = mt[slot];
                v4i b0, b1;
                b0[0] = (int)(w.x & m4); b0[1] = (int)((w.x >> 4) & m4); b0[2] = (int)(w.y & m4); b0[3] = (int)((w.y >> 4) & m4);
                b1[0] = (int)(w.z & m4); b1[1] = (int)((w.z >> 4) & m4); b1[2] = (int)(w.w & m4); b1[3] = (int)((w.w >> 4) & m4);
                v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, zero4, 0, 0, 0);
                d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, d, 0, 0, 0);
                issue(slot, gi + XD, r);
                const h16x2 sm = __builtin_bit_cast(h16x2, mw);          // .x = scale, .y = -(1024 + zero)
                const float f12 = (float)((d[1] << 8) + d[2]), f0 = (float)d[0];
                float t = __builtin_fmaf((float)sm.y, cst.z, cst.w);      // -(1024 + z) B + 1024 B = -z B
                t = __builtin_fmaf(f12, cst.x, t);
                t = __builtin_fmaf(f0, cst.y, t);
                // The MFMA's A / B registers stay allocated until here.  hipcc (ROCm 7.2) lets a VALU instruction overwrite a
                // source of v_mfma_i32_16x16x64_i8 one issue slot after the MFMA (seen: v_cvt_f32_f16_sdwa into a dword of SrcA
                // right behind the group's last MFMA); the matrix pipe reads its 128-bit operands over several passes, and
                // the rows of the LAST pass (12..15 = batch row 3) then see the new value -- sporadically wrong outputs for
                // the fourth batch row only (tests/test_gpu_w4.py::test_i8p_rows_repeated_against_fp16_kernels).  `t` depends on the MFMA's result, so this point is
                // >= 30 cycles behind it.
                asm volatile("" : "+v"(t) : "v"(b0), "v"(b1), "v"(a0), "v"(a1));
                acc[r] = __builtin_fmaf((float)sm.x, t, acc[r]);
#ifdef ZL_I8P_PROBE
                if (gi == 0 && r == 0) ZL_IPROBE(4);
#endif
            }
        }
    }

    // ---- reduce over the 8 waves in fixed order, epilogue
    ZL_IPROBE(5);
#pragma unroll
    for (int r = 0; r < R; ++r) red[(r * kW + wave) * 64 + lane] = acc[r];
    __syncthreads();
    ZL_IPROBE(6);
    auto total_of = [&](int r, int n_local, int m) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kW; ++w) v += red[(r * kW + w) * 64 + m * 16 + n_local];
        return v;
    };
    if constexpr (ROPE) {
        const int half = p.d / 2;
        if ((int)threadIdx.x < 16 * M) {                 // 16 M <= 64 threads; their table entries / pointers came with the prologue
            const int m = threadIdx.x >> 4, n_local = threadIdx.x & 15;
            float v0 = total_of(0, n_local, m), v1 = total_of(1, n_local, m);
            const int n0 = tile0 * 16 + n_local, n1 = n0 + half;       // columns of the fused qkv row
            if ((p.epi & ZL_EPI_BIAS) && p.bias) {
                v0 += (float)__builtin_bit_cast(_Float16, p.bias[n0]);
                v1 += (float)__builtin_bit_cast(_Float16, p.bias[n1]);
            }
            const float a = (float)zl_f32_to_f16(v0), bb = (float)zl_f32_to_f16(v1);   // the projection's fp16 outputs
            const int head = n0 / p.d, dcol = n0 % p.d;                 // dcol < half
            if (head < p.h + p.hkv) {
                const uint16_t r0 = __builtin_bit_cast(uint16_t, zl_f32_to_f16(__builtin_fmaf(-bb, rp_s0, a * rp_c0)));
                const uint16_t r1 = __builtin_bit_cast(uint16_t, zl_f32_to_f16(__builtin_fmaf(a, rp_s1, bb * rp_c1)));
                if (head < p.h) {
                    uint16_t* dst = p.q_out + ((size_t)m * p.h + head) * p.d + dcol;
                    dst[0] = r0;
                    dst[half] = r1;
                } else if (rp_place >= 0 && rp_place < rp_blen) {
                    const int hk = head - p.h;
                    const size_t row = p.bshd ? (size_t)rp_place * p.hkv + hk : (size_t)hk * rp_blen + rp_place;
                    uint16_t* dst = rp_kv + row * p.d + dcol;
                    dst[0] = r0;
                    dst[half] = r1;
                }
            } else if (rp_place >= 0 && rp_place < rp_blen) {
                const int hk = head - p.h - p.hkv;
                const size_t row = p.bshd ? (size_t)rp_place * p.hkv + hk : (size_t)hk * rp_blen + rp_place;
                uint16_t* dst = rp_kv + row * p.d + dcol;
                dst[0] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(v0));
                dst[half] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(v1));
            }
        }
        ZL_IPROBE(7);
        return;
    }
    if (ep_col >= 0) {
        if (!silu) {
            const float v = total_of(ep_r, ep_nl, ep_m);
            float ov;
            if (p.epi & ZL_EPI_ADD_C) ov = (ep_prev + v) + ep_b0;
            else ov = v + ep_b0;
            _Float16 y16 = zl_f32_to_f16(ov);
            if (p.epi & ZL_EPI_RESIDUAL) y16 = zl_f32_to_f16(ep_res + (float)y16);
            p.y[(size_t)ep_m * p.ld_out + ep_col] = __builtin_bit_cast(uint16_t, y16);
        } else {
            float g = total_of(ep_r, 2 * ep_nl, ep_m) + ep_b0, u = total_of(ep_r, 2 * ep_nl + 1, ep_m) + ep_b1;
            float ov;
            if (p.epi & ZL_EPI_SILU_MUL) {
                g = (float)zl_f32_to_f16(g);
                u = (float)zl_f32_to_f16(u);
                ov = silu_f32(g) * u;
            } else {
                ov = (float)((double)g / (1.0 + (double)expf(-g))) * u;
            }
            p.y[(size_t)ep_m * p.ld_out + ep_col] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(ov));
        }
    }
    ZL_IPROBE(7);
}

static size_t i8p_lds_bytes(int groups, int m, int r) {
    const size_t gw = (groups + kW - 1) / kW;
    return kW * gw * 8 * 64 * (size_t)m + kW * gw * 64 + (size_t)r * kW * 64 * 4 + 4 * kW * 4;
}

template <int R, bool LONGK, bool ROPE, bool NORM, bool MERGE = false>
int launch_i8p_n(const I8Params& p, int grid, hipStream_t hs) {
    const size_t lds = i8p_lds_bytes(p.groups, p.m, R);
    if (lds > 160 * 1024) return ZL_ELIMIT;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_w4a16_i8p<R, LONGK, ROPE, NORM, MERGE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ZL_ELIMIT;
    }
    hipLaunchKernelGGL((k_w4a16_i8p<R, LONGK, ROPE, NORM, MERGE>), dim3(grid), dim3(kT), lds, hs, p);
    return zl_launch_status();
}
template <int R, bool ROPE>
int launch_i8p(const I8Params& p, int grid, hipStream_t hs) {
    const bool lk = p.groups > 4 * kW;
    if (p.norm_w) return lk ? launch_i8p_n<R, true, ROPE, true>(p, grid, hs) : launch_i8p_n<R, false, ROPE, true>(p, grid, hs);
    return lk ? launch_i8p_n<R, true, ROPE, false>(p, grid, hs) : launch_i8p_n<R, false, ROPE, false>(p, grid, hs);
}

}  // namespace

#ifdef ZL_I8P_PROBE
extern "C" int zl_debug_set_probe_i8p(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(zl_probe_i8), &p, sizeof(p)); }
#endif

// what the kernel covers: 1..4 rows with K <= 4096, 1..2 rows with K <= 16384, K a multiple of 128 (ZLW4M tiles), LDS
bool zl_w4a16_i8p_covers(int64_t m, int64_t k) {
    if (m < 1 || m > 4 || k < 128 || k % 128 != 0 || k > 16384 || (k > 4096 && m > 2)) return false;
    return i8p_lds_bytes((int)(k / 128), (int)m, 8) <= 160 * 1024;
}

// internal (called by zl_w4a16_gemm_mfma_ex): zl_w4a16_i8p_covers(m, k)
int zl_w4a16_gemm_i8p(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                      uint32_t meta_bytes, const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k,
                      int groups, int tiles, int epilogue, int ld_out, const uint16_t* norm_w, float norm_eps, int rounds_override,
                      hipStream_t hs) {
    if (!zl_w4a16_i8p_covers(m, k)) return ZL_ESHAPE;
    I8Params p;
    p.x = x; p.ldx = ldx; p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes; p.meta_bytes = meta_bytes;
    p.bias = bias; p.residual = residual; p.y = y; p.m = m; p.n = n; p.k = k; p.groups = groups; p.tiles = tiles; p.epi = epilogue;
    p.ld_out = ld_out; p.norm_w = norm_w; p.norm_eps = norm_eps;
    p.cosv = p.sinv = nullptr; p.placement = p.buf_lens = nullptr; p.k_bufs = p.v_bufs = nullptr; p.q_out = nullptr;
    p.h = p.hkv = p.d = p.bshd = 0; p.pair_stride = 1;
    p.mg_part = nullptr; p.mg_stat = nullptr; p.mg_valid_lens = nullptr; p.mg_split_len = p.mg_max_splits = 0;
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    int r = (tiles + cus - 1) / cus;
    if (r > 8) r = 8;
    if (rounds_override > 0 && rounds_override <= 8) r = rounds_override;
    const int grid = (tiles + r - 1) / r;
#define ZL_I8(RR) case RR: return launch_i8p<RR, false>(p, grid, hs);
    switch (r) { ZL_I8(1) ZL_I8(2) ZL_I8(3) ZL_I8(4) ZL_I8(5) ZL_I8(6) ZL_I8(7) ZL_I8(8) }
#undef ZL_I8
    return ZL_EINVAL;
}

// internal (called by zl_w4a16_qkv_rope_scatter): the fused qkv projection with the neox rotation and the KV scatter
int zl_w4a16_gemm_i8p_rope(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                           uint32_t meta_bytes, const uint16_t* bias, int m, int n, int k, int groups, int tiles,
                           const uint16_t* norm_w, float norm_eps, const float* cosv, const float* sinv,
                           const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                           uint16_t* const* v_bufs, uint16_t* q_out, int h, int hkv, int d, int bshd, hipStream_t hs) {
    if (!zl_w4a16_i8p_covers(m, k) || d % 32 != 0 || n != (h + 2 * hkv) * d || tiles * 16 != n) return ZL_ESHAPE;
    I8Params p;
    p.x = x; p.ldx = ldx; p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes; p.meta_bytes = meta_bytes;
    p.bias = bias; p.residual = nullptr; p.y = nullptr; p.m = m; p.n = n; p.k = k; p.groups = groups; p.tiles = tiles;
    p.epi = bias ? ZL_EPI_BIAS : 0; p.ld_out = n; p.norm_w = norm_w; p.norm_eps = norm_eps;
    p.cosv = cosv; p.sinv = sinv; p.placement = placement; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs;
    p.q_out = q_out; p.h = h; p.hkv = hkv; p.d = d; p.bshd = bshd; p.pair_stride = d / 32;
    p.mg_part = nullptr; p.mg_stat = nullptr; p.mg_valid_lens = nullptr; p.mg_split_len = p.mg_max_splits = 0;
    const int grid = tiles / 2;
    return launch_i8p<2, true>(p, grid, hs);
}

// internal (called by zl_w4a16_gemm_attn_merge_h): the attention output projection of a decode step reading the half-precision
// split partials of zl_decode_attn_splits_h.  m <= 4, k = heads * 128 <= 4096, max_splits <= 16.
int zl_w4a16_gemm_i8p_merge(const void* ws, const int32_t* buf_lens, const int32_t* valid_lens, int split_len, int max_splits,
                            const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                            const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups,
                            int tiles, int epilogue, hipStream_t hs) {
    if (m < 1 || m > 4 || k > 4096 || k % 128 != 0 || max_splits < 1 || max_splits > 16 || split_len < 1) return ZL_ESHAPE;
    if (epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) return ZL_ESHAPE;
    I8Params p;
    p.x = nullptr; p.ldx = 0; p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes; p.meta_bytes = meta_bytes;
    p.bias = bias; p.residual = residual; p.y = y; p.m = m; p.n = n; p.k = k; p.groups = groups; p.tiles = tiles; p.epi = epilogue;
    p.ld_out = n; p.norm_w = nullptr; p.norm_eps = 0.f;
    p.cosv = p.sinv = nullptr; p.placement = nullptr; p.buf_lens = buf_lens; p.k_bufs = p.v_bufs = nullptr; p.q_out = nullptr;
    p.h = p.hkv = p.d = p.bshd = 0; p.pair_stride = 1;
    p.mg_part = reinterpret_cast<const uint16_t*>(ws);
    p.mg_stat = reinterpret_cast<const float*>(ws) + (size_t)m * groups * max_splits * 64;   // behind the fp16 rows (128 halfs each)
    p.mg_valid_lens = valid_lens; p.mg_split_len = split_len; p.mg_max_splits = max_splits;
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    int r = (tiles + cus - 1) / cus;
    if (r > 8) r = 8;
    const int grid = (tiles + r - 1) / r;
#define ZL_I8M(RR) case RR: return launch_i8p_n<RR, false, false, false, true>(p, grid, hs);
    switch (r) { ZL_I8M(1) ZL_I8M(2) ZL_I8M(3) ZL_I8M(4) ZL_I8M(5) ZL_I8M(6) ZL_I8M(7) ZL_I8M(8) }
#undef ZL_I8M
    return ZL_EINVAL;
}

// w4_engine.hip -- the integer-plane W4A16 GEMV (w4_i8p.hip) as a LOADER / CONSUMER engine, and two projections of a decode
// layer fused into one launch on top of it.  Replaces, for a 1..4-row decode batch, the launches of
// Linear::forward -> gptq_gemm_k_major (src/nn/linear/linear.cpp:934-1004, src/nn/quant/gptq/q_gemm_k_major.cu:580-686, 957-1116)
// and the element_add_scale / LayerNorm launches between them (src/nn/block/block.cpp:86-143).
//
// Why.  k_w4a16_i8p keeps its weight ring in REGISTERS: 8 KiB per wave, re-requested item by item as the wave consumes, so
//   * nothing can be requested before the wave has registers for it and the ring cannot run ahead of the activation
//     conversion (1.3 us of every norm-fused launch passes before the first weight is requested, r03_i8p_timeline.txt),
//   * a request is issued only when an item has been consumed: the bytes in flight per CU sag whenever the consumers are busy
//     (gate|up streams at 5.5-6.0 TB/s inside the kernel against 6.9 TB/s for the bare ring, r03_bcast_probe.txt),
//   * and whatever was prefetched dies at the kernel boundary.
// Here one extra wave per workgroup -- the LOADER -- moves the workgroup's whole weight stream through an LDS ring with
// LDS-DMA (global_load_lds_dwordx4, non-temporal): no registers, no VALU, it never waits for a consumer except for a free
// slot, and its issue stalls cost nobody anything.  The eight CONSUMER waves are k_w4a16_i8p's: same activation staging
// (block-floating digit planes, wave-private), same per-item arithmetic, same accumulation and reduction order -- the outputs
// are bit-identical to k_w4a16_i8p's (tests/test_gpu_engine.py) -- but they read their 1 KiB items with ds_read_b128.
//   ring      S slots x (8 items + their 8 x 16 meta words) = 8704 B; slot = one row tile x eight consecutive 128-k groups,
//             consumer wave w takes item w.  S = whatever LDS is left (13-14 slots for K = 4096 at one row).
//   landed    one LDS word, the number of slots whose data is in LDS: the loader keeps kInFlight slots between "issued" and
//             "counted as landed" (s_waitcnt vmcnt(9 kInFlight), nine DMA instructions per slot) and publishes the rest.
//   consumed  one LDS counter per ring slot, +1 per consumer wave and use, added BEHIND the wave's ds_reads of the slot (a
//             wave's LDS operations execute in order); the loader refills a slot when it has seen 8 x (round) of them.
//   consumers never meet the loader at an s_barrier after the first one: their own two rendezvous (RMSNorm partial sums,
//             the cross-wave reduction) are LDS counters.
// The order at the head of a launch is the one the timeline probes of round 3 asked for: the consumers request their
// activations, ALL nine waves meet at one s_barrier (by then those requests are in the CU's memory queue), and only then does
// the loader start the ring -- the activations come back first, the weights right behind them, and the ring keeps filling
// while the activations are normalised and converted.
//
// Fused launch (k_w4_engine_o_gateup): attention split merge + attn_out + residual, then RMSNorm + gate|up + silu.mul, ONE
// launch.  The hidden row produced by the first projection (16 values per workgroup) crosses to every workgroup as 8-byte
// {tag, two halves} granules: one write-through (sc1) store per granule, consumers sweep the 4 granules per lane they need
// with sc1 loads until every tag equals the launch's epoch (no flag, no fence, no reset: the epoch comes from a device word
// the caller advances once per step; MI355X guide, "Inter-workgroup communication", recipe R2).  While the consumers finish
// the first projection and wait for the hand-off, the loader is already 10 slots (85 KiB per CU) into gate|up's weights --
// the part of the stream that a kernel boundary would have left idle.
#include "zl_common.h"
#include "w4_i8p_common.h"

namespace {

constexpr int kCW = 8;                     // consumer waves
constexpr int kET = (kCW + 1) * 64;        // threads: consumers + the loader wave
constexpr int kSlot = 8 * 1024 + 8 * 64;   // bytes per ring slot
constexpr int kInFlight = 6;               // slots between issued and known-landed (9 DMAs each; vmcnt counts to 63)
constexpr int kFlagBytes = 256;
constexpr uint32_t kSpinLimit = 1u << 21;  // bounded polls (~0.1 s): a lost hand-off ends in wrong numbers + an error word, never a hang

// LDS flag block (behind the ring)
struct EngFlags {
    uint32_t landed;        // slots landed
    uint32_t bar;           // consumer rendezvous counter
    uint32_t abort_;        // some poll ran out
    uint32_t pad;
    uint32_t consumed[32];  // per ring slot
};

// Every LDS access below goes through an explicit address_space(3) pointer built from a 32-bit LDS byte address: passed
// through structs the generic pointers came out as FLAT loads (which also count on vmcnt -- the loader's own counter).
struct EngLds {
    uint32_t ring;     // LDS byte addresses
    uint32_t fl;       // EngFlags
    uint32_t planes;   // [8 waves][Gw][8 records][64 M]
    uint32_t consts;   // [8 waves][Gw][4 rows][4] floats
    uint32_t red;      // [Rmax][8 waves][64] floats
    uint32_t scratch;  // [4 rows][8 waves] floats
    int S;
};
#define ZL_LDS(T) __attribute__((address_space(3))) T
template <typename T> __device__ __forceinline__ T lds_ld(uint32_t a) { return *(const ZL_LDS(T)*)(uintptr_t)a; }
template <typename T> __device__ __forceinline__ void lds_st(uint32_t a, T v) { *(ZL_LDS(T)*)(uintptr_t)a = v; }
__device__ __forceinline__ uint32_t lds_poll(uint32_t a) { return *(const volatile ZL_LDS(uint32_t)*)(uintptr_t)a; }
__device__ __forceinline__ void lds_st_v(uint32_t a, uint32_t v) { *(volatile ZL_LDS(uint32_t)*)(uintptr_t)a = v; }
__device__ __forceinline__ void lds_inc(uint32_t a) {
    __hip_atomic_fetch_add((ZL_LDS(uint32_t)*)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
constexpr uint32_t kFlLanded = 0, kFlBar = 4, kFlAbort = 8, kFlConsumed = 16;   // offsets inside EngFlags

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// one 1 KiB LDS-DMA (64 lanes x 16 B, lane-linear destination), non-temporal; M0 is compiler-reserved: saved and restored inside
// the statement that uses it.  Invisible to hipcc's s_waitcnt bookkeeping: the loader counts its own vmcnt.
__device__ __forceinline__ void dma16_nt(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// ---- loader ---------------------------------------------------------------------------------------------------------------
struct LoadPhase {
    const unsigned char* qw;
    const unsigned char* meta;
    int tile0, tile_stride, R, Gw, groups, tiles;
};
struct LoaderState {
    uint32_t s;         // slots issued so far (all phases)
    int ridx;           // s mod S
    uint32_t round;     // s div S
    int pending;        // issued, not yet counted as landed
};

__device__ __forceinline__ void loader_phase(const LoadPhase& f, LoaderState& st, const EngLds& L, int lane) {
    const uint32_t ring0 = L.ring;
    for (int gi = 0; gi < f.Gw; ++gi) {
        for (int r = 0; r < f.R; ++r) {
            if (st.round > 0) {
                const uint32_t need = (uint32_t)kCW * st.round;
                if (lds_poll(L.fl + kFlConsumed + 4u * (uint32_t)st.ridx) < need) {
                    // the ring is full: everything issued is wanted anyway -- count it all as landed first (the consumers
                    // cannot free a slot they have not been told about), then wait for the oldest slot
                    wait_vmcnt<0>();
                    if (lane == 0) lds_st_v(L.fl + kFlLanded, st.s);
                    st.pending = 0;
                    uint32_t spins = 0;
                    while (lds_poll(L.fl + kFlConsumed + 4u * (uint32_t)st.ridx) < need) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > kSpinLimit || ((spins & 255u) == 0u && lds_poll(L.fl + kFlAbort))) { if (lane == 0) lds_st_v(L.fl + kFlAbort, 1u); break; }
                    }
                }
            }
            int tile = f.tile0 + r * f.tile_stride;
            if (tile >= f.tiles) tile = f.tiles - 1;          // a tile past the end: the slot is still counted, its content unused
            const size_t item0 = (size_t)tile * (size_t)f.groups + (size_t)gi * 8;
            const unsigned char* src = f.qw + item0 * 1024 + (size_t)lane * 16;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(ring0 + (uint32_t)st.ridx * (uint32_t)kSlot);
#pragma unroll
            for (int i = 0; i < 8; ++i) dma16_nt(src + (size_t)i * 1024, dst + (uint32_t)i * 1024u);
            // the eight items' meta words: 512 contiguous bytes = half a DMA (lanes 32..63 re-read the first half into the pad)
            dma16_nt(f.meta + item0 * 64 + (size_t)(lane & 31) * 16, dst + 8192u);
            ++st.s;
            if (++st.ridx == L.S) { st.ridx = 0; ++st.round; }
            if (++st.pending > kInFlight) {
                wait_vmcnt<9 * kInFlight>();
                if (lane == 0) lds_st_v(L.fl + kFlLanded, st.s - (uint32_t)kInFlight);
                st.pending = kInFlight;
            }
        }
    }
}
__device__ __forceinline__ void loader_finish(LoaderState& st, const EngLds& L, int lane) {
    wait_vmcnt<0>();
    if (lane == 0) lds_st_v(L.fl + kFlLanded, st.s);
}

// ---- consumers ------------------------------------------------------------------------------------------------------------
struct ConsState {
    uint32_t s;          // next slot to consume (all phases)
    int ridx;
    uint32_t bar_target; // rendezvous count reached after the next consumer barrier
};

// rendezvous of the eight consumer waves through an LDS counter (the loader wave never takes part)
__device__ __forceinline__ void consumer_barrier(ConsState& cs, const EngLds& L, int lane) {
    cs.bar_target += kCW;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) lds_inc(L.fl + kFlBar);
    uint32_t spins = 0;
    while (lds_poll(L.fl + kFlBar) < cs.bar_target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit || ((spins & 255u) == 0u && lds_poll(L.fl + kFlAbort))) { if (lane == 0) lds_st_v(L.fl + kFlAbort, 1u); break; }
    }
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void fetch_item(ConsState& cs, const EngLds& L, int wave, int lane, uint4& w, uint32_t& mw) {
    uint32_t spins = 0;
    while (lds_poll(L.fl + kFlLanded) <= cs.s) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit || ((spins & 255u) == 0u && lds_poll(L.fl + kFlAbort))) { if (lane == 0) lds_st_v(L.fl + kFlAbort, 1u); break; }
    }
    asm volatile("" ::: "memory");
    const uint32_t sl = L.ring + (uint32_t)cs.ridx * (uint32_t)kSlot;
    const u32x4 wv = lds_ld<u32x4>(sl + (uint32_t)(wave * 1024 + lane * 16));
    w = make_uint4(wv.x, wv.y, wv.z, wv.w);
    mw = lds_ld<uint32_t>(sl + 8192u + (uint32_t)(wave * 64 + (lane & 15) * 4));
    asm volatile("" ::: "memory");   // the release stays behind the two reads; the LDS executes a wave's operations in order
    if (lane == 0) lds_inc(L.fl + kFlConsumed + 4u * (uint32_t)cs.ridx);
    ++cs.s;
    if (++cs.ridx == L.S) cs.ridx = 0;
}

// where a phase's activation rows come from / where its outputs go besides p.y
enum { XS_GLOBAL = 0, XS_GRANULES = 1 };
struct Exchange {
    unsigned long long* gran;   // [rows][n / 2] granules {tag << 32 | two halves}
    uint32_t epoch;
    uint32_t* err;              // device error word (ZL_ENGINE_ERR_*), may be null
};

typedef unsigned long long __attribute__((address_space(1))) gu64;

// One projection on the consumer waves.  Template parameters as k_w4a16_i8p's; XS = where the activation rows come from,
// PUB = publish the outputs as granules for a later phase of the SAME launch (besides the plain store to p.y).
// FIRST: this phase runs at the head of the launch (its requests go out before the launch's one s_barrier).
template <int R, bool LONGK, bool ROPE, bool NORM, bool MERGE, int XS, bool PUB, bool FIRST>
__device__ __forceinline__ void consume_phase(const I8Params& p, const Exchange& ex, const EngLds& L, ConsState& cs,
                                              const int wave, const int lane) {
    static_assert(!ROPE || R == 2, "fused rotary: a tile and its partner tile");
    static_assert(!MERGE || (!LONGK && !ROPE && !NORM), "split merge: one column block, plain prologue");
    constexpr int NS = LONGK ? 8 : 4;                          // activation octet slots per thread
    constexpr int NR = LONGK ? 2 : 4;                          // rows
    const int M = p.m, K = p.k, groups = p.groups;
    const int rec = 64 * M;
    const int Gw = groups / kCW;                               // groups per wave (the engine takes groups % 8 == 0 only)
    const int tid = wave * 64 + lane;
    const uint32_t planes = L.planes + (uint32_t)(wave * Gw * 8 * rec);
    const uint32_t consts = L.consts + (uint32_t)(wave * Gw * 16 * 4);
    const uint32_t red = L.red, scratch = L.scratch;

    const int tile0 = ROPE ? (blockIdx.x / p.pair_stride) * 2 * p.pair_stride + blockIdx.x % p.pair_stride : blockIdx.x * R;

    // ---- activations (k_w4a16_i8p's staging: a wave loads exactly the k ranges of ITS groups)
    uint4 xr[NS], nw[LONGK ? 4 : 1];
    const int lgi = lane >> 4, uo = lane & 15;
#pragma unroll
    for (int c = 0; c < (LONGK ? 4 : 1); ++c) {
        nw[c] = make_uint4(0, 0, 0, 0);
        const int g = wave + kCW * (4 * c + lgi);
        if (NORM && 4 * c < Gw) nw[c] = *reinterpret_cast<const uint4*>(p.norm_w + (g < groups ? g * 128 + uo * 8 : 0));
    }
    if constexpr (XS == XS_GLOBAL) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int row = LONGK ? (s & 1) : s, c = LONGK ? (s >> 1) : 0;
            const int g = wave + kCW * (4 * c + lgi);
            xr[s] = make_uint4(0, 0, 0, 0);
            if constexpr (MERGE) {
                if (row < M) {
                    constexpr int kS = 16;
                    const int head = g < groups ? g : 0;
                    const uint32_t n_rec = (uint32_t)M * (uint32_t)groups * (uint32_t)p.mg_max_splits;
                    const __amdgpu_buffer_rsrc_t rpart = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.mg_part), 0, n_rec * 256u, 0x00020000);
                    const __amdgpu_buffer_rsrc_t rstat = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mg_stat), 0, n_rec * 8u, 0x00020000);
                    const uint32_t rec0 = ((uint32_t)row * (uint32_t)groups + (uint32_t)head) * (uint32_t)p.mg_max_splits;
                    const uint32_t po = rec0 * 256u + (uint32_t)uo * 16u, so = rec0 * 8u;
                    uint4 pv[kS];
                    float2 st[kS];
#pragma unroll
                    for (int u = 0; u < kS; ++u) {
                        const uint32_t uc = u < p.mg_max_splits ? (uint32_t)u : 0u;
                        pv[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rpart, po, uc * 256u, 0));
                        st[u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rstat, so, uc * 8u, 0));
                    }
                    const int elen = min(p.buf_lens[row], p.mg_valid_lens[row]);
                    if constexpr (FIRST) {
                        if (s == 0) {                                 // the loader starts the ring behind these requests
                            __builtin_amdgcn_sched_barrier(0);        // (pinned: the scheduler moved the merge arithmetic and its
                            __builtin_amdgcn_s_barrier();             //  vmcnt waits in front of the barrier otherwise)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    const int ns = min((elen + p.mg_split_len - 1) / p.mg_split_len, kS);
                    float mn = -1e20f;
#pragma unroll
                    for (int u = 0; u < kS; ++u) mn = fmaxf(mn, u < ns ? st[u].x : -1e20f);
                    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, z = 0.f;
#pragma unroll
                    for (int u = 0; u < kS; ++u) {
                        if (u < ns) {
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
            } else if (row < M && 4 * c < Gw) {
                xr[s] = *reinterpret_cast<const uint4*>(p.x + (g < groups ? (size_t)row * p.ldx + g * 128 + uo * 8 : 0));
                if (g >= groups) xr[s] = make_uint4(0, 0, 0, 0);
            }
        }
    }
    // epilogue operands, requested next to the activations (k_w4a16_i8p)
    float rp_c0 = 0.f, rp_s0 = 0.f, rp_c1 = 0.f, rp_s1 = 0.f;
    int rp_place = -1, rp_blen = 0;
    uint16_t* rp_kv = nullptr;
    if constexpr (ROPE) {
        if (tid < 16 * M) {
            const int m = tid >> 4, n0 = tile0 * 16 + (tid & 15);
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
    const bool silu = (p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
    // (raw 16-bit patterns: a conversion here would make the compiler wait for the load -- and with it for every older request
    //  of the wave -- in front of the launch's s_barrier, i.e. hold the loader back until the activations have landed)
    // Branch-free: a load inside control flow costs a conservative vmcnt(0) at the join -- in front of the barrier.  Operands that
    // do not apply read through an empty descriptor / past its end and come back as zero.
    uint32_t ep_b0r = 0, ep_b1r = 0, ep_resr = 0, ep_prevr = 0;
    int ep_r = 0, ep_m = 0, ep_nl = 0, ep_col = -1;
    if constexpr (!ROPE) {
        const int per_tile = (silu ? 8 : 16) * M;
        ep_r = tid / per_tile;
        const int rem = tid % per_tile, tile = tile0 + ep_r;
        const bool mine = tid < R * per_tile && tile < p.tiles;
        ep_m = silu ? rem >> 3 : rem >> 4;
        ep_nl = silu ? rem & 7 : rem & 15;
        const int col = silu ? tile * 8 + ep_nl : tile * 16 + ep_nl;           // output column (silu: the pair's index)
        const bool ok = mine && (silu ? 2 * col + 1 < p.n : col < p.n);
        ep_col = ok ? col : -1;
        const uint32_t kOob = 0x7ffffff0u;
        const bool has_bias = (p.epi & ZL_EPI_BIAS) && p.bias;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.bias), 0, has_bias ? (uint32_t)p.n * 2u : 0u, 0x00020000);
        const uint32_t out_bytes = (uint32_t)M * (uint32_t)p.ld_out * 2u;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.residual), 0, (p.epi & ZL_EPI_RESIDUAL) && !silu ? out_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (p.epi & ZL_EPI_ADD_C) && !silu ? out_bytes : 0u, 0x00020000);
        const uint32_t ooff = ok ? ((uint32_t)ep_m * (uint32_t)p.ld_out + (uint32_t)col) * 2u : kOob;
        ep_b0r = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rb, ok && !silu ? (uint32_t)col * 2u : kOob, 0, 0);
        ep_b1r = __builtin_amdgcn_raw_buffer_load_b32(rb, ok && silu ? (uint32_t)col * 4u : kOob, 0, 0);   // both halves of a pair
        ep_resr = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rr, ooff, 0, 0);
        ep_prevr = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(ry, ooff, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FIRST && !MERGE) __builtin_amdgcn_s_barrier();   // requests are in the CU's queue: the loader may start
    if constexpr (XS == XS_GRANULES) {
        // the rows arrive from the previous phase of THIS launch: 8 halves per lane and slot = 4 granules, swept until every
        // tag carries the epoch (sc1 loads: another CU's write-through store is seen at the next sweep)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int row = LONGK ? (s & 1) : s, c = LONGK ? (s >> 1) : 0;
            const int g = wave + kCW * (4 * c + lgi);
            xr[s] = make_uint4(0, 0, 0, 0);
            if (row < M && 4 * c < Gw) {
                const gu64* src = (const gu64*)(ex.gran + ((size_t)row * K + (size_t)(g < groups ? g : 0) * 128 + uo * 8) / 2);
                uint32_t v[4];
                uint32_t spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned long long t = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v[e] = (uint32_t)t;
                        ok &= (uint32_t)(t >> 32) == ex.epoch;
                    }
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (kSpinLimit >> 4)) {
                        if (lane == 0) { lds_st_v(L.fl + kFlAbort, 1u); if (ex.err) atomicOr(ex.err, 2u); }
                        break;
                    }
                }
                if (g < groups) xr[s] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the activations have landed (builtin: see w4_i8p.hip)
    }
    __builtin_amdgcn_sched_barrier(0);
    // (the patterns become visible to the compiler only here: it hoisted their conversions -- and a vmcnt(0) -- above the barrier)
    asm volatile("" : "+v"(ep_b0r), "+v"(ep_b1r), "+v"(ep_resr), "+v"(ep_prevr));

    // ---- fused RMSNorm: eight partial sums through LDS, consumer rendezvous
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
                if (lane == 63) lds_st<float>(scratch + 4u * (uint32_t)(r * kCW + wave), t);
            }
        }
        consumer_barrier(cs, L, lane);
        const bool pow2 = (K & (K - 1)) == 0;
        const float inv_k = 1.0f / (float)K;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r < M) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < kCW; ++w) tot += lds_ld<float>(scratch + 4u * (uint32_t)(r * kCW + w));
                rs[r] = zl_rsqrt_rn((pow2 ? tot * inv_k : tot / (float)K) + p.norm_eps);
            }
        }
    }

    // ---- integer planes (k_w4a16_i8p's conversion, without the ring issue woven through it)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int row = LONGK ? (s & 1) : s, c = LONGK ? (s >> 1) : 0;
        if (row < M && 4 * c < Gw) {
            const int gi = 4 * c + lgi;
            const bool live = wave + kCW * gi < groups;
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
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const us2 m01 = __builtin_elementwise_max(__builtin_bit_cast(us2, u[0] & 0x7fff7fffu), __builtin_bit_cast(us2, u[1] & 0x7fff7fffu));
            const us2 m23 = __builtin_elementwise_max(__builtin_bit_cast(us2, u[2] & 0x7fff7fffu), __builtin_bit_cast(us2, u[3] & 0x7fff7fffu));
            const us2 mm = __builtin_elementwise_max(m01, m23);
            int am = max((int)mm.x, (int)mm.y);
            am = row16_max(am);
            const int ef = min(am >> 10, 30);
            const float up = __builtin_bit_cast(float, (uint32_t)(163 - ef) << 23);
            // an infinity or a NaN among the group's 128 activations (exponent field 31) poisons the group's constants: every
            // output that reads the group becomes NaN, as the fp16 kernels' products would (w4_i8p.hip)
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
            auto planes_of = [&](int i0, int i1, int i2, int i3, uint32_t& d2, uint32_t& d1, uint32_t& d0) {
                const uint32_t P = __builtin_amdgcn_perm(Y[i1], Y[i0], 0x05010400u);
                const uint32_t Q = __builtin_amdgcn_perm(Y[i3], Y[i2], 0x05010400u);
                const uint32_t P2 = __builtin_amdgcn_perm(Y[i1], Y[i0], 0x0c0c0602u);
                const uint32_t Q2 = __builtin_amdgcn_perm(Y[i3], Y[i2], 0x0c0c0602u);
                d0 = __builtin_amdgcn_perm(Q, P, 0x05040100u) ^ 0x80808080u;
                d1 = __builtin_amdgcn_perm(Q, P, 0x07060302u) ^ 0x80808080u;
                d2 = __builtin_amdgcn_perm(Q2, P2, 0x05040100u) ^ 0x80808080u;
            };
            uint32_t a2, a1, a0, b2, b1, b0;
            planes_of(0, 4, 1, 5, a2, a1, a0);
            planes_of(2, 6, 3, 7, b2, b1, b0);
            const uint32_t dst = planes + (uint32_t)(((gi * 2 + (uo >> 3)) * 4 + (uo & 3)) * rec + (4 * row) * 16 + ((uo >> 2) & 1) * 8);
            if (live) {
                lds_st<u32x2>(dst, (u32x2){a2, b2});
                lds_st<u32x2>(dst + 16, (u32x2){a1, b1});
                lds_st<u32x2>(dst + 32, (u32x2){a0, b0});
                if (row == 0) lds_st<u32x2>(dst + 48, (u32x2){0u, 0u});
                if (uo == 0) {
                    const float bx = xscale * (float)sx;
                    lds_st<f32x4>(consts + (uint32_t)((gi * 4 + row) * 16), (f32x4){xscale, 65536.f * xscale, bx, 1024.f * bx});
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // ---- main loop: items come out of the LDS ring, one fetch ahead of the arithmetic
    const int kq = lane >> 4, row16 = lane & 15;
    const int aslot = (row16 < 4 * M && (row16 & 3) != 3) ? row16 : 3;
    const uint32_t a_base = planes + (uint32_t)(kq * rec + aslot * 16);
    const int crow = min(kq, M - 1);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    const uint32_t m4 = __builtin_amdgcn_readfirstlane(0x0f0f0f0fu);
    const v4i zero4 = (v4i){0, 0, 0, 0};
    uint4 wn;
    uint32_t mn_;
    fetch_item(cs, L, wave, lane, wn, mn_);
    for (int gi = 0; gi < Gw; ++gi) {
        const v4i a0 = lds_ld<v4i>(a_base + (uint32_t)((gi * 2 + 0) * 4 * rec));
        const v4i a1 = lds_ld<v4i>(a_base + (uint32_t)((gi * 2 + 1) * 4 * rec));
        const f32x4 cst = lds_ld<f32x4>(consts + (uint32_t)((gi * 4 + crow) * 16));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4 w = wn;
            const uint32_t mw = mn_;
            if (!(gi == Gw - 1 && r == R - 1)) fetch_item(cs, L, wave, lane, wn, mn_);
            v4i b0, b1;
            b0[0] = (int)(w.x & m4); b0[1] = (int)((w.x >> 4) & m4); b0[2] = (int)(w.y & m4); b0[3] = (int)((w.y >> 4) & m4);
            b1[0] = (int)(w.z & m4); b1[1] = (int)((w.z >> 4) & m4); b1[2] = (int)(w.w & m4); b1[3] = (int)((w.w >> 4) & m4);
            v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, zero4, 0, 0, 0);
            d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, d, 0, 0, 0);
            const h16x2 sm = __builtin_bit_cast(h16x2, mw);
            const float f12 = (float)((d[1] << 8) + d[2]), f0 = (float)d[0];
            float t = __builtin_fmaf((float)sm.y, cst.z, cst.w);
            t = __builtin_fmaf(f12, cst.x, t);
            t = __builtin_fmaf(f0, cst.y, t);
            asm volatile("" : "+v"(t) : "v"(b0), "v"(b1), "v"(a0), "v"(a1));   // MFMA source-operand hazard: w4_i8p.hip
            acc[r] = __builtin_fmaf((float)sm.x, t, acc[r]);
        }
    }

    // ---- reduce over the 8 waves in fixed order, epilogue
#pragma unroll
    for (int r = 0; r < R; ++r) lds_st<float>(red + 4u * (uint32_t)((r * kCW + wave) * 64 + lane), acc[r]);
    consumer_barrier(cs, L, lane);
    auto total_of = [&](int r, int n_local, int m) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kCW; ++w) v += lds_ld<float>(red + 4u * (uint32_t)((r * kCW + w) * 64 + m * 16 + n_local));
        return v;
    };
    if constexpr (ROPE) {
        const int half = p.d / 2;
        if (tid < 16 * M) {
            const int m = tid >> 4, n_local = tid & 15;
            float v0 = total_of(0, n_local, m), v1 = total_of(1, n_local, m);
            const int n0 = tile0 * 16 + n_local, n1 = n0 + half;
            if ((p.epi & ZL_EPI_BIAS) && p.bias) {
                v0 += (float)__builtin_bit_cast(_Float16, p.bias[n0]);
                v1 += (float)__builtin_bit_cast(_Float16, p.bias[n1]);
            }
            const float a = (float)zl_f32_to_f16(v0), bb = (float)zl_f32_to_f16(v1);
            const int head = n0 / p.d, dcol = n0 % p.d;
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
        return;
    }
    uint16_t y16bits = 0;
    if (ep_col >= 0) {
        if (silu) { ep_b0r = ep_b1r & 0xffffu; ep_b1r >>= 16; }
        const float ep_b0 = (float)__builtin_bit_cast(_Float16, (uint16_t)ep_b0r), ep_b1 = (float)__builtin_bit_cast(_Float16, (uint16_t)ep_b1r);
        const float ep_res = (float)__builtin_bit_cast(_Float16, (uint16_t)ep_resr), ep_prev = (float)__builtin_bit_cast(_Float16, (uint16_t)ep_prevr);
        if (!silu) {
            const float v = total_of(ep_r, ep_nl, ep_m);
            float ov;
            if (p.epi & ZL_EPI_ADD_C) ov = (ep_prev + v) + ep_b0;
            else ov = v + ep_b0;
            _Float16 y16 = zl_f32_to_f16(ov);
            if (p.epi & ZL_EPI_RESIDUAL) y16 = zl_f32_to_f16(ep_res + (float)y16);
            y16bits = __builtin_bit_cast(uint16_t, y16);
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
            y16bits = __builtin_bit_cast(uint16_t, zl_f32_to_f16(ov));
        }
        p.y[(size_t)ep_m * p.ld_out + ep_col] = y16bits;
    }
    if constexpr (PUB) {
        // hand the outputs to the next phase of this launch: neighbouring columns share a granule (even column = low half).
        // Only full tiles are published (n % 16 == 0 is a condition of the fused launcher).
        const uint32_t other = (uint32_t)__shfl_xor((int)y16bits, 1, 64);
        if (ep_col >= 0 && (ep_col & 1) == 0) {
            const unsigned long long gv = ((unsigned long long)ex.epoch << 32) | (unsigned long long)((uint32_t)y16bits | (other << 16));
            __hip_atomic_store((gu64*)(ex.gran + ((size_t)ep_m * p.ld_out + ep_col) / 2), gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- LDS carve --------------------------------------------------------------------------------------------------------------
struct EngGeom {
    int S;              // ring slots
    int planes_off, consts_off, red_off, scratch_off, flags_off;
    int total;
};
static inline EngGeom eng_geom(int gw_max, int m, int r_max, int slots_cap) {
    EngGeom g;
    const int planes = kCW * gw_max * 8 * 64 * m;
    const int consts = kCW * gw_max * 64;
    const int red = r_max * kCW * 64 * 4;
    const int scratch = 4 * kCW * 4;
    const int fixed = planes + consts + red + scratch + kFlagBytes;
    int S = (160 * 1024 - fixed) / kSlot;
    if (S > 32) S = 32;
    if (slots_cap > 0 && S > slots_cap) S = slots_cap;
    g.S = S;
    g.flags_off = S * kSlot;
    g.planes_off = g.flags_off + kFlagBytes;
    g.consts_off = g.planes_off + planes;
    g.red_off = g.consts_off + consts;
    g.scratch_off = g.red_off + red;
    g.total = g.scratch_off + scratch;
    return g;
}
__device__ __forceinline__ EngLds eng_lds(unsigned char* smem, const EngGeom& g) {
    EngLds L;
    const uint32_t base = lds_addr_of(smem);
    L.ring = base;
    L.fl = base + (uint32_t)g.flags_off;
    L.planes = base + (uint32_t)g.planes_off;
    L.consts = base + (uint32_t)g.consts_off;
    L.red = base + (uint32_t)g.red_off;
    L.scratch = base + (uint32_t)g.scratch_off;
    L.S = g.S;
    return L;
}

__device__ __forceinline__ void loader_prologue(const EngLds& L, int lane) {
    // the flag block is zeroed by the loader before the launch's one s_barrier; nobody reads it earlier
    if (lane < (int)(sizeof(EngFlags) / 4)) lds_st_v(L.fl + 4u * (uint32_t)lane, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ LoadPhase load_phase_of(const I8Params& p, int R, bool rope) {
    LoadPhase f;
    f.qw = reinterpret_cast<const unsigned char*>(p.qw);
    f.meta = reinterpret_cast<const unsigned char*>(p.meta);
    f.tile0 = rope ? (blockIdx.x / p.pair_stride) * 2 * p.pair_stride + blockIdx.x % p.pair_stride : blockIdx.x * R;
    f.tile_stride = rope ? p.pair_stride : 1;
    f.R = R;
    f.Gw = p.groups / kCW;
    f.groups = p.groups;
    f.tiles = p.tiles;
    return f;
}

// ---- kernels ----------------------------------------------------------------------------------------------------------------
template <int R, bool LONGK, bool ROPE, bool NORM, bool MERGE>
__global__ __launch_bounds__(kET, 1) void k_w4_engine(const I8Params p, const EngGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const EngLds L = eng_lds(smem, g);
    if (wave == kCW) {
        loader_prologue(L, lane);
        LoaderState st = {0u, 0, 0u, 0};
        const LoadPhase f = load_phase_of(p, R, ROPE);
        loader_phase(f, st, L, lane);
        loader_finish(st, L, lane);
        return;
    }
    ConsState cs = {0u, 0, 0u};
    const Exchange ex = {nullptr, 0u, nullptr};
    consume_phase<R, LONGK, ROPE, NORM, MERGE, XS_GLOBAL, false, true>(p, ex, L, cs, wave, lane);
}

// attention split merge + attn_out + residual  ->  RMSNorm + gate|up + silu.mul, one launch (see the header)
template <int R2>
__global__ __launch_bounds__(kET, 1) void k_w4_engine_o_gateup(const I8Params p1, const I8Params p2, const EngGeom g,
                                                                unsigned long long* gran, const uint32_t* epoch_ptr,
                                                                uint32_t epoch_add, uint32_t* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const EngLds L = eng_lds(smem, g);
    if (wave == kCW) {
        loader_prologue(L, lane);
        LoaderState st = {0u, 0, 0u, 0};
        const LoadPhase f1 = load_phase_of(p1, 1, false);
        loader_phase(f1, st, L, lane);
        const LoadPhase f2 = load_phase_of(p2, R2, false);
        loader_phase(f2, st, L, lane);
        loader_finish(st, L, lane);
        return;
    }
    ConsState cs = {0u, 0, 0u};
    Exchange ex;
    ex.gran = gran;
    ex.epoch = *epoch_ptr + epoch_add;
    ex.err = err;
    consume_phase<1, false, false, false, true, XS_GLOBAL, true, true>(p1, ex, L, cs, wave, lane);
    consume_phase<R2, false, false, true, false, XS_GRANULES, false, false>(p2, ex, L, cs, wave, lane);
    if (lane == 0 && wave == 0 && err && lds_poll(L.fl + kFlAbort)) atomicOr(err, 1u);
}

__global__ void k_engine_epoch_advance(uint32_t* epoch, uint32_t by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *epoch += by;
}

template <typename KT>
static int eng_set_lds(KT kern, int bytes) {
    if (bytes > 160 * 1024) return ZL_ELIMIT;
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return ZL_ELIMIT;
    }
    return ZL_OK;
}

template <int R, bool LONGK, bool ROPE, bool NORM, bool MERGE>
int launch_engine(const I8Params& p, int grid, int slots_cap, hipStream_t hs) {
    const EngGeom g = eng_geom(p.groups / kCW, p.m, R, slots_cap);
    if (g.S < kInFlight + 2) return ZL_ELIMIT;
    int st = eng_set_lds(&k_w4_engine<R, LONGK, ROPE, NORM, MERGE>, g.total);
    if (st) return st;
    hipLaunchKernelGGL((k_w4_engine<R, LONGK, ROPE, NORM, MERGE>), dim3(grid), dim3(kET), g.total, hs, p, g);
    return zl_launch_status();
}

I8Params eng_params(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                    const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups, int tiles,
                    int epilogue, int ld_out, const uint16_t* norm_w, float norm_eps) {
    I8Params p;
    p.x = x; p.ldx = ldx; p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes; p.meta_bytes = meta_bytes;
    p.bias = bias; p.residual = residual; p.y = y; p.m = m; p.n = n; p.k = k; p.groups = groups; p.tiles = tiles; p.epi = epilogue;
    p.ld_out = ld_out; p.norm_w = norm_w; p.norm_eps = norm_eps;
    p.cosv = p.sinv = nullptr; p.placement = p.buf_lens = nullptr; p.k_bufs = p.v_bufs = nullptr; p.q_out = nullptr;
    p.h = p.hkv = p.d = p.bshd = 0; p.pair_stride = 1;
    p.mg_part = nullptr; p.mg_stat = nullptr; p.mg_valid_lens = nullptr; p.mg_split_len = p.mg_max_splits = 0;
    return p;
}

}  // namespace

// what the engine covers: the integer-plane kernel's range with whole slots (K a multiple of 1024) and an LDS budget that
// leaves the loader a ring worth having
bool zl_w4_engine_covers(int64_t m, int64_t k, int r) {
    if (m < 1 || m > 4 || k < 1024 || k % 1024 != 0 || k > 16384 || (k > 4096 && m > 2)) return false;
    return eng_geom((int)(k / 1024), (int)m, r, 0).S >= kInFlight + 2;
}

// internal (called by zl_w4a16_gemm_mfma_ex under zl_w4_opts_t::small_algo == 2)
int zl_w4a16_gemm_engine(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                         uint32_t meta_bytes, const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k,
                         int groups, int tiles, int epilogue, int ld_out, const uint16_t* norm_w, float norm_eps, int slots_cap,
                         hipStream_t hs) {
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    int r = (tiles + cus - 1) / cus;
    if (r > 8) r = 8;
    if (!zl_w4_engine_covers(m, k, r)) return ZL_ESHAPE;
    const I8Params p = eng_params(x, ldx, qw, meta, qw_bytes, meta_bytes, bias, residual, y, m, n, k, groups, tiles, epilogue, ld_out,
                                  norm_w, norm_eps);
    const int grid = (tiles + r - 1) / r;
    const bool lk = groups > 4 * kCW;
#define ZL_ENG(RR)                                                                                           \
    case RR:                                                                                                 \
        if (norm_w) return lk ? ZL_ESHAPE : launch_engine<RR, false, false, true, false>(p, grid, slots_cap, hs); \
        return lk ? launch_engine<RR, true, false, false, false>(p, grid, slots_cap, hs)                     \
                  : launch_engine<RR, false, false, false, false>(p, grid, slots_cap, hs);
    switch (r) { ZL_ENG(1) ZL_ENG(2) ZL_ENG(3) ZL_ENG(4) ZL_ENG(5) ZL_ENG(6) ZL_ENG(7) ZL_ENG(8) }
#undef ZL_ENG
    return ZL_EINVAL;
}

int zl_w4a16_gemm_engine_rope(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                              uint32_t meta_bytes, const uint16_t* bias, int m, int n, int k, int groups, int tiles,
                              const uint16_t* norm_w, float norm_eps, const float* cosv, const float* sinv,
                              const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                              uint16_t* const* v_bufs, uint16_t* q_out, int h, int hkv, int d, int bshd, hipStream_t hs) {
    if (!zl_w4_engine_covers(m, k, 2) || k > 4096 || d % 32 != 0 || n != (h + 2 * hkv) * d || tiles * 16 != n) return ZL_ESHAPE;
    I8Params p = eng_params(x, ldx, qw, meta, qw_bytes, meta_bytes, bias, nullptr, nullptr, m, n, k, groups, tiles,
                            bias ? ZL_EPI_BIAS : 0, n, norm_w, norm_eps);
    p.cosv = cosv; p.sinv = sinv; p.placement = placement; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs;
    p.q_out = q_out; p.h = h; p.hkv = hkv; p.d = d; p.bshd = bshd; p.pair_stride = d / 32;
    const int grid = tiles / 2;
    return norm_w ? launch_engine<2, false, true, true, false>(p, grid, 0, hs) : launch_engine<2, false, true, false, false>(p, grid, 0, hs);
}

static I8Params eng_merge_params(const void* ws, const int32_t* buf_lens, const int32_t* valid_lens, int split_len, int max_splits,
                                 const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                                 const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups,
                                 int tiles, int epilogue) {
    I8Params p = eng_params(nullptr, 0, qw, meta, qw_bytes, meta_bytes, bias, residual, y, m, n, k, groups, tiles, epilogue, n, nullptr, 0.f);
    p.buf_lens = buf_lens;
    p.mg_part = reinterpret_cast<const uint16_t*>(ws);
    p.mg_stat = reinterpret_cast<const float*>(ws) + (size_t)m * groups * max_splits * 64;
    p.mg_valid_lens = valid_lens; p.mg_split_len = split_len; p.mg_max_splits = max_splits;
    return p;
}

int zl_w4a16_gemm_engine_merge(const void* ws, const int32_t* buf_lens, const int32_t* valid_lens, int split_len, int max_splits,
                               const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                               const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups,
                               int tiles, int epilogue, hipStream_t hs) {
    if (!zl_w4_engine_covers(m, k, 1) || k > 4096 || max_splits < 1 || max_splits > 16 || split_len < 1) return ZL_ESHAPE;
    if (epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) return ZL_ESHAPE;
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    if (tiles > cus) return ZL_ESHAPE;                     // one row tile per workgroup
    const I8Params p = eng_merge_params(ws, buf_lens, valid_lens, split_len, max_splits, qw, meta, qw_bytes, meta_bytes, bias,
                                        residual, y, m, n, k, groups, tiles, epilogue);
    return launch_engine<1, false, false, false, true>(p, tiles, 0, hs);
}

// The fused launch.  Geometry: the first projection has ONE row tile per workgroup (tiles1 = grid <= CUs, every workgroup
// resident at once: the hand-off is an all-to-all), the second R2 = tiles2 / grid tiles per workgroup.
int zl_w4_engine_o_gateup_launch(const void* ws, const int32_t* buf_lens, const int32_t* valid_lens, int split_len, int max_splits,
                                 const uint32_t* qw1, const uint32_t* meta1, uint32_t qw1_bytes, uint32_t meta1_bytes,
                                 const uint16_t* bias1, uint16_t* hidden, int m, int n1, int k1, int groups1, int tiles1,
                                 const uint32_t* qw2, const uint32_t* meta2, uint32_t qw2_bytes, uint32_t meta2_bytes,
                                 const uint16_t* bias2, const uint16_t* norm_w, float norm_eps, uint16_t* act, int n2, int groups2,
                                 int tiles2, int epilogue2, void* granules, const uint32_t* epoch_ptr, uint32_t epoch_add,
                                 uint32_t* err, hipStream_t hs) {
    int cus = zl_device_cu_count();
    if (cus <= 0) return ZL_ELIMIT;
    if (!zl_w4_engine_covers(m, k1, 1) || k1 > 4096 || n1 > 4096 || n1 % 1024 != 0 || max_splits < 1 || max_splits > 16 || split_len < 1)
        return ZL_ESHAPE;
    if (tiles1 > cus || tiles2 % tiles1 != 0) return ZL_ESHAPE;
    const int r2 = tiles2 / tiles1;
    if (r2 < 1 || r2 > 8 || !(epilogue2 & ZL_EPI_SILU_MUL)) return ZL_ESHAPE;
    I8Params p1 = eng_merge_params(ws, buf_lens, valid_lens, split_len, max_splits, qw1, meta1, qw1_bytes, meta1_bytes, bias1,
                                   hidden, hidden, m, n1, k1, groups1, tiles1, ZL_EPI_RESIDUAL | (bias1 ? ZL_EPI_BIAS : 0));
    I8Params p2 = eng_params(nullptr, 0, qw2, meta2, qw2_bytes, meta2_bytes, bias2, nullptr, act, m, n2, n1, groups2, tiles2,
                             epilogue2, n2 / 2, norm_w, norm_eps);
    const int gw = (groups1 > groups2 ? groups1 : groups2) / kCW;
    const EngGeom g = eng_geom(gw, m, r2, 0);
    if (g.S < kInFlight + 2) return ZL_ELIMIT;
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(granules);
#define ZL_FUSE(RR)                                                                                                  \
    case RR: {                                                                                                       \
        int st = eng_set_lds(&k_w4_engine_o_gateup<RR>, g.total);                                                    \
        if (st) return st;                                                                                           \
        hipLaunchKernelGGL((k_w4_engine_o_gateup<RR>), dim3(tiles1), dim3(kET), g.total, hs, p1, p2, g, gran, epoch_ptr, \
                           epoch_add, err);                                                                          \
        return zl_launch_status();                                                                                   \
    }
    switch (r2) { ZL_FUSE(1) ZL_FUSE(2) ZL_FUSE(3) ZL_FUSE(4) ZL_FUSE(5) ZL_FUSE(6) ZL_FUSE(7) ZL_FUSE(8) }
#undef ZL_FUSE
    return ZL_EINVAL;
}

int zl_engine_epoch_advance_launch(uint32_t* epoch, uint32_t by, hipStream_t hs) {
    hipLaunchKernelGGL(k_engine_epoch_advance, dim3(1), dim3(64), 0, hs, epoch, by);
    return zl_launch_status();
}

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
//   ring      slots of 9 KiB (8 items + their 8 x 16 meta words + padding: every DMA destination 1 KiB aligned); slot = one row
//             tile x eight consecutive 128-k groups, consumer wave w takes item w.  The ring is cut into NB ROUNDS of D slots
//             (D = the register ring of a consumer wave = R tiles x XD groups, NB >= 2: 14 slots = 2 x 7 for gate|up at one row).
//   landed    one LDS word, the number of slots whose data is in LDS: at most 7 slots sit between "issued" and "counted as
//             landed" (nine DMA instructions per slot, vmcnt counts to 63); whenever the loader cannot issue (seven in flight, or
//             the next round of the ring not handed back) it waits for the OLDEST slot with a counted s_waitcnt vmcnt and publishes it.
//   rounds    a consumer wave looks at `landed` once per round, copies its D items from STATIC ring addresses into registers,
//             hands the round back with one LDS add (8 adds = free) and then folds the D items with no bookkeeping in between:
//             the per-item chain (nibble expansion, two dependent MFMAs, five fp32 ops) leaves no room -- with a poll, an
//             address computation and a release PER ITEM the consumers alone needed 10.5 us for gate|up's 28 slots (r22).
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
//
// Measured (round 4, profiles/r04_engine_ab.txt, r04_engine_timeline.txt, r04_ldsdma_stream.txt): bit-identical to k_w4a16_i8p
// and NOT faster -- which is why it is an opt-in route (zl_w4_opts_t::small_algo = 2; LLaMA: ZL_W4_SMALL_ALGO=2,
// ZL_FUSE_O_GATEUP=1) and k_w4a16_i8p stays the default.  Stand-alone launches: 43.1 us per layer in the step against 39.2-39.8;
// with the fused launch on k_w4a16_i8p's other three launches: 39.8-39.9 (a tie).  Why the guide's engine row does not
// transfer to this layer: (1) an LDS-DMA costs the loader wave ~85 cycles of issue per KiB (the texture-address path, not
// HBM): a bare loader stream reaches 6.0 TB/s on 224 KiB per workgroup (register ring: 5.4-5.5), but every poll, counted wait
// and round hand-back the loader does between DMAs comes straight out of that -- inside the engine a slot takes 0.45 us
// instead of 0.30; (2) the consumers' per-item chain (expand, two dependent MFMAs, five fp32 ops: 0.25-0.3 us per slot and
// wave) has no slack to hide ring bookkeeping, so the ring is consumed in rounds and a round's tail waits for its slowest
// wave; (3) the fused hand-off costs what the boundary it removes costs: granule publish -> all 256 workgroups have swept
// them = 2.0-2.5 us, plus 1.6 us of RMSNorm + plane conversion that cannot start earlier, while the 14-slot ring (126 of
// 160 KiB of LDS) is full after 9.5 us and the loader idles until the first round is handed back.
#include "zl_common.h"
#include "w4_i8p_common.h"

namespace {

constexpr int kCW = 8;                     // consumer waves
constexpr int kET = (kCW + 1) * 64;        // threads: consumers + the loader wave
constexpr int kSlot = 8 * 1024 + 1024;     // bytes per ring slot: 8 items + their 8 x 16 meta words (512 B) + 512 B of padding -- every DMA destination stays 1 KiB aligned
constexpr int kInFlight = 7;               // slots between issued and known-landed (9 DMAs each; vmcnt counts to 63)
constexpr int kFlagBytes = 256;
constexpr uint32_t kSpinLimit = 1u << 21;  // bounded polls (~0.1 s): a lost hand-off ends in wrong numbers + an error word, never a hang

// ---- optional timeline probe (build with -DZL_ENG_PROBE; tools/ubench/probe_engine.py): wall-clock stamps (100 MHz),
//      [workgroup][wave < 10][2][68]: row 0 = {entry, past the barrier, activations landed, staged, slot stamps A...},
//      row 1 = slot stamps B.  Loader: A[s] = slot s issued, B[j] = its j-th slot published; consumer: A[s] = slot s fetched,
//      B[s] = slot s folded in.
#ifdef ZL_ENG_PROBE
__device__ unsigned long long* zl_probe_eng = nullptr;
// (the pointer is read ONCE per wave, in eng_lds: a load per stamp costs ~0.8 us under a saturated memory pipe)
#define ZL_EPROBE(wave, lane, row, idx)                                                                          \
    do {                                                                                                         \
        if (L.probe && (lane) == 0 && (idx) < 68)                                                                \
            L.probe[(((size_t)blockIdx.x * 10 + (wave)) * 2 + (row)) * 68 + (idx)] = wall_clock64();             \
    } while (0)
#else
#define ZL_EPROBE(wave, lane, row, idx) do {} while (0)
#endif

// LDS flag block (behind the ring)
struct EngFlags {
    uint32_t landed;        // slots landed
    uint32_t pad0;
    uint32_t bar;           // consumer rendezvous counter
    uint32_t abort_;        // some poll ran out
    uint32_t consumed[32];  // per ring ROUND: +1 per consumer wave and use
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
    int S;             // ring slots = NB * D
    int D, NB;         // slots per round, rounds in the ring
    unsigned long long* probe;   // timeline stamps (ZL_ENG_PROBE builds), else null
};
#define ZL_LDS(T) __attribute__((address_space(3))) T
template <typename T> __device__ __forceinline__ T lds_ld(uint32_t a) { return *(const ZL_LDS(T)*)(uintptr_t)a; }
template <typename T> __device__ __forceinline__ void lds_st(uint32_t a, T v) { *(ZL_LDS(T)*)(uintptr_t)a = v; }
// a look at a flag word: every lane reads the same word, and the value is handed back as a SCALAR -- a branch on the raw
// VGPR is a divergent branch to the compiler, which then keeps the loader's / consumer's whole state machine (slot counters,
// the vmcnt switch) in VGPRs under exec masks: ~100 extra instructions per slot on the one wave that bounds the stream
__device__ __forceinline__ uint32_t lds_poll(uint32_t a) {
    return __builtin_amdgcn_readfirstlane(*(const volatile ZL_LDS(uint32_t)*)(uintptr_t)a);
}
__device__ __forceinline__ void lds_st_v(uint32_t a, uint32_t v) { *(volatile ZL_LDS(uint32_t)*)(uintptr_t)a = v; }
__device__ __forceinline__ void lds_inc(uint32_t a) {
    __hip_atomic_fetch_add((ZL_LDS(uint32_t)*)(uintptr_t)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
constexpr uint32_t kFlLanded = 0, kFlBar = 8, kFlAbort = 12, kFlConsumed = 16;   // offsets inside EngFlags
// (fused launches; rings have at most 8 rounds, the upper counters are free) consumer waves that have their first phase's inputs in
// registers / that have started / finished gathering a hand-off
constexpr uint32_t kFlStaged = kFlConsumed + 4 * 29, kFlGatherIn = kFlConsumed + 4 * 30, kFlGatherOut = kFlConsumed + 4 * 31;

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// LDS-DMA, non-temporal: one instruction moves 1 KiB (64 lanes x 16 B, lane-linear destination = M0 + instruction offset + 16 lane).
// The instruction offset moves the global source AND the LDS destination (tools/ubench/ldsdma_probe.hip), so FOUR consecutive
// KiB go out behind one M0 write: scalar base, one 32-bit lane offset, offsets 0 / 1 / 2 / 3 KiB -- a third of the issue slots of
// four (M0 save, M0 write, 64-bit address add, DMA, M0 restore) statements, and the loader's issue rate is what bounds the
// stream (r04 timeline: 0.4 us per 8.5 KiB slot with single-DMA statements).  M0 is compiler-reserved: saved and restored
// inside the statement.  Invisible to hipcc's s_waitcnt bookkeeping: the loader counts its own vmcnt.
__device__ __forceinline__ const void* uniform_ptr(const void* p) {      // in SGPRs whatever the compiler could prove
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (const void*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void dma4k_nt(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 nt\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma1k_nt(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// ---- loader ---------------------------------------------------------------------------------------------------------------
struct LoadPhase {
    const unsigned char* qw;
    const unsigned char* meta;
    int tile0, tile_stride, R, Gw, groups, tiles;
    int thin;            // this phase's stream runs beside a hand-off sweep of the consumers
};
struct LoaderState {
    uint32_t s;         // slots issued so far (all phases)
    uint32_t round_no;  // round being filled (all phases; a phase starts on a round boundary)
    int sir;            // slot inside the round
    int pending;        // issued, not yet counted as landed
};

// count the OLDEST slot in flight as landed: nine DMAs per slot, vmcnt retires in order
__device__ __forceinline__ void publish_oldest(LoaderState& st, const EngLds& L, int lane) {
    switch (st.pending) {
        case 7: wait_vmcnt<54>(); break;
        case 6: wait_vmcnt<45>(); break;
        case 5: wait_vmcnt<36>(); break;
        case 4: wait_vmcnt<27>(); break;
        case 3: wait_vmcnt<18>(); break;
        case 2: wait_vmcnt<9>(); break;
        default: wait_vmcnt<0>(); break;
    }
    --st.pending;
    if (lane == 0) lds_st_v(L.fl + kFlLanded, st.s - (uint32_t)st.pending);
    ZL_EPROBE(kCW, lane, 1, 4 + (int)(st.s - (uint32_t)st.pending) - 1);
}

__device__ __forceinline__ void loader_phase(const LoadPhase& f, LoaderState& st, const EngLds& L, int lane) {
    const uint32_t ring0 = L.ring;
    const uint32_t voff = (uint32_t)lane * 16u;
    if (st.sir != 0) { st.sir = 0; ++st.round_no; }                    // a phase starts on a round boundary (consumers agree)
    for (int gi = 0; gi < f.Gw; ++gi) {
        for (int r = 0; r < f.R; ++r) {
            // at most kInFlight slots between issued and counted (63 DMAs: the width of vmcnt); whenever the loader cannot
            // issue it counts: a slot is published as soon as it has landed, never later than the next issue
            while (st.pending >= kInFlight) publish_oldest(st, L, lane);
            if (f.thin) {
                // a consumer wave of this CU is sweeping a hand-off: its sc1 loads queue behind whatever the loader has in the
                // CU's memory pipe (a sweep took 2-3 us behind seven slots, r29) -- one slot in flight until the sweep is over
                while (st.pending >= 1 && lds_poll(L.fl + kFlGatherIn) != lds_poll(L.fl + kFlGatherOut)) publish_oldest(st, L, lane);
            }
            const uint32_t half = st.round_no % (uint32_t)L.NB;
            if (st.sir == 0 && st.round_no >= (uint32_t)L.NB) {
                // a new round of the ring: handed back when all eight consumer waves have copied its previous content
                const uint32_t need = (uint32_t)kCW * (st.round_no / (uint32_t)L.NB);
                uint32_t spins = 0;
                while (lds_poll(L.fl + kFlConsumed + 4u * half) < need) {
                    if (st.pending > 0) publish_oldest(st, L, lane);          // (the consumers cannot free what they were not told about)
                    else {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > kSpinLimit || ((spins & 255u) == 0u && lds_poll(L.fl + kFlAbort))) { if (lane == 0) lds_st_v(L.fl + kFlAbort, 1u); break; }
                    }
                }
            }
            int tile = f.tile0 + r * f.tile_stride;
            if (tile >= f.tiles) tile = f.tiles - 1;          // a tile past the end: the slot is still counted, its content unused
            const size_t item0 = (size_t)tile * (size_t)f.groups + (size_t)gi * 8;
            const unsigned char* src = (const unsigned char*)uniform_ptr(f.qw + item0 * 1024);   // scalar base of the DMAs
            const uint32_t dst = __builtin_amdgcn_readfirstlane(ring0 + (half * (uint32_t)L.D + (uint32_t)st.sir) * (uint32_t)kSlot);
            dma4k_nt(src, voff, dst);
            dma4k_nt(src + 4096, voff, dst + 4096u);
            // the eight items' meta words: 512 contiguous bytes = half a DMA (lanes 0..31; an inactive lane moves nothing)
            const void* msrc = uniform_ptr(f.meta + item0 * 64);
            if (lane < 32) dma1k_nt(msrc, voff, dst + 8192u);
            ZL_EPROBE(kCW, lane, 0, 4 + (int)st.s);
            ++st.s;
            ++st.pending;
            if (++st.sir == L.D) { st.sir = 0; ++st.round_no; }
        }
    }
}
__device__ __forceinline__ void loader_finish(LoaderState& st, const EngLds& L, int lane) {
    while (st.pending > 0) publish_oldest(st, L, lane);
}

// ---- consumers ------------------------------------------------------------------------------------------------------------
struct ConsState {
    uint32_t taken;      // slots taken so far (all phases) -- the loader's `landed` counts the same slots
    uint32_t seen;       // landed count last read: the flag is looked at again only when a round runs past it
    uint32_t round_no;   // round being read (all phases)
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

// wait until `upto` slots of the launch have landed
__device__ __forceinline__ void wait_landed(ConsState& cs, const EngLds& L, uint32_t upto, int lane) {
    if (cs.seen >= upto) return;
    uint32_t spins = 0, v;
    while ((v = lds_poll(L.fl + kFlLanded)) < upto) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit || ((spins & 255u) == 0u && lds_poll(L.fl + kFlAbort))) { if (lane == 0) lds_st_v(L.fl + kFlAbort, 1u); break; }
    }
    cs.seen = v;
    asm volatile("" ::: "memory");
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
template <int R, bool LONGK, bool ROPE, bool NORM, bool MERGE, int XS, bool PUB, bool FIRST, int D>
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

    // MERGE: the launch's s_barrier comes FIRST -- the 32 record loads per lane of the split merge keep the CU's address unit busy
    // for ~2 us (135 KB per workgroup), and a loader held back behind them starts the ring at 2.5 us (r27 timeline); the records'
    // first touch is latency the ring can use
    if constexpr (FIRST && MERGE) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
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
    // ---- the launch's one s_barrier: the activation requests are in the CU's memory queue, the loader may start the ring
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FIRST) ZL_EPROBE(wave, lane, 0, 0);
    if constexpr (FIRST && !MERGE) __builtin_amdgcn_s_barrier();
    if constexpr (FIRST) ZL_EPROBE(wave, lane, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue operands, requested now (behind the first ring slots: they are needed at the very end).  Branch-free and a
    //      FIXED number of loads (kEpiLoads): the wait for the activations below is a counted vmcnt that leaves exactly these
    //      in flight.  Operands that do not apply read through an empty descriptor / past its end and come back as zero.
    const uint32_t kOob = 0x7ffffff0u;
    constexpr int kEpiLoads = ROPE ? 10 : 4;
    uint32_t rp_c0r = 0, rp_s0r = 0, rp_c1r = 0, rp_s1r = 0, rp_placer = 0, rp_blenr = 0, rp_b0r = 0, rp_b1r = 0;
    unsigned long long rp_kptr = 0, rp_vptr = 0;
    bool rp_kv_row = false;
    if constexpr (ROPE) {
        const int m = tid >> 4, n0 = tile0 * 16 + (tid & 15);
        const int head = n0 / p.d, dcol = n0 % p.d, half = p.d / 2;
        const bool mine = tid < 16 * M;
        const bool rot = mine && head < p.h + p.hkv;
        rp_kv_row = mine && head >= p.h;
        const uint32_t tab_bytes = (uint32_t)M * (uint32_t)p.d * 4u;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.cosv), 0, tab_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.sinv), 0, tab_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rpl = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.placement), 0, (uint32_t)M * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rbl = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.buf_lens), 0, (uint32_t)M * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rkb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t**>(p.k_bufs), 0, (uint32_t)M * 8u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rvb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t**>(p.v_bufs), 0, (uint32_t)M * 8u, 0x00020000);
        const bool has_bias = (p.epi & ZL_EPI_BIAS) && p.bias;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.bias), 0, has_bias ? (uint32_t)p.n * 2u : 0u, 0x00020000);
        const uint32_t toff = rot ? ((uint32_t)m * (uint32_t)p.d + (uint32_t)dcol) * 4u : kOob;
        rp_c0r = __builtin_amdgcn_raw_buffer_load_b32(rc, toff, 0, 0);
        rp_s0r = __builtin_amdgcn_raw_buffer_load_b32(rsn, toff, 0, 0);
        rp_c1r = __builtin_amdgcn_raw_buffer_load_b32(rc, toff + (uint32_t)half * 4u, 0, 0);
        rp_s1r = __builtin_amdgcn_raw_buffer_load_b32(rsn, toff + (uint32_t)half * 4u, 0, 0);
        rp_placer = __builtin_amdgcn_raw_buffer_load_b32(rpl, rp_kv_row ? (uint32_t)m * 4u : kOob, 0, 0);
        rp_blenr = __builtin_amdgcn_raw_buffer_load_b32(rbl, rp_kv_row ? (uint32_t)m * 4u : kOob, 0, 0);
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const u2 kp = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(rkb, rp_kv_row && head < p.h + p.hkv ? (uint32_t)m * 8u : kOob, 0, 0));
        const u2 vp = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(rvb, rp_kv_row && head >= p.h + p.hkv ? (uint32_t)m * 8u : kOob, 0, 0));
        rp_kptr = __builtin_bit_cast(unsigned long long, kp);
        rp_vptr = __builtin_bit_cast(unsigned long long, vp);
        rp_b0r = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rb, mine ? (uint32_t)n0 * 2u : kOob, 0, 0);
        rp_b1r = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rb, mine ? (uint32_t)(n0 + half) * 2u : kOob, 0, 0);
    }
    const bool silu = (p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
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
    if constexpr (XS == XS_GRANULES) {
        if (lane == 0) lds_inc(L.fl + kFlGatherIn);
        // the rows arrive from the previous phase of THIS launch: 8 halves per lane and slot = 4 granules, swept until every
        // tag carries the epoch (sc1 loads: another CU's write-through store is seen at the next sweep)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int row = LONGK ? (s & 1) : s, c = LONGK ? (s >> 1) : 0;
            const int g = wave + kCW * (4 * c + lgi);
            xr[s] = make_uint4(0, 0, 0, 0);
            if (row < M && 4 * c < Gw) {
                const gu64* src = (const gu64*)(ex.gran + ((size_t)row * K + (size_t)(g < groups ? g : 0) * 128 + uo * 8) / 2);
                uint32_t v[4] = {0u, 0u, 0u, 0u};
                bool have[4] = {false, false, false, false};
                uint32_t spins = 0;
                for (;;) {
                    // only the granules still missing are read again, and the sweeps are ~0.2 us apart: eight waves sweeping
                    // without a pause next to the loader stretched its slots from 3 to 4.6 us (r28 timeline; the guide's
                    // polling-cost row)
                    bool ok = true;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!have[e]) {
                            const unsigned long long t = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            have[e] = (uint32_t)(t >> 32) == ex.epoch;
                            if (have[e]) v[e] = (uint32_t)t;
                        }
                        ok &= have[e];
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
        if (lane == 0) lds_inc(L.fl + kFlGatherOut);
    } else {
        // the activations (and norm weights) have landed; the kEpiLoads younger requests stay in flight (a builtin wait: hipcc
        // models it, see w4_i8p.hip).  MERGE: the merge arithmetic above already consumed its records
        if constexpr (!MERGE) __builtin_amdgcn_s_waitcnt(0x0F70 | kEpiLoads);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FIRST && MERGE && PUB) {                     // fused launch: the loader may go on to the next projection's weights
        asm volatile("" :: "v"(xr[0].x), "v"(xr[0].y), "v"(xr[0].z), "v"(xr[0].w));     // (after the merged row exists)
        if (lane == 0) lds_inc(L.fl + kFlStaged);
    }
    ZL_EPROBE(wave, lane, FIRST ? 0 : 1, 2);

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
    ZL_EPROBE(wave, lane, FIRST ? 0 : 1, 3);

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
    // rounds of D = R x XD slots: one look at the landed count, D item reads from static ring addresses, one release, then
    // D folds with nothing in between (see the header)
    constexpr int XD = D / R;
    static_assert(D % R == 0 && XD >= 1, "a round holds whole groups of R tiles");
    for (int gi0 = 0; gi0 < Gw; gi0 += XD) {
        const int n = min(XD, Gw - gi0) * R;                           // slots of this round
        const uint32_t base = cs.taken;
        const uint32_t roff = (cs.round_no % (uint32_t)L.NB) * (uint32_t)(D * kSlot);
        const uint32_t rbase = L.ring + roff + (uint32_t)(wave * 1024 + lane * 16);
        const uint32_t mbase = L.ring + roff + 8192u + (uint32_t)(wave * 64 + (lane & 15) * 4);
        uint4 wq[D];
        uint32_t mt[D];
        auto read_item = [&](int q) {                                  // q static
            wait_landed(cs, L, base + (uint32_t)q + 1u, lane);         // a scalar compare unless the consumers have caught up
            const u32x4 wv = lds_ld<u32x4>(rbase + (uint32_t)(q * kSlot));
            wq[q] = make_uint4(wv.x, wv.y, wv.z, wv.w);
            mt[q] = lds_ld<uint32_t>(mbase + (uint32_t)(q * kSlot));
        };
        read_item(0);
        v4i a0 = zero4, a1 = zero4;
        f32x4 cst = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < D; ++q) {
            if (q >= n) break;
            constexpr int dummy = 0; (void)dummy;
            const int j = q / R, r = q % R, gi = gi0 + j;
            if (q + 1 < D && q + 1 < n) read_item(q + 1);              // one item ahead of the arithmetic
            if (q + 1 == n) {                                          // every item of the round is on its way to registers: hand the round back
                asm volatile("" ::: "memory");                         // (the release stays behind the reads; the LDS executes a wave's operations in order)
                if (lane == 0) lds_inc(L.fl + kFlConsumed + 4u * (cs.round_no % (uint32_t)L.NB));
            }
            if (r == 0) {
                a0 = lds_ld<v4i>(a_base + (uint32_t)((gi * 2 + 0) * 4 * rec));
                a1 = lds_ld<v4i>(a_base + (uint32_t)((gi * 2 + 1) * 4 * rec));
                cst = lds_ld<f32x4>(consts + (uint32_t)((gi * 4 + crow) * 16));
            }
            const uint4 w = wq[q];
            const uint32_t mw = mt[q];
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
            ZL_EPROBE(wave, lane, 1, 4 + (int)base + q);
        }
        cs.taken = base + (uint32_t)n;
        ++cs.round_no;
    }

    // ---- reduce over the 8 waves in fixed order, epilogue
#pragma unroll
    for (int r = 0; r < R; ++r) lds_st<float>(red + 4u * (uint32_t)((r * kCW + wave) * 64 + lane), acc[r]);
    consumer_barrier(cs, L, lane);
    // the epilogue operands become visible to the compiler only HERE (volatile statements keep their place): their first use --
    // and with it the wait for the loads, which sit behind the first ring slots in the CU's queue -- cannot be scheduled earlier
    asm volatile("" : "+v"(ep_b0r), "+v"(ep_b1r), "+v"(ep_resr), "+v"(ep_prevr));
    asm volatile("" : "+v"(rp_c0r), "+v"(rp_s0r), "+v"(rp_c1r), "+v"(rp_s1r), "+v"(rp_placer), "+v"(rp_blenr), "+v"(rp_b0r), "+v"(rp_b1r));
    asm volatile("" : "+v"(rp_kptr), "+v"(rp_vptr));
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
            const int n0 = tile0 * 16 + n_local;                       // columns n0 and n0 + half of the fused qkv row
            if ((p.epi & ZL_EPI_BIAS) && p.bias) {
                v0 += (float)__builtin_bit_cast(_Float16, (uint16_t)rp_b0r);
                v1 += (float)__builtin_bit_cast(_Float16, (uint16_t)rp_b1r);
            }
            const float rp_c0 = __builtin_bit_cast(float, rp_c0r), rp_s0 = __builtin_bit_cast(float, rp_s0r);
            const float rp_c1 = __builtin_bit_cast(float, rp_c1r), rp_s1 = __builtin_bit_cast(float, rp_s1r);
            const int rp_place = rp_kv_row ? (int)rp_placer : -1, rp_blen = (int)rp_blenr;
            uint16_t* rp_kv = reinterpret_cast<uint16_t*>(rp_kptr | rp_vptr);
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
        ZL_EPROBE(wave, lane, 1, 0);
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
    int S;              // ring slots = NB * D
    int D, NB;          // slots per round, rounds in the ring
    int planes_off, consts_off, red_off, scratch_off, flags_off;
    int total;
};
// slots per round of a projection with R row tiles per workgroup: the consumer's register ring (R tiles x XD groups)
constexpr int eng_round(int r) { return r >= 3 ? r : 4; }

static inline EngGeom eng_geom(int gw_max, int m, int r_max, int d, int rounds_cap) {
    EngGeom g;
    const int planes = kCW * gw_max * 8 * 64 * m;
    const int consts = kCW * gw_max * 64;
    const int red = r_max * kCW * 64 * 4;
    const int scratch = 4 * kCW * 4;
    const int fixed = planes + consts + red + scratch + kFlagBytes;
    int nb = (160 * 1024 - fixed) / (kSlot * d);
    if (nb > 8) nb = 8;
    if (rounds_cap > 0 && nb > rounds_cap) nb = rounds_cap;
    g.D = d;
    g.NB = nb;
    g.S = nb * d;
    g.flags_off = g.S * kSlot;
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
    L.D = g.D;
    L.NB = g.NB;
    L.probe = nullptr;
#ifdef ZL_ENG_PROBE
    L.probe = blockIdx.x < 256 ? zl_probe_eng : nullptr;
#endif
    return L;
}

__device__ __forceinline__ void loader_prologue(const EngLds& L, int lane, bool first) {
    // the flag block is zeroed by the (first) loader before the launch's one s_barrier; nobody reads it earlier
    if (first && lane < (int)(sizeof(EngFlags) / 4)) lds_st_v(L.fl + 4u * (uint32_t)lane, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ZL_EPROBE(first ? kCW : kCW + 1, lane, 0, 0);
    __builtin_amdgcn_s_barrier();
    ZL_EPROBE(first ? kCW : kCW + 1, lane, 0, 1);
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
    f.thin = 0;
    return f;
}

// ---- kernels ----------------------------------------------------------------------------------------------------------------
template <int R, bool LONGK, bool ROPE, bool NORM, bool MERGE>
__global__ __launch_bounds__(kET, 1) void k_w4_engine(const I8Params p, const EngGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const EngLds L = eng_lds(smem, g);
    if (wave >= kCW) {
        loader_prologue(L, lane, true);
        LoaderState st = {0u, 0u, 0, 0};
        const LoadPhase f = load_phase_of(p, R, ROPE);
        loader_phase(f, st, L, lane);
        loader_finish(st, L, lane);
        return;
    }
    ConsState cs = {0u, 0u, 0u, 0u};
    const Exchange ex = {nullptr, 0u, nullptr};
    consume_phase<R, LONGK, ROPE, NORM, MERGE, XS_GLOBAL, false, true, eng_round(R)>(p, ex, L, cs, wave, lane);
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
    if (wave >= kCW) {
        loader_prologue(L, lane, true);
        LoaderState st = {0u, 0u, 0, 0};
        const LoadPhase f1 = load_phase_of(p1, 1, false);
        loader_phase(f1, st, L, lane);
        // Two schedules of the second projection's stream were measured (profiles/r04_engine_ab.txt): started at once and running
        // at full depth beside the hand-off sweep (kFusedHoldBack = false: 39.8 us per layer in the step, run r28), or held back
        // until the first projection's inputs are in registers and thinned to one slot in flight while a consumer wave sweeps
        // the granules (true: 43.4-43.5, runs r29 / r30 -- the loader's own polls cost more than the sweep gains).
        constexpr bool kFusedHoldBack = false;
        if (kFusedHoldBack) {
            uint32_t spins = 0;
            while (lds_poll(L.fl + kFlStaged) < (uint32_t)kCW) {
                if (st.pending > 0) publish_oldest(st, L, lane);
                else {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kSpinLimit) break;
                }
            }
        }
        LoadPhase f2 = load_phase_of(p2, R2, false);
        f2.thin = kFusedHoldBack ? 1 : 0;
        loader_phase(f2, st, L, lane);
        loader_finish(st, L, lane);
        return;
    }
    ConsState cs = {0u, 0u, 0u, 0u};
    Exchange ex;
    ex.gran = gran;
    ex.epoch = *epoch_ptr + epoch_add;
    ex.err = err;
    consume_phase<1, false, false, false, true, XS_GLOBAL, true, true, eng_round(R2)>(p1, ex, L, cs, wave, lane);
    consume_phase<R2, false, false, true, false, XS_GRANULES, false, false, eng_round(R2)>(p2, ex, L, cs, wave, lane);
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

// experiment knob (tools/bench_engine.py; not part of the ABI): cap on the rounds of the ring
static int g_dbg_rounds_cap = 0;

template <int R, bool LONGK, bool ROPE, bool NORM, bool MERGE>
int launch_engine(const I8Params& p, int grid, int rounds_cap, hipStream_t hs) {
    const EngGeom g = eng_geom(p.groups / kCW, p.m, R, eng_round(R), rounds_cap > 0 ? rounds_cap : g_dbg_rounds_cap);
    if (g.NB < 2) return ZL_ELIMIT;
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
    return eng_geom((int)(k / 1024), (int)m, r, eng_round(r), 0).NB >= 2;
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
    const EngGeom g = eng_geom(gw, m, r2, eng_round(r2), g_dbg_rounds_cap);
    if (g.NB < 2) return ZL_ELIMIT;
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

#ifdef ZL_ENG_PROBE
extern "C" int zl_debug_set_probe_engine(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(zl_probe_eng), &p, sizeof(p)); }
#endif

extern "C" void zl_debug_engine_knobs(int rounds_cap) { g_dbg_rounds_cap = rounds_cap; }

int zl_engine_epoch_advance_launch(uint32_t* epoch, uint32_t by, hipStream_t hs) {
    hipLaunchKernelGGL(k_engine_epoch_advance, dim3(1), dim3(64), 0, hs, epoch, by);
    return zl_launch_status();
}

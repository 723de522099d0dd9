// w4_slab.hip -- W4A16 GEMM for decode batches of 5..32 rows (round 6): 2-D tiles with an optional K split, activations by
// LDS-DMA into wave-private fragment stores.
//
// Why (profiles/r06_rows_ablation.txt, profiles/r06_phase_timeline_rows.txt): the phase kernel (w4_phase.hip) gives a workgroup
// 16 R output columns and the WHOLE K and feeds the activations through ONE LDS phase buffer behind a workgroup barrier per
// 1024 k.  With few tiles per workgroup (attn_out, down: R = 1) a phase is ONE weight item per wave, so the launch is a chain
// of barrier + stage + barrier round trips: ~1 us per phase (attn_out at 32 rows: 4 phases in 4.3 us, 10.3 us per launch for
// 8.7 MB; down: 20 us for 30.5 MB).  With the staging and the barriers ablated the four projections of a Llama-3-8B layer take
// 44.9 us instead of 61.3 at 32 rows; with the arithmetic ablated instead, still 49.1: the batch pays per ROW for how the
// activations reach the matrix cores, not for FLOPs and not for weight bytes.
//
// Here no two waves share an activation fragment, so nothing needs a workgroup barrier until the partial tiles meet:
//   * a workgroup owns R row tiles (16 R output columns; R = 1 / 2 / 4 / 8) x one K slice of NW x GPW 128-k groups (KS slices
//     cover K); wave w owns GPW consecutive groups for ALL R tiles;
//   * the activations of a group (16 MB rows x 256 B) go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no registers,
//     no ds_write, asynchronous), row-contiguous and therefore coalesced (a first cut loaded the MFMA fragments straight from
//     global memory: 16 bytes out of 16 different cache lines per 16 lanes -- 64 L1 tag look-ups per instruction, and every
//     launch 5..8 us SLOWER than the phase kernel, profiles/r06_slab_v1_sweep.txt).  The 16-byte pieces of a row are stored
//     XOR-swizzled by the row (piece q of row r at position q ^ (r & 15): the lane picks the global piece that belongs at ITS
//     LDS position), so the fragment reads (ds_read_b128, lane = (row, k quarter)) touch every bank exactly once;
//   * a wave reads a group's fragments ONCE into registers (16 MB VGPRs) and keeps them for the group's R weight items -- an
//     R-th of the phase kernel's LDS reads; the region is wave-private (no barrier), two groups deep (the DMA of group j + 2 is
//     issued when group j has been read), and waited for with counted vmcnt: the DMAs are older than every weight load that
//     may still be in flight when they are needed;
//   * the weight stream is w4_phase.hip's: 1 KiB ZLW4M items + one meta word through a ring of up to 8 non-temporal buffer
//     loads, exact (q - z) on the VALU (0x6400 trick), fp32 group sums on the matrix cores, one v_fma_mix_f32 with the group's
//     scale per C register -- bit for bit the per-(column, group) arithmetic of k_w4a16_phase / k_w4a16_mfma;
//   * the NW waves' partial tiles meet in LDS in wave order; with a K split (long K: the down projection) the workgroup then
//     writes its fp32 slab write-through (16-byte sc1 stores, drained with vmcnt(0)), draws a ticket on its tile group's counter,
//     and the LAST arriver adds the KS slabs in split order (sc1 loads) and runs the epilogue -- the guide's publish-large /
//     splitk-seam form.  Everything is summed in a fixed order: two runs return the same bits.
// NORM instantiations (the ROW STATISTICS HAND-OFF, zl_w4_opts_t::row_ss / row_ss_out): the launches that write the residual stream
// (attn_out, down: plain epilogues of this kernel) leave per row and 16-column tile the sum of squares of the fp16 values they
// stored (16-lane DPP sum, zl_sum16); a launch that normalises that stream (qkv, gate|up) reads the row's K / 16 tile sums --
// 1 KB per row instead of the row -- forms rs = rsqrt_rn(sum / K + eps) once per workgroup (a 16-lane group per row, one barrier)
// and applies T(x rs w) to its fragments on the way from LDS to the matrix cores: the next group's fragments in R slices between
// the current group's items.  zl_rmsnorm's rounding points (two fp32 products, one rounding per element); the row sum in the
// statistics' association instead of zl_rmsnorm's (rs agrees to an fp32 rounding).  No stand-alone norm launch from 9 rows on
// (4.9 us each at 32 rows), no register-resident norm prologue below.
// The accumulation ORDER over the K / 128 groups differs from the phase kernel's (wave-major instead of phase-major), so the
// two kernels agree to fp32 rounding of the sums, not bit for bit; both are held to the same bars against the oracle
// (tests/test_gpu_w4.py, test_gpu_fullgeom.py).
// Reference: gptq_gemm_k_major's M <= 40 route, src/nn/quant/gptq/q_gemm_k_major.cu:580-686 (streams the weights once per 16
// rows: twice at M = 32).
#include <type_traits>
#include "zl_common.h"
#include "w4_i8p_common.h"

namespace {

constexpr int kMaxKS = 32;
#ifndef ZL_SLAB_DMAX
#define ZL_SLAB_DMAX 8
#endif
typedef __attribute__((address_space(3))) void* lds_ptr;

// ---- optional timeline probe (build a variant with -DZL_SLAB_PROBE; tools/ubench/probe_slab.py): wall-clock stamps per wave
#ifdef ZL_SLAB_PROBE
__device__ unsigned long long* zl_sprobe_p = nullptr;   // [workgroups * 8 waves][8] ticks of 10 ns
// (stamps stay in registers until the tile is summed: a store inside the loop would join the vmcnt queue the counted waits rely on)
#define ZL_SPROBE_INIT() unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define ZL_SPROBE(slot) do { st_[slot] = wall_clock64(); } while (0)
#define ZL_SPROBE_DUMP()                                                                                       \
    if (zl_sprobe_p && (threadIdx.x & 63) == 0 && blockIdx.x < 4096) {                                          \
        unsigned long long* sp_ = zl_sprobe_p + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;             \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) sp_[i_] = st_[i_];                                    \
    }
#else
#define ZL_SPROBE_INIT() do {} while (0)
#define ZL_SPROBE(slot) do {} while (0)
#define ZL_SPROBE_DUMP() do {} while (0)
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

struct SlabParams {
    const uint16_t* x;
    int64_t ldx;
    uint32_t x_bytes;
    const uint4* qw;
    const uint32_t* meta;
    uint32_t qw_bytes, meta_bytes;
    const uint16_t* bias;
    const uint16_t* residual;
    uint16_t* y;
    int m, n, k;
    int groups;        // 128-k items per row tile
    int tiles;         // 16-row tiles
    int epi, ld_out;
    int ks;            // K splits (workgroups per tile group)
    f4* ws;            // [grid][R MB 64] fp32 slabs (ks > 1)
    int* counters;     // one per tile group, zero between launches
    // ROPE instantiation (the fused qkv projection of a decode step)
    const float* cosv;
    const float* sinv;
    const int32_t* placement;
    const int32_t* buf_lens;
    uint16_t* const* k_bufs;
    uint16_t* const* v_bufs;
    uint16_t* q_out;
    int h, hkv, d, bshd;
    int tstride;       // tiles between a workgroup's consecutive tiles: 1; ROPE: d / 32 (a column block and its rotation partners)
    // row statistics hand-off (zl_w4_opts_t::row_ss / row_ss_out)
    const uint16_t* norm_w;   // NORM instantiation: the RMSNorm weight of x ...
    const float* row_ss;      // ... and its tile sums of squares [m][ss_parts]
    int ss_parts;             // k / 16
    float eps;
    float* ss_out;            // tile sums of the stored rows [m][tiles] (plain epilogues), or null
};

__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask_s, uint32_t magic_v) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask_s), "v"(magic_v));
    return r;
}

// eight nibbles of a ZLW4M word -> eight halves (q - z), exact (w4_phase.hip dequant_word)
__device__ __forceinline__ h8 dequant_word(uint32_t w, hv2 z1, hv2 z16, uint32_t mask_lo, uint32_t mask_hi, uint32_t magic) {
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

__device__ __forceinline__ void store_sc1(f4* dst, f4 v) {      // write-through 16-byte store
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ f4 load_sc1(const f4* src) {          // L1-bypassing read of a slab entry: two 8-byte sc1 loads the compiler can see
    const uint64_t* q = reinterpret_cast<const uint64_t*>(src);
    const uint64_t lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f4 v;
    v[0] = __builtin_bit_cast(float, (uint32_t)lo); v[1] = __builtin_bit_cast(float, (uint32_t)(lo >> 32));
    v[2] = __builtin_bit_cast(float, (uint32_t)hi); v[3] = __builtin_bit_cast(float, (uint32_t)(hi >> 32));
    return v;
}

// vmcnt field of s_waitcnt on gfx9 (6 bits: [3:0] and [15:14]); expcnt / lgkmcnt left at "no wait"
constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }

// Sum over the 16 lanes of a DPP row, every lane ending with the same bits: partners lane ^ 1, lane ^ 2, then the mirror image
// inside the half row (7 - i: the other quad of the half, whose lanes all hold one value by then) and inside the row (15 - i).
// This association -- ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) ... -- is THE definition of a tile's sum of squares
// (zl_w4_opts_t::row_ss): zl_row_ss and the producing epilogue both come through here.
__device__ __forceinline__ float zl_sum16(float t) {
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xf, 0xf, false));   // row_half_mirror
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x140, 0xf, 0xf, false));   // row_mirror
    return t;
}

template <int R, int GPW, int MB, bool ROPE, bool NORM>
__global__ __launch_bounds__(512, 2) void k_w4a16_slab(const SlabParams p) {
    // ring slots: D - 1 weight items (1 KiB + a meta word each) in flight per wave + the one being finished.  An 8-wave workgroup per CU
    // is all the registers allow at two row blocks, so the bytes in flight per CU are 8 (D - 1) KiB: profiles/r06_slab_ring_depth.txt
    constexpr int TOTAL = R * GPW, D = TOTAL < ZL_SLAB_DMAX ? TOTAL + 1 : ZL_SLAB_DMAX;
    constexpr int XS = GPW < 2 ? GPW : 2;             // activation groups resident in the wave's LDS region
    constexpr int DM = 4 * MB;                        // DMA instructions per group: 4 rows x 256 B each
    constexpr int kSet = MB * 16 * 256;               // bytes per group image
    constexpr int SLOTS = R * MB * 64;                // float4 per workgroup tile: slot = (r MB + b) 64 + lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    const int nrow = lane & 15, kq = lane >> 4;
    ZL_SPROBE_INIT();
    ZL_SPROBE(0);
    const int ksi = (int)(blockIdx.x % (unsigned)p.ks), tg = (int)(blockIdx.x / (unsigned)p.ks);
    // ROPE (R = 2): workgroup tg owns tiles {base, base + s}, base = (tg / s) * 2 s + tg % s, s = d / 32 -- a column block and the
    // block D / 2 columns further, its partners in the neox rotation (w4_phase.hip's pairing)
    const int tile0 = ROPE ? (tg / p.tstride) * 2 * p.tstride + tg % p.tstride : tg * R;
    const int tstride = ROPE ? p.tstride : 1;
    const int g0 = (ksi * nw + wave) * GPW;           // this wave's first 128-k group

    // ---- activations: group image [16 MB rows][16 pieces of 16 B], piece q of row r at position q ^ (r & 15)
    const uint32_t kOob = 0x80000000u;                // a lane offset no descriptor covers: returns zero, moves nothing
    // ---- NORM: the rows' statistics first (the oldest loads of the wave: they are back while the weight ring's prologue is still
    //      being requested).  A 16-lane group per row, four rows per wave and pass: row = (pass nw + wave) 4 + (lane >> 4); lane q of
    //      the group takes the tile sums 4 q .. 4 q + 3, + 64 u for u = 0 .. K / 1024 - 1.  Loads that have nothing to fetch (rows
    //      past M, u past K) go out of the descriptor's range: zeros, no traffic.
    constexpr int kSP = NORM ? (16 * MB / 16 > 1 ? 2 : 1) : 1;     // passes at the smallest workgroup (4 waves: 16 rows per pass)
    constexpr int kSU = 8;                                         // K <= 8192
    f4 sst[kSP][kSU];
    if constexpr (NORM) {
        const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.row_ss), 0,
                                                                             (uint32_t)p.m * (uint32_t)p.ss_parts * 4u, 0x00020000);
#pragma unroll
        for (int ps = 0; ps < kSP; ++ps) {
            const int row = (ps * nw + wave) * 4 + (lane >> 4);
#pragma unroll
            for (int u = 0; u < kSU; ++u) {
                const int c = 4 * (lane & 15) + 64 * u;
                const bool live = row < p.m && c < p.ss_parts;
                sst[ps][u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rst, live ? (uint32_t)(row * p.ss_parts + c) * 4u : kOob, 0, 0));
            }
        }
    }
    // ---- NORM: rs = rsqrt_rn(sum x^2 / K + eps) per row, through LDS to the lanes that hold the row's fragments
    float rs[MB];
    auto compute_rs = [&]() {
      if constexpr (NORM) {
        float* rsl = reinterpret_cast<float*>(smem + (size_t)nw * (XS * kSet) + (size_t)nw * 1024);
#pragma unroll
        for (int ps = 0; ps < kSP; ++ps) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < kSU; ++u) t += (sst[ps][u][0] + sst[ps][u][1]) + (sst[ps][u][2] + sst[ps][u][3]);
            t = zl_sum16(t);
            const int row = (ps * nw + wave) * 4 + (lane >> 4);
            if ((lane & 15) == 0 && row < 16 * MB) rsl[row] = zl_rsqrt_rn(t / (float)p.k + p.eps);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (not __syncthreads: its fence would drain the weight ring)
#pragma unroll
        for (int b = 0; b < MB; ++b) rs[b] = rsl[16 * b + nrow];
    }
    };
#ifdef ZL_SLAB_STATS_FIRST
    compute_rs();                                     // (A/B: the barrier at entry, before anything else is requested)
#endif
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    unsigned char* xw = smem + (size_t)wave * (XS * kSet);
    // DMA lane -> (row 4 i + (lane >> 4), position lane & 15): rows past M re-read the last row (their outputs are never
    // stored, and a row of C depends on its own row of x only)
    uint32_t x_off[DM];
#pragma unroll
    for (int i = 0; i < DM; ++i) {
        const int row = min(4 * i + (lane >> 4), p.m - 1);
        x_off[i] = (uint32_t)((row * (int)p.ldx) * 2 + (((lane & 15) ^ ((4 * i + (lane >> 4)) & 15)) * 16));
    }
    // No control flow around the DMAs: behind a branch the compiler's waitcnt pass merges the two paths' pending counts and
    // then waits for the weight items as if no DMA were in flight, i.e. for the DMAs too.  A group past K (a partial last split)
    // reads through an out-of-range lane offset instead -- zeros; and since it also meets zero WEIGHTS, whose 0 x inf would be
    // NaN, the wave's region is cleared once up front in case such a load leaves LDS alone.
    if (g0 + GPW > p.groups) {                        // wave-uniform, rare: this wave's slice reaches past K
#pragma unroll
        for (int i = 0; i < XS * DM; ++i) *reinterpret_cast<uint4*>(xw + i * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
        if constexpr (NORM) *reinterpret_cast<uint4*>(smem + (size_t)nw * (XS * kSet) + (size_t)wave * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    auto dma_x = [&](int set, int j) {                // both static; group g0 + j into region `set`
        const bool live = g0 + j < p.groups;
#pragma unroll
        for (int i = 0; i < DM; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass silently drops a kernel stub whose body names this builtin: ROCm 7.2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(xw + set * kSet + i * 1024), 16, live ? x_off[i] : kOob, (g0 + j) * 256, 0, 0);
#else
            (void)live;
#endif
        }
    };
    // NORM: the norm weights of the wave's GPW groups (GPW x 256 B, consecutive) behind the x regions, one DMA
    unsigned char* nww = smem + (size_t)nw * (XS * kSet) + (size_t)wave * 1024;
    auto dma_prologue = [&]() {
        if constexpr (NORM) {
#if defined(__HIP_DEVICE_COMPILE__)
            const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.norm_w), 0, (uint32_t)p.k * 2u, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rn, (lds_ptr)nww, 16, (uint32_t)lane * 16u, g0 * 256, 0, 0);
#endif
        }
#pragma unroll
        for (int j = 0; j < XS; ++j) dma_x(j, j);
        __builtin_amdgcn_sched_barrier(0);
    };
    // ZL_SLAB_RING_FIRST (A/B): the weight ring's prologue is requested BEFORE the activation DMAs -- HBM requests leave 1..3 us
    // earlier, the images (L2) land behind them; the prologue items are then OLDER than DMA(0) / DMA(1) in the wait counts below
#ifdef ZL_SLAB_RING_FIRST
    constexpr bool kRingFirst = true;
#else
    constexpr bool kRingFirst = false;
    dma_prologue();
#endif

    // ---- weight ring: item i = (group j = i / R, tile r = i % R)
    uint4 wq[D];
    uint32_t mt[D];
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.qw), 0, p.qw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.meta), 0, p.meta_bytes, 0x00020000);
    // groups past K: the lane offset goes out of range (zeros, nothing moved); tiles past N (a ragged last tile group): the
    // item of the LAST tile instead -- real memory, and the epilogue never writes a tile >= tiles
    uint32_t q_off[GPW], m_off[GPW];
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
        q_off[j] = g0 + j < p.groups ? (uint32_t)lane * 16u : kOob;
        m_off[j] = g0 + j < p.groups ? (uint32_t)nrow * 4u : kOob;
    }
    auto issue = [&](int slot, int i) {               // both static
        const int j = i / R, r = i % R;
        const uint32_t it = (uint32_t)min(tile0 + r * tstride, p.tiles - 1) * (uint32_t)p.groups + (uint32_t)(g0 + j);
        wq[slot] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rq, q_off[j], it * 1024u, 2 /* nt */));
        mt[slot] = __builtin_amdgcn_raw_buffer_load_b32(rm, m_off[j], it * 64u, 2);
    };
#pragma unroll
    for (int s = 0; s < D - 1; ++s) {
        issue(s, s);
        __builtin_amdgcn_sched_barrier(0);
    }
    mt[D - 1] = 0;                                    // the neutral "previous item" of the first step
    wq[D - 1] = make_uint4(0, 0, 0, 0);
    if constexpr (kRingFirst) dma_prologue();
    ZL_SPROBE(1);                                     // prologue requested

#ifndef ZL_SLAB_STATS_FIRST
    compute_rs();
#endif

    const uint32_t mask_lo = __builtin_amdgcn_readfirstlane(0x000f000fu);
    const uint32_t mask_hi = __builtin_amdgcn_readfirstlane(0x00f000f0u);
    uint32_t magic = 0x64006400u;
    asm volatile("" : "+v"(magic));

    f4 acc[R][MB], accg_prev[MB];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[r][b] = (f4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < MB; ++b) accg_prev[b] = (f4){0.f, 0.f, 0.f, 0.f};
    auto finish_prev = [&](int rp, int pslot) {       // acc[rp] += scale * group sums of the previous item
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            asm("v_fma_mix_f32 %0, %4, %8, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n\t"
                "v_fma_mix_f32 %1, %5, %8, %1 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n\t"
                "v_fma_mix_f32 %2, %6, %8, %2 op_sel:[0,0,0] op_sel_hi:[0,1,0]\n\t"
                "v_fma_mix_f32 %3, %7, %8, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]"
                : "+v"(acc[rp][b][0]), "+v"(acc[rp][b][1]), "+v"(acc[rp][b][2]), "+v"(acc[rp][b][3])
                : "v"(accg_prev[b][0]), "v"(accg_prev[b][1]), "v"(accg_prev[b][2]), "v"(accg_prev[b][3]), "v"(mt[pslot]));
        }
    };

    // fragment reads: the asm keeps the compiler from putting its own (conservative: vmcnt(0)) wait between an LDS-DMA and a
    // read of LDS -- the wait in front of the reads is the counted one below
    const uint32_t xr_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)xw + (uint32_t)nrow * 256u;
    constexpr int XB = NORM ? 2 : 1;                  // NORM: the next group's fragments are normalised while this one's are used
    uint4 xa[XB][MB][4], nwf[4];
    auto wait_vm = [&](int n) {                       // n: static after unrolling (the builtin wants an immediate)
        switch (n) {
#define ZL_W(n) case n: __builtin_amdgcn_s_waitcnt(vmcnt_imm(n)); break;
            ZL_W(0) ZL_W(2) ZL_W(4) ZL_W(6) ZL_W(8) ZL_W(10) ZL_W(12) ZL_W(14) ZL_W(16) ZL_W(18) ZL_W(20) ZL_W(22) ZL_W(24) ZL_W(26)
            ZL_W(28) ZL_W(30) ZL_W(32) ZL_W(34) ZL_W(36) ZL_W(38) ZL_W(40) ZL_W(42) ZL_W(44) ZL_W(46) ZL_W(48) ZL_W(50) ZL_W(52) ZL_W(54)
#undef ZL_W
            default: __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); break;
        }
    };
    auto read_group = [&](int buf, int g) {           // group g's fragments (and, NORM, its norm weights) LDS -> registers
#pragma unroll
        for (int b = 0; b < MB; ++b) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t addr = xr_base + (uint32_t)((g % XS) * kSet + b * 4096) + (uint32_t)(((4 * t + kq) ^ nrow) * 16);
                asm volatile("ds_read_b128 %0, %1" : "=v"(xa[buf][b][t]) : "v"(addr) : "memory");
            }
        }
        if constexpr (NORM) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)nww + (uint32_t)(g * 256 + (4 * t + kq) * 16);
                asm volatile("ds_read_b128 %0, %1" : "=v"(nwf[t]) : "v"(addr) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // T(x rs w): zl_rmsnorm's two fp32 products and its one rounding (misc_ops.hip k_rmsnorm; layernorm.cu:10-42), fragment f = 4 b + t
    auto norm_frag = [&](int buf, int f) {
        const int b = f / 4, t = f % 4;
        const h8 wv = __builtin_bit_cast(h8, nwf[t]);
        h8 xv = __builtin_bit_cast(h8, xa[buf][b][t]);
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = (_Float16)(((float)xv[e] * rs[b]) * (float)wv[e]);
        xa[buf][b][t] = __builtin_bit_cast(uint4, xv);
    };
    // WHEN HAS A GROUP'S IMAGE LANDED?  Once at most the loads issued AFTER its DMAs are outstanding (in-order return): the DMAs of
    // later groups already requested, and the ring items in flight that were requested after it.  With few tiles per workgroup
    // (R = 1, 2) the items in flight are mostly OLDER than the DMA -- counting them all let the fragment reads run ahead of it (the
    // first cut of this kernel: wrong rows for R <= 2 with four groups per wave).  The counts never exceed what the next weight
    // item itself has to wait for, so the waits cost nothing on top.
    if constexpr (NORM) {
        // NORM schedule: group 0 is read and normalised here, behind the prologue (nothing else to do until the first item lands);
        // group j + 1 is read at the START of group j and normalised in R slices in front of group j's items' MFMAs -- a burst of ~200
        // VALU per group in front of its first item keeps the wave from refilling its ring for that long.
        // DMA order: groups 0, 1 in the prologue, 2 here, g + 2 when group g has been read.
        wait_vm((GPW > 1 ? DM : 0) + (kRingFirst ? 0 : 2 * (D - 1)));
        read_group(0, 0);
        ZL_SPROBE(2);
        if (XS < GPW) dma_x(0, XS);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 4 * MB; ++f) norm_frag(0, f);
        __builtin_amdgcn_sched_barrier(0);
    }
    // (two nested loops, not one over the R GPW items with the group's work under `if (r == 0)`: the unroller prices a loop by trip
    //  count x body BEFORE it knows the condition is static, and gave up on the NORM instantiations with 28 items)
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
        const int cur = NORM ? (j & 1) : 0;
        if constexpr (NORM) {
            if (j + 1 < GPW) {
                const int g = j + 1, i = j * R;
                // younger than DMA(g): DMA(g + 1) (requested when group g - 1 was read), and the ring items requested after DMA(g) went
                // out -- in the prologue for g < XS (all D - 1 prologue items follow it), else when group g - XS was read, i.e. ahead of
                // the refill that follows item max(0, g - XS - 1) R
                const int fy = g < XS ? (kRingFirst ? D - 1 : 0) : ((g - XS - 1 > 0 ? g - XS - 1 : 0) * R + D - 1);
                const int lo = i > fy ? i : fy, hi = (i + D - 2) < (TOTAL - 1) ? (i + D - 2) : (TOTAL - 1);
                wait_vm((g + 1 < GPW ? DM : 0) + 2 * (hi - lo + 1 > 0 ? hi - lo + 1 : 0));
                read_group(cur ^ 1, g);
                if (g < 4) ZL_SPROBE(2 + g);
                if (g + XS < GPW) dma_x(g % XS, g + XS);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const int i = j * R;
            // DMA(j) went out in the prologue (j < XS) or when group j - XS had been read, behind that step's fragment reads and ahead
            // of its refill: only items from (j - XS) R + D - 1 on are younger
            const int younger_groups = (j == 0) ? (XS - 1 < GPW - 1 ? XS - 1 : GPW - 1) : ((j + 1 < GPW) ? 1 : 0);
            const int fly_hi = (i + D - 2) < (TOTAL - 1) ? (i + D - 2) : (TOTAL - 1);
            const int first_younger = j < XS ? (kRingFirst && D - 1 > i ? D - 1 : i) : ((j - XS) * R + D - 1 > i ? (j - XS) * R + D - 1 : i);
            const int ring_fly = fly_hi - first_younger + 1 > 0 ? fly_hi - first_younger + 1 : 0;
            wait_vm(younger_groups * DM + 2 * ring_fly);
            read_group(0, j);
            if (j < 4) ZL_SPROBE(2 + j);              // group j's fragments in registers
            if (j + XS < GPW) dma_x(j % XS, j + XS);  // the region is free again: two groups ahead
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
        const int i = j * R + r, slot = i % D, pslot = (i + D - 1) % D;
        const hv2 z1 = __builtin_bit_cast(hv2, __builtin_amdgcn_perm(mt[slot], mt[slot], 0x03020302u));
        const hv2 c960 = {(_Float16)960.f, (_Float16)960.f};
        const hv2 z16 = z1 + c960;
        const uint32_t wds[4] = {wq[slot].x, wq[slot].y, wq[slot].z, wq[slot].w};
        h8 a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = dequant_word(wds[t], z1, z16, mask_lo, mask_hi, magic);
        finish_prev((i + R - 1) % R, pslot);
        __builtin_amdgcn_sched_barrier(0);
        // This item's slice of the next group's normalisation goes IN FRONT of the item's MFMAs.  Behind them -- where it would run in
        // the shadow of the matrix pipe -- the R = 8 / one-row-block instantiation returned element 0 of every lane's C fragment wrong
        // for tiles 2 and 3 of each workgroup (rows 4 q), deterministically, with every operand of the MFMAs kept allocated past the
        // slice (-DZL_SLAB_SLICE_AFTER rebuilds that order; tools/debug_slab_norm.py shows it): something in a dependent chain of four
        // v_mfma_f32_16x16x32_f16 does not survive ~30 VALU (v_cvt_f32_f16_sdwa, v_pk_mul_f32, v_cvt_pk_f16_f32) in the issue slots
        // right behind it on gfx950 / ROCm 7.2.  What follows the MFMAs now is what has followed them in every kernel of this family
        // since round 2: the refill's address arithmetic and its two loads.
#ifndef ZL_SLAB_SLICE_AFTER
        if constexpr (NORM) {
            if (j + 1 < GPW) {
#pragma unroll
                for (int f = 0; f < 4 * MB; ++f) {
                    if (f % R == r) norm_frag(cur ^ 1, f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        f4 accg[MB], mid[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) accg[b] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int b = 0; b < MB; ++b) {
                if (t == 3) mid[b] = accg[b];
                accg[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, xa[cur][b][t]), a[t], accg[b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef ZL_SLAB_SLICE_AFTER
        if constexpr (NORM) {
            if (j + 1 < GPW) {
#pragma unroll
                for (int f = 0; f < 4 * MB; ++f) {
                    if (f % R == r) norm_frag(cur ^ 1, f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            asm volatile("" : : "v"(mid[b]));
            accg_prev[b] = accg[b];
        }
        if (i + D - 1 < TOTAL) issue(pslot, i + D - 1);
        __builtin_amdgcn_sched_barrier(0);
        // The MFMAs' OPERANDS stay allocated up to here, past the refill's issue.  hipcc (ROCm 7.2) hands the registers of the LAST MFMA's
        // B operand to the next VALU result without a wait state (`v_mfma .. v[36:39] ..` / `v_cvt_f32_f16 v36, ..` in consecutive
        // issue slots: seen with the slice behind the MFMAs, R = 7 / one row block -- rows 4 q of three to five tiles of every
        // workgroup wrong); the recogniser knows no write-after-read hazard on the A / B operands of v_mfma_f32_16x16x32_f16
        // (w4_i8p.hip met the same gap in round 3 with the i8 instruction).
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            asm volatile("" : : "v"(__builtin_bit_cast(f4, a[t])));
#pragma unroll
            for (int b = 0; b < MB; ++b) asm volatile("" : : "v"(__builtin_bit_cast(f4, xa[cur][b][t])));
        }
        __builtin_amdgcn_sched_barrier(0);
        }
    }
    {
        // The last item's scale-accumulate, in plain C++: inside the loop a dequantisation block (>= 36 VALU) sits between an
        // item's MFMAs and the asm that reads their results; here nothing does, and the hazard recognizer does not look into
        // inline asm -- the v_fma_mix read the accumulators before the matrix pipe had written them (the first cuts: the LAST
        // tile of every workgroup wrong in rows 4 q + {0, 1, 2}).  fma(group sum, float(scale), acc) is what v_fma_mix_f32 computes.
        constexpr int rl = (TOTAL - 1) % R, sl = (TOTAL - 1) % D;
        const float sc_last = (float)__builtin_bit_cast(hv2, mt[sl]).x;
#pragma unroll
        for (int b = 0; b < MB; ++b) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[rl][b][e] = __builtin_fmaf(accg_prev[b][e], sc_last, acc[rl][b][e]);
        }
    }
    ZL_SPROBE(6);                                      // this wave's items done
    __syncthreads();                                   // every wave is done with its activation region: LDS is reused below

    // ---- the NW partial tiles meet in LDS, summed in wave order
    f4* red = reinterpret_cast<f4*>(smem);             // [nw][R MB][64]
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int b = 0; b < MB; ++b) red[((size_t)(wave * R + r) * MB + b) * 64 + lane] = acc[r][b];
    }
    __syncthreads();
    constexpr int kMaxIt = SLOTS < 256 ? 1 : SLOTS / 256;   // slots per thread at the smallest workgroup (4 waves)
    f4 val[kMaxIt];
    const int bd = (int)blockDim.x;
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
        const int s = (int)threadIdx.x + it * bd;
        f4 v = (f4){0.f, 0.f, 0.f, 0.f};
        if (s < SLOTS) {
            for (int w = 0; w < nw; ++w) v += red[(size_t)w * SLOTS + s];
        }
        val[it] = v;
    }

    ZL_SPROBE(7);                                      // the workgroup's tile is summed (epilogue / K-split tail follow)
    ZL_SPROBE_DUMP();
    if (p.ks > 1) {
        // ---- K split: slab out (write-through), ticket, the last arriver of the tile group folds the KS slabs in split order
        f4* slab = p.ws + (size_t)blockIdx.x * SLOTS;
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            const int s = (int)threadIdx.x + it * bd;
            if (s < SLOTS) store_sc1(slab + s, val[it]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // every wave's stores are out; red is free
        int* flag = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(p.counters + tg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == p.ks - 1) __hip_atomic_store(p.counters + tg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = old == p.ks - 1;
        }
        __syncthreads();
        if (!*flag) return;
        const f4* first = p.ws + (size_t)tg * p.ks * SLOTS;
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            const int s = (int)threadIdx.x + it * bd;
            f4 v = (f4){0.f, 0.f, 0.f, 0.f};
            if (s < SLOTS) {
                for (int sp0 = 0; sp0 < p.ks; sp0 += 8) {
                    f4 part[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) part[u] = load_sc1(first + (size_t)min(sp0 + u, p.ks - 1) * SLOTS + s);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (sp0 + u < p.ks) v += part[u];
                    }
                }
            }
            val[it] = v;
        }
    }

    if constexpr (ROPE) {
        // ---- rotate q and k (neox) on the fp16-rounded projection outputs, scatter k / v into the ragged buffers
        //      (rope_qk_cache + copy_to_rag_buffer2, src/nn/position/rotary_embedding_fuse_cache.cu:65-125,
        //      src/kvcache/ragged_buffer_kernel.cu:194-222; roundings of the separate kernels: rope_common.cuh:14-34)
        f4* fin = reinterpret_cast<f4*>(smem);         // [SLOTS]
        __syncthreads();                               // every thread is done with red / the flag: smem is reused
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            const int s = (int)threadIdx.x + it * bd;
            if (s < SLOTS) fin[s] = val[it];
        }
        __syncthreads();
        const float* finf = reinterpret_cast<const float*>(fin);
        const int half = p.d >> 1;
        const int nout = 16 * p.m;
        for (int o = (int)threadIdx.x; o < nout; o += bd) {
            const int n_local = o & 15, m = o >> 4;
            const int tile = tile0;
            if (tile >= p.tiles) continue;
            const int b = m >> 4, ln = ((m & 15) >> 2) * 16 + n_local, i = m & 3;
            float v0 = finf[((size_t)(0 * MB + b) * 64 + ln) * 4 + i], v1 = finf[((size_t)(1 * MB + b) * 64 + ln) * 4 + i];
            const int n0 = tile * 16 + n_local, n1 = n0 + half;
            if ((p.epi & ZL_EPI_BIAS) && p.bias) {
                v0 += (float)__builtin_bit_cast(_Float16, p.bias[n0]);
                v1 += (float)__builtin_bit_cast(_Float16, p.bias[n1]);
            }
            const float a = (float)zl_f32_to_f16(v0), bb = (float)zl_f32_to_f16(v1);
            const int head = n0 / p.d, dcol = n0 % p.d;
            if (head < p.h + p.hkv) {
                const float c0 = p.cosv[(size_t)m * p.d + dcol], s0 = p.sinv[(size_t)m * p.d + dcol];
                const float c1 = p.cosv[(size_t)m * p.d + dcol + half], s1 = p.sinv[(size_t)m * p.d + dcol + half];
                const uint16_t r0v = __builtin_bit_cast(uint16_t, zl_f32_to_f16(__builtin_fmaf(-bb, s0, a * c0)));
                const uint16_t r1v = __builtin_bit_cast(uint16_t, zl_f32_to_f16(__builtin_fmaf(a, s1, bb * c1)));
                if (head < p.h) {
                    uint16_t* dst = p.q_out + ((size_t)m * p.h + head) * p.d + dcol;
                    dst[0] = r0v;
                    dst[half] = r1v;
                } else {
                    const int place = p.placement[m], blen = p.buf_lens[m];
                    if (place >= 0 && place < blen) {
                        const int hk = head - p.h;
                        const size_t row = p.bshd ? (size_t)place * p.hkv + hk : (size_t)hk * blen + place;
                        uint16_t* dst = p.k_bufs[m] + row * p.d + dcol;
                        dst[0] = r0v;
                        dst[half] = r1v;
                    }
                }
            } else {
                const int place = p.placement[m], blen = p.buf_lens[m];
                if (place >= 0 && place < blen) {
                    const int hk = head - p.h - p.hkv;
                    const size_t row = p.bshd ? (size_t)place * p.hkv + hk : (size_t)hk * blen + place;
                    uint16_t* dst = p.v_bufs[m] + row * p.d + dcol;
                    dst[0] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(v0));
                    dst[half] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(v1));
                }
            }
        }
        return;
    } else {
        // ---- epilogue from the C fragment: a thread holds rows 16 b + 4 (lane >> 4) + i, i < 4, of ONE output column
        const bool silu = (p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
#pragma unroll
        for (int it = 0; it < kMaxIt; ++it) {
            const int s = (int)threadIdx.x + it * bd;
            if (s >= SLOTS) continue;                 // wave-uniform (bd and SLOTS are multiples of 64)
            const int rb = s >> 6, ln = s & 63, r = rb / MB, b = rb % MB;
            const int tile = tile0 + r * tstride, n_local = ln & 15, mrow0 = b * 16 + (ln >> 4) * 4;
            f4 v = val[it];
            if (silu) {
                // gate at even, up at odd columns of the interleaved weight: the neighbour lane holds the other one
                f4 u;
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = __shfl_xor(v[i], 1, 64);
                const int prc = tile * 8 + (n_local >> 1);
                if ((n_local & 1) || tile >= p.tiles || 2 * prc + 1 >= p.n) continue;
                const bool has_bias = (p.epi & ZL_EPI_BIAS) && p.bias;
                const float bg = has_bias ? (float)__builtin_bit_cast(_Float16, p.bias[2 * prc]) : 0.f;
                const float bu = has_bias ? (float)__builtin_bit_cast(_Float16, p.bias[2 * prc + 1]) : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = mrow0 + i;
                    if (m >= p.m) continue;
                    float g = has_bias ? v[i] + bg : v[i], uu = has_bias ? u[i] + bu : u[i];
                    float ov;
                    if (p.epi & ZL_EPI_SILU_MUL) {
                        g = (float)zl_f32_to_f16(g);
                        uu = (float)zl_f32_to_f16(uu);
                        ov = silu_f32(g) * uu;
                    } else {
                        ov = (float)((double)g / (1.0 + (double)expf(-g))) * uu;
                    }
                    p.y[(size_t)m * p.ld_out + prc] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(ov));
                }
                continue;
            }
            const int col = tile * 16 + n_local;
            const bool live = tile < p.tiles && col < p.n;
            if (!live && !p.ss_out) continue;
            const float bb = (live && (p.epi & ZL_EPI_BIAS) && p.bias) ? (float)__builtin_bit_cast(_Float16, p.bias[col]) : 0.f;
            float res[4] = {0.f, 0.f, 0.f, 0.f}, cin[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {             // operands first, arithmetic after: one round trip for the four rows
                const int m = mrow0 + i;
                if (live && m < p.m) {
                    if (p.epi & ZL_EPI_RESIDUAL) res[i] = (float)__builtin_bit_cast(_Float16, p.residual[(size_t)m * p.ld_out + col]);
                    if (p.epi & ZL_EPI_ADD_C) cin[i] = (float)__builtin_bit_cast(_Float16, p.y[(size_t)m * p.ld_out + col]);
                }
            }
            float sq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mrow0 + i;
                sq[i] = 0.f;
                if (!live || m >= p.m) continue;
                float ov;
                if (p.epi & ZL_EPI_ADD_C) ov = (cin[i] + v[i]) + bb;
                else ov = v[i] + bb;
                _Float16 y16 = zl_f32_to_f16(ov);
                if (p.epi & ZL_EPI_RESIDUAL) y16 = zl_f32_to_f16(res[i] + (float)y16);
                p.y[(size_t)m * p.ld_out + col] = __builtin_bit_cast(uint16_t, y16);
                sq[i] = (float)y16 * (float)y16;      // exact in fp32
            }
            if (p.ss_out) {
                // the tile's sum of squares per row: 16-lane butterfly (every lane ends with the same bits), zl_row_ss's order.
                // Whole waves get here: s < SLOTS is wave-uniform and dead columns came along with zeros.
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sq[i] = zl_sum16(sq[i]);
                    const int m = mrow0 + i;
                    if (n_local == 0 && tile < p.tiles && m < p.m) p.ss_out[(size_t)m * p.tiles + tile] = sq[i];
                }
            }
        }
    }
}

// zl_row_ss: the tile sums a producing launch leaves in row_ss_out, from rows that are already in memory.  A 16-lane group per
// (row, tile): one value per lane, the epilogue's butterfly.
__global__ __launch_bounds__(256) void k_row_ss(const uint16_t* __restrict__ x, int64_t ldx, int m, int parts, float* __restrict__ out) {
    const int row = (int)blockIdx.y;
    const int tile = (int)((blockIdx.x * 256u + threadIdx.x) >> 4), c = (int)(threadIdx.x & 15);
    const bool live = tile < parts;                   // 16-lane groups are all-live or all-dead
    const float v = live ? (float)__builtin_bit_cast(_Float16, x[(size_t)row * ldx + (size_t)tile * 16 + c]) : 0.f;
    const float q = zl_sum16(v * v);
    if (live && c == 0) out[(size_t)row * parts + tile] = q;
}

template <int R, int GPW, int MB, bool ROPE, bool NORM>
int launch_slab(const SlabParams& p, int grid, int nw, hipStream_t hs) {
    const size_t red_bytes = (size_t)nw * R * MB * 64 * 16;
    const size_t x_bytes = (size_t)nw * (GPW < 2 ? GPW : 2) * MB * 16 * 256 + (NORM ? (size_t)nw * 1024 + 32 * 4 : 0);
    const size_t lds = red_bytes > x_bytes ? red_bytes : x_bytes;
    if (lds > 64 * 1024) {
        // every launch: the attribute is per device, and one process may drive several
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_w4a16_slab<R, GPW, MB, ROPE, NORM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ZL_ELIMIT;
    }
    hipLaunchKernelGGL((k_w4a16_slab<R, GPW, MB, ROPE, NORM>), dim3(grid), dim3(64 * nw), lds, hs, p);
    return zl_launch_status();
}

template <int R, bool ROPE>
int launch_slab_r(const SlabParams& p, int gpw, int mb, int grid, int nw, hipStream_t hs) {
    if (p.norm_w) {                                   // the NORM instantiations: four groups per wave only (K > 2048)
        if (gpw != 4) return ZL_ESHAPE;
        if (mb == 1) return launch_slab<R, 4, 1, ROPE, true>(p, grid, nw, hs);
        return launch_slab<R, 4, 2, ROPE, true>(p, grid, nw, hs);
    }
    if (mb == 1) {
        if (gpw == 1) return launch_slab<R, 1, 1, ROPE, false>(p, grid, nw, hs);
        if (gpw == 2) return launch_slab<R, 2, 1, ROPE, false>(p, grid, nw, hs);
        return launch_slab<R, 4, 1, ROPE, false>(p, grid, nw, hs);
    }
    if (gpw == 1) return launch_slab<R, 1, 2, ROPE, false>(p, grid, nw, hs);
    if (gpw == 2) return launch_slab<R, 2, 2, ROPE, false>(p, grid, nw, hs);
    return launch_slab<R, 4, 2, ROPE, false>(p, grid, nw, hs);
}

struct SlabPlan {
    int r, nw, gpw, ks, grid;
};

// Geometry, from the sweep in profiles/r06_slab_sweep.txt (Llama-3-8B shapes, 9 / 16 / 32 rows; every (R, NW, GPW) that fits):
//   * eight waves x four groups per wave win wherever K allows (32 groups = 4096 k per workgroup: the most weight items per wave,
//     and no K split up to K = 4096 -- a split costs its tail, 3..5 us: slab out, ticket, fold); shorter K: fewer groups per wave;
//   * longer K splits: KS = ceil(groups / 32) (down: 4);
//   * R = the fewest tiles per workgroup that still give ONE generation of workgroups (attn_out 1, qkv 2, down 4): more tiles per
//     workgroup would cut activation traffic further but leave CUs without a workgroup, which costs more;
//   * many tiles per CU (gate|up: 7 x 256) -- the phase kernel already amortises its staging there: a tie up to 16 rows (not taken),
//     20.9 against 22..23 us with 17..32 rows (taken, R = 7);
//   * the ring depth (8 / 12 / 16 slots) changes nothing (profiles/r06_slab_ring_depth.txt): 8.
// the NORM instantiations read the rows' statistics (zl_w4_opts_t::row_ss): K a multiple of 1024 up to 8192 (a lane takes 4
// tile sums, twice at most), four groups per wave
bool norm_ok(const zl_w4_opts_t& o, int k) { return o.row_ss != nullptr && k % 1024 == 0 && k <= 8192 && k > 2048; }

bool plan_slab(int m, int tiles, int groups, bool have_scratch, bool rope, bool norm, const zl_w4_opts_t& o, SlabPlan* out) {
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    int nw = groups >= 8 ? 8 : 4, gpw = groups > 16 ? 4 : groups > 8 ? 2 : 1;
    if (o.slab_nw) nw = o.slab_nw;
    if (o.slab_gpw) gpw = o.slab_gpw;
    const int gw = nw * gpw, ks = (groups + gw - 1) / gw;
    if (ks > kMaxKS || (ks > 1 && !have_scratch)) return false;
    const bool forced = o.slab_r != 0;
    int r = rope ? 2 : 1;
    if (forced) r = o.slab_r;
    else if (!rope) {
        if (tiles > 3 * cus && o.slab != 2 && m <= 16 && !norm) return false;   // (two row blocks: 20.9 vs 22..23 us on gate|up, taken; one: a tie)
        while (r < 8 && (long)((tiles + r - 1) / r) * ks > cus) r *= 2;
        if (r == 8 && (long)((tiles + 6) / 7) * ks <= cus) r = 7;      // gate|up: 1792 tiles = 256 x 7
    }
    if (rope && r != 2) return false;
    if (norm && gpw != 4) return false;
    const long grid = (long)((tiles + r - 1) / r) * ks;
    if (grid > 2L * cus && !forced) return false;
    out->r = r; out->nw = nw; out->gpw = gpw; out->ks = ks; out->grid = (int)grid;
    return true;
}

template <bool ROPE>
int launch_slab_any(const SlabParams& p, const SlabPlan& pl, int mb, hipStream_t hs) {
    if constexpr (ROPE) return launch_slab_r<2, true>(p, pl.gpw, mb, pl.grid, pl.nw, hs);
    else switch (pl.r) {
        case 1: return launch_slab_r<1, false>(p, pl.gpw, mb, pl.grid, pl.nw, hs);
        case 2: return launch_slab_r<2, false>(p, pl.gpw, mb, pl.grid, pl.nw, hs);
        case 4: return launch_slab_r<4, false>(p, pl.gpw, mb, pl.grid, pl.nw, hs);
        case 7: return launch_slab_r<7, false>(p, pl.gpw, mb, pl.grid, pl.nw, hs);
        default: return launch_slab_r<8, false>(p, pl.gpw, mb, pl.grid, pl.nw, hs);
    }
}

void fill_common(SlabParams& p, const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                 uint32_t meta_bytes, const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups,
                 int tiles, int epilogue, int ld_out) {
    p.x = x; p.ldx = ldx; p.x_bytes = (uint32_t)((((int64_t)m - 1) * ldx + k) * 2);
    p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes; p.meta_bytes = meta_bytes;
    p.bias = bias; p.residual = residual; p.y = y; p.m = m; p.n = n; p.k = k; p.groups = groups; p.tiles = tiles;
    p.epi = epilogue; p.ld_out = ld_out;
}

bool slab_shape_ok(int m, int k, int groups, int tiles, int64_t ldx, uint32_t qw_bytes) {
    if (m < 1 || m > 32 || k % 128 != 0 || groups * 128 != k) return false;
    if ((((int64_t)m - 1) * ldx + k) * 2 >= ((int64_t)1 << 31)) return false;                     // 32-bit lane offsets into x
    if (((int64_t)tiles + 8) * groups * 1024 >= ((int64_t)1 << 32) || qw_bytes == 0) return false;    // 32-bit item offsets
    return true;
}

bool take_scratch(const zl_w4_opts_t& o, const SlabPlan& pl, int mb, int tiles, SlabParams& p) {
    const int64_t need = ZL_SCRATCH_HEADER + (int64_t)pl.grid * pl.r * mb * 64 * 16;
    if (!o.scratch || o.scratch_bytes < need || (tiles + pl.r - 1) / pl.r > ZL_SCRATCH_HEADER / 4) return false;
    p.counters = reinterpret_cast<int*>(o.scratch);
    p.ws = reinterpret_cast<f4*>(reinterpret_cast<char*>(o.scratch) + ZL_SCRATCH_HEADER);
    return true;
}

}  // namespace

// internal (zl_w4a16_gemm_mfma_ex): rows 1..32 without a fused norm, K a multiple of 128.  Returns ZL_ESHAPE when the shape (or
// the caller's scratch) does not fit -- the caller then takes the phase kernel.
int zl_w4a16_gemm_slab(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                       const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups, int tiles,
                       int epilogue, int ld_out, const uint16_t* norm_w, float norm_eps, const zl_w4_opts_t* opts, hipStream_t hs) {
    static const zl_w4_opts_t kNoOpts = {};
    const zl_w4_opts_t& o = opts ? *opts : kNoOpts;
    if (!slab_shape_ok(m, k, groups, tiles, ldx, qw_bytes)) return ZL_ESHAPE;
    if (norm_w && !norm_ok(o, k)) return ZL_ESHAPE;
    SlabPlan pl;
    if (!plan_slab(m, tiles, groups, o.scratch != nullptr, false, norm_w != nullptr, o, &pl)) return ZL_ESHAPE;
    SlabParams p = {};
    fill_common(p, x, ldx, qw, meta, qw_bytes, meta_bytes, bias, residual, y, m, n, k, groups, tiles, epilogue, ld_out);
    if (norm_w) { p.norm_w = norm_w; p.row_ss = o.row_ss; p.ss_parts = k / 16; p.eps = norm_eps; }
    if (o.row_ss_out && !(epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) && tiles * 16 == n) p.ss_out = o.row_ss_out;
    const int mb = m <= 16 ? 1 : 2;
    p.ks = pl.ks;
    if (pl.ks > 1 && !take_scratch(o, pl, mb, tiles, p)) return ZL_ESHAPE;
    return launch_slab_any<false>(p, pl, mb, hs);
}

// internal (zl_w4a16_qkv_rope_scatter): the fused qkv projection of a decode step with the neox rotation and the KV scatter in
// the epilogue; d % 32 == 0 (a workgroup's two tiles are a column block and its rotation partners)
int zl_w4a16_gemm_slab_rope(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                            uint32_t meta_bytes, const uint16_t* bias, int m, int n, int k, int groups, int tiles, const float* cosv,
                            const float* sinv, const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                            uint16_t* const* v_bufs, uint16_t* q_out, int h, int hkv, int d, int bshd, const uint16_t* norm_w,
                            float norm_eps, const zl_w4_opts_t* opts, hipStream_t hs) {
    static const zl_w4_opts_t kNoOpts = {};
    const zl_w4_opts_t& o = opts ? *opts : kNoOpts;
    if (!slab_shape_ok(m, k, groups, tiles, ldx, qw_bytes)) return ZL_ESHAPE;
    if (d % 32 != 0 || n != (h + 2 * hkv) * d || tiles * 16 != n) return ZL_ESHAPE;
    if (norm_w && !norm_ok(o, k)) return ZL_ESHAPE;
    SlabPlan pl;
    if (!plan_slab(m, tiles, groups, o.scratch != nullptr, true, norm_w != nullptr, o, &pl)) return ZL_ESHAPE;
    SlabParams p = {};
    fill_common(p, x, ldx, qw, meta, qw_bytes, meta_bytes, bias, nullptr, nullptr, m, n, k, groups, tiles, bias ? ZL_EPI_BIAS : 0, n);
    if (norm_w) { p.norm_w = norm_w; p.row_ss = o.row_ss; p.ss_parts = k / 16; p.eps = norm_eps; }
    p.cosv = cosv; p.sinv = sinv; p.placement = placement; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs;
    p.q_out = q_out; p.h = h; p.hkv = hkv; p.d = d; p.bshd = bshd; p.tstride = d / 32;
    const int mb = m <= 16 ? 1 : 2;
    p.ks = pl.ks;
    if (pl.ks > 1 && !take_scratch(o, pl, mb, tiles, p)) return ZL_ESHAPE;
    return launch_slab_any<true>(p, pl, mb, hs);
}

extern "C" int zl_row_ss(const uint16_t* x, int64_t ldx, int64_t m, int64_t k, float* out, zl_stream_t s) {
    ZL_CHECK_ARG(x && out && m > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 16 == 0 && ldx >= k && m <= 65535, ZL_ESHAPE);
    const int parts = (int)(k / 16);
    hipLaunchKernelGGL(k_row_ss, dim3((unsigned)((parts + 15) / 16), (unsigned)m), dim3(256), 0, (hipStream_t)s, x, ldx, (int)m, parts, out);
    return zl_launch_status();
}

// the dispatch questions of zl_w4a16_gemm_mfma_ex / zl_w4a16_qkv_rope_scatter_ex, asked without launching (w4_mfma.hip calls the
// launchers above under the same conditions)
bool zl_slab_route(int64_t m, int64_t n, int64_t k, int64_t group_size, bool rope, bool norm, bool silu, const zl_w4_opts_t* opts) {
    static const zl_w4_opts_t kNoOpts = {};
    const zl_w4_opts_t& o = opts ? *opts : kNoOpts;
    if (o.slab < 0 || group_size <= 0 || group_size % 128 != 0 || k % 128 != 0 || n % 16 != 0) return false;
    if (m < (o.slab_min_m > 0 ? o.slab_min_m : (rope || o.small_algo == 1 ? 5 : 3)) || m > 32) return false;
    if (norm && !(o.row_ss != nullptr && k % 1024 == 0 && k <= 8192 && k > 2048)) return false;
    if (rope && !opts) return false;
    (void)silu;
    SlabPlan pl;
    return plan_slab((int)m, (int)(n / 16), (int)(k / 128), o.scratch != nullptr, rope, norm, o, &pl);
}

extern "C" int zl_w4a16_emits_row_ss(int64_t m, int64_t n, int64_t k, int64_t group_size, int epilogue, const zl_w4_opts_t* opts) {
    if (!opts || !opts->row_ss_out || (epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32))) return 0;
    return zl_slab_route(m, n, k, group_size, false, false, false, opts) ? 1 : 0;
}

extern "C" int zl_w4a16_takes_row_ss(int64_t m, int64_t n, int64_t k, int64_t group_size, int rope, const zl_w4_opts_t* opts) {
    if (!opts || !opts->row_ss) return 0;
    return zl_slab_route(m, n, k, group_size, rope != 0, true, false, opts) ? 1 : 0;
}

#ifdef ZL_SLAB_PROBE
extern "C" int zl_debug_set_slab_probe(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(zl_sprobe_p), &p, sizeof(p)); }
#endif

// w4_phase.hip -- W4A16 GEMM for decode batches of 5..32 rows: the streaming kernel of w4_mfma.hip with the
// activations fed through LDS in K phases instead of being staged whole.
//
// Same arithmetic as k_w4a16_mfma (fp32 group sums on the matrix cores, one fp32 fma per group with the
// scale; see the header of w4_mfma.hip for the reference branch it corresponds to), same ZLW4M weight
// layout, same epilogues.  What changes is how x reaches the MFMA:
//   * k_w4a16_mfma stages all M x K activations in LDS before the first weight is used.  Up to 4 rows that
//     hides behind the first HBM round trip; 8-16 rows cost 1.4-3 us of exposed staging per launch, a K of
//     14336 does not fit at all (its fallback reads fragments from L2 per item: 14-18 us), and 32 rows never
//     fit (two passes, or the M-tiled kernel + split-K epilogue: 84 us per Llama-3-8B layer at M = 32).
//   * here K is cut into phases of 1024 k.  A workgroup (8 waves, one per CU) owns R row tiles; in a phase
//     wave w takes the w-th 128-k item of every tile, so the eight waves finish a phase together and ONE
//     LDS buffer pair (2 x 16 MB rows x 1024 k) serves any K.  Accumulators of all R tiles stay in
//     registers across the phases (R x MB x 4 VGPRs), the weight ring runs across phase boundaries.
//   * vmcnt retires in order, so an x load must be OLDER than every weight load that is still in flight
//     when the x data is needed, or waiting for it drains the ring: the loads of phase p are issued XP
//     phases ahead (XP * R >= D - 1, D = ring depth) and ride in registers until the end of phase p - 1,
//     when they are stored to the idle LDS buffer; one barrier per phase.
// MB = 1: M <= 16, MB = 2: M <= 32 (two accumulator sets share every dequantised weight fragment).
#include <type_traits>
#include "zl_common.h"
#include "w4_i8p_common.h"

#ifdef ZL_PHASE_PROBE
static int zl_probe_seq = 0;
#endif

namespace {

constexpr int kT = 512, kW = 8;
constexpr int kPK = 1024;            // k per phase = kW items of 128
constexpr int kXS = kPK + 8;         // LDS x row, halfs (padded: conflict-free b128 fragment reads)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

// ---- optional timeline probe (build with -DZL_PHASE_PROBE; tools/ubench/probe_phase.py) ---------------
#ifdef ZL_PHASE_PROBE
// Launches of this kernel are numbered on the host (PhaseParams::probe_id); the four launches zl_probe_sel .. +3 record.
__device__ unsigned long long* zl_probe_p = nullptr;  // [4 launches][2048 workgroups * 8 waves][8] wall-clock ticks (100 MHz)
__device__ int zl_probe_sel = 0;
// The slot pointer is resolved ONCE at kernel entry (ZL_PPROBE_INIT): resolving it per stamp needs a vector load and the
// vmcnt(0) behind it drains the weight ring, i.e. the stamp would move what it measures.
#define ZL_PPROBE_INIT()                                                                                      \
    unsigned long long* pp_ = nullptr;                                                                        \
    {                                                                                                         \
        const int pl_ = p.probe_id - zl_probe_sel;                                                            \
        if (zl_probe_p && (threadIdx.x & 63) == 0 && pl_ >= 0 && pl_ < 4 && blockIdx.x < 2048)                \
            pp_ = zl_probe_p + ((size_t)pl_ * 16384 + blockIdx.x * kW + (threadIdx.x >> 6)) * 8;              \
    }
#define ZL_PPROBE(slot)                                                                                       \
    do {                                                                                                      \
        if (pp_) pp_[slot] = wall_clock64();                                                                  \
    } while (0)
#else
#define ZL_PPROBE_INIT() do {} while (0)
#define ZL_PPROBE(slot) do {} while (0)
#endif

struct PhaseParams {
    const uint16_t* x;
    int64_t ldx;
    const uint4* qw;
    const uint32_t* meta;
    uint32_t qw_bytes, meta_bytes;
    const uint16_t* bias;
    const uint16_t* residual;
    uint16_t* y;
    int m, n, k;
    int groups;        // 128-k items per row tile
    int tiles;         // 16-row tiles
    int phases;        // ceil(groups / 8)
    int epi, ld_out;
    const uint16_t* norm_w;   // NORM instantiations: fused RMSNorm prologue (M <= 4, K <= 4096)
    float norm_eps;
    // ROPE instantiations (the fused qkv projection of a decode step: rotate q and k, scatter k / v into the ragged
    // KV buffers, q to its own buffer -- rope_qk_cache + copy_to_rag_buffer2 in the GEMV epilogue)
    const float* cosv;          // (M, D) neox tables
    const float* sinv;
    const int32_t* placement;   // (M)
    const int32_t* buf_lens;    // (M)
    uint16_t* const* k_bufs;
    uint16_t* const* v_bufs;
    uint16_t* q_out;            // (M, H * D)
    int h, hkv, d, bshd;
    int pair_stride;            // tiles between a column and its rotation partner: D / 32
    // KS > 1 instantiations: K split over KS adjacent workgroups (long K with 17..32 rows: halves / quarters the
    // activation bytes each workgroup pulls through L2); fp32 partials meet in ks_ws, the last arriver sums them in
    // split order and runs the epilogue
    float* ks_ws;               // [KS][M][N]
    int* ks_counter;            // one per tile group, zero between launches
    // MERGE instantiations (the attention output projection of a decode step): the activations are the split-KV
    // partials of the decode attention kernel (attention.hip workspace: [row][head][split][128 acc | max | sum] fp32);
    // the merge of k_decode_attn_combine runs in the prologue, so the step has one launch less.  Uses buf_lens above.
    const float* mg_ws;
    const int32_t* mg_valid_lens;
    int mg_split_len, mg_max_splits;
    // I8 instantiations (5..32 rows on the integer matrix cores): the activations arrive as digit planes made ONCE per
    // activation matrix by k_w4_planes (below) instead of fp16 rows staged and dequantised against by every workgroup
    const unsigned char* planes;   // [group][row block][digit 2 1 0][mfma 0 1][64 lanes][16 bytes]: A operands as they sit in registers
    const float* pconsts;          // [group][row block][16 rows][xscale, xscale * sum X]
    uint32_t planes_bytes, pconsts_bytes;
#ifdef ZL_PHASE_PROBE
    int probe_id;
#endif
};

// Ring depth.  Register-resident (NORM) staging has no ordering constraint between x and the ring, so the short streams
// of one or two tiles per workgroup (K = 4096: 4 / 8 items per wave) have EVERY item in flight from the start: with 2-3
// in flight they paid two to three dependent memory round trips (qkv 6.6 -> , o 4.6 ->  us at one row).
#ifndef ZL_PH_D1
#define ZL_PH_D1 3
#endif
#ifndef ZL_PH_D2
#define ZL_PH_D2 4
#endif
#ifndef ZL_PH_D7
#define ZL_PH_D7 7
#endif
constexpr int ring_depth(int r, bool norm = false) {
    return norm ? (r == 1 ? ZL_PH_D1 : r == 2 ? ZL_PH_D2 : r == 3 ? 6 : r == 4 ? 8 : r == 7 ? ZL_PH_D7 : r)
                : (r == 1 ? 3 : r == 2 ? 4 : r == 3 ? 6 : r == 4 ? 8 : r);
}
constexpr int x_ahead(int r) { return r <= 4 ? 2 : 1; }
constexpr int gcd_(int a, int b) { return b == 0 ? a : gcd_(b, a % b); }
constexpr int lcm_(int a, int b) { return a / gcd_(a, b) * b; }

__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask_s, uint32_t magic_v) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask_s), "v"(magic_v));
    return r;
}

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

struct Guard { static constexpr bool value = true; };
struct NoGuard { static constexpr bool value = false; };

// NORM (MB = 1, M <= 4, K <= 4096): the whole activation block is register-resident from the start (thread t holds
// halfs 8 t .. 8 t + 7 of every row, the stand-alone RMSNorm kernel's assignment and summation order, so the
// normalised values are bit-identical to a separate zl_rmsnorm launch); the thread quarter that holds phase p's
// slice stores it, normalised, at the end of phase p - 1.
// ROPE (R = 2): the workgroup's two tiles are a column block and its rotation partners (D/2 columns = D/32 tiles
// further), so the neox rotation of q and k happens in the epilogue on the fp16-rounded projection outputs, with the
// roundings of the separate kernels (rope_common.cuh:14-34: one rounding to T after the fp32 rotation).
// MERGE (NORM staging, M <= 4, K = heads x 128 <= 4096, <= 16 splits): x row m = the merged decode-attention splits of
// task m, computed per thread for its own 8 halfs with k_decode_attn_combine's arithmetic and order (bit-identical).
// One workgroup per CU (the grid has at most one generation), so the prologue may hold 16 splits x 8 floats in VGPRs
// and have every load of a row in flight at once.
// I8 (rows 5..32, no NORM / MERGE): integer digit planes from memory (PhaseParams::planes) instead of fp16 rows through LDS.
// A wave's A operands of a phase -- its own group, all row blocks, three digits, two MFMAs: 6 MB KiB -- are loaded straight into
// registers in MFMA layout one phase ahead (two sets), so there is NO activation staging, NO LDS traffic in the stream and NO
// phase barrier: the eight waves only meet at the final reduction.  Per item: 8 VALU expand the nibbles to bytes (shared by all
// rows), 6 MB v_mfma_i32_16x16x64_i8, and per row of the lane 6 VALU turn the three exact digit sums into the fp32 group value
//     xscale * (65536 D2 + 256 D1 + D0) - z * xscale * sum X          (w4_i8p.hip's arithmetic; the group constants per row
// come from the planes' side table, parked in LDS per wave at kernel entry), scale-accumulated one step later like the fp16 path.
// DN (rows 9..32 with a fused RMSNorm, no NORM / MERGE / I8 / K split): the norm DEFERRED.  x reaches a workgroup one K phase at a
// time, so the row's sum of squares is not known when a phase is staged; but y = W (x . w rs) = rs (W (x . w)) -- rs is one scalar
// per row.  Staging multiplies the phase's slice by the norm weight (one packed fp16 multiply: T(x w), a single rounding, where
// the stand-alone kernel rounds T(x rs w) once) and adds the slice's squares to the thread's row sums (v_dot2_f32_f16); after the
// last phase the 128 threads that staged a row meet in LDS, and the epilogue scales the fp32 totals by rs[row] before bias /
// rotation / activation.  Two stand-alone k_rmsnorm launches per layer at batch 32 (9 % of the step) go away;
// the result differs from norm-then-GEMM by the rounding of T(x w) against T(x rs w) -- the same 2^-11 relative noise per
// activation, at a different place -- not bit-identical, tests/test_gpu_w4.py::test_deferred_norm_rows_9_32 holds it to the
// output rounding.
template <int R, int MB, bool NORM, bool ROPE, int KS = 1, bool MERGE = false, bool I8 = false, bool DN = false>
__global__ __launch_bounds__(kT, (MERGE || I8) ? 1 : 2) void k_w4a16_phase(const PhaseParams p) {
    static_assert(!DN || (!NORM && !MERGE && !I8 && KS == 1), "deferred norm: the plain staged-activation instantiations");
    static_assert(!MERGE || (NORM && !ROPE && KS == 1), "split merge: register-resident staging");
    static_assert(!ROPE || R == 2, "fused rotary: a tile and its partner tile");
    static_assert(KS == 1 || (!ROPE && !NORM), "K split: plain / bias / residual epilogues only");
    static_assert(!I8 || (!NORM && !MERGE), "digit planes: the norm / merge happened where the planes were made");
    constexpr int D = I8 ? (R <= 2 ? R + 2 : R == 3 ? 4 : R) : ring_depth(R, NORM);
    constexpr int XP = I8 ? 2 : NORM ? 1 : x_ahead(R), BODY = lcm_(D, R * XP);
    static_assert(!NORM || MB == 1, "fused norm: one row block");
    constexpr int XC = 4 * MB;                       // 16-byte x chunks per thread per phase (16 MB rows x 128 chunks)
    constexpr int kBuf = MB * 16 * kXS;              // halfs per LDS phase buffer
    static_assert(BODY % R == 0 && (BODY / R) % XP == 0 && BODY % D == 0, "static ring / accumulator / x-set indices");
    static_assert(NORM || I8 || XP * R >= D - 2, "x loads must be older than the weights in flight when they are consumed");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem);
    ZL_PPROBE_INIT();
    ZL_PPROBE(0);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrow = lane & 15, kq = lane >> 4;
    // K split: workgroup (tile group tg, split ksi) streams phases ph0 .. ph0 + P - 1 of its R tiles
    const int ksi = KS > 1 ? (int)(blockIdx.x % KS) : 0, tg = KS > 1 ? (int)(blockIdx.x / KS) : (int)blockIdx.x;
    const int PP = (p.phases + KS - 1) / KS, ph0 = ksi * PP;
    const int P = KS > 1 ? max(0, min(p.phases, ph0 + PP) - ph0) : p.phases, total = P * R;
    // ROPE: workgroup b owns tiles {base, base + s}, base = (b / s) * 2 s + b % s  (s = pair_stride)
    const int tile0 = ROPE ? (blockIdx.x / p.pair_stride) * 2 * p.pair_stride + blockIdx.x % p.pair_stride : tg * R;
    const int tile_stride = ROPE ? p.pair_stride : 1;

    // ---- activations: chunk c of a thread = row (tid >> 7) + 4 c, halfs 8 (tid & 127) .. +7 of the phase
    const int xrow0 = threadIdx.x >> 7, xcc = (threadIdx.x & 127) * 8;
    uint4 xr[XP][XC], nwr[DN ? XP : 1];
    float ss[DN ? XC : 1];                           // DN: sum of squares of the slices this thread staged, per row xrow0 + 4 c
#pragma unroll
    for (int c = 0; c < (DN ? XC : 1); ++c) ss[c] = 0.f;
    auto load_x = [&](int set, int ph) {             // set: static
#pragma unroll
        for (int c = 0; c < XC; ++c) {
            const int row = xrow0 + 4 * c, kk = (ph0 + ph) * kPK + xcc;
            const bool live = row < p.m && kk < p.k && ph < P;
            const uint16_t* src = p.x + (live ? (size_t)row * p.ldx + kk : 0);
            xr[set][c] = *reinterpret_cast<const uint4*>(src);
            if (!live) xr[set][c] = make_uint4(0, 0, 0, 0);
        }
        if constexpr (DN) {
            const int kk = (ph0 + ph) * kPK + xcc;
            nwr[set] = *reinterpret_cast<const uint4*>(p.norm_w + ((kk < p.k && ph < P) ? kk : 0));
        }
    };
    auto store_x = [&](int set, int ph) {            // into buffer ph & 1
        uint16_t* dst = xs + (ph & 1) * kBuf + xcc;
        if constexpr (DN) {
            const uint32_t wu[4] = {nwr[set].x, nwr[set].y, nwr[set].z, nwr[set].w};
#pragma unroll
            for (int c = 0; c < XC; ++c) {
                uint32_t u[4] = {xr[set][c].x, xr[set][c].y, xr[set][c].z, xr[set][c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const hv2 hh = __builtin_bit_cast(hv2, u[e]);
                    ss[c] = __builtin_amdgcn_fdot2(hh, hh, ss[c], false);
                    u[e] = __builtin_bit_cast(uint32_t, hh * __builtin_bit_cast(hv2, wu[e]));   // v_pk_mul_f16: T(x w), one rounding
                }
                *reinterpret_cast<uint4*>(dst + (xrow0 + 4 * c) * kXS) = make_uint4(u[0], u[1], u[2], u[3]);
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < XC; ++c) *reinterpret_cast<uint4*>(dst + (xrow0 + 4 * c) * kXS) = xr[set][c];
    };
    constexpr int kNR = MERGE ? 4 : 8;               // rows the register-resident (NORM) variant carries
    uint4 xn4[kNR], nw4 = make_uint4(0, 0, 0, 0);   // NORM: rows 0..7, this thread's 8 halfs; the norm weight slice
    if constexpr (MERGE) {
#pragma unroll
        for (int m = 0; m < kNR; ++m) xn4[m] = make_uint4(0, 0, 0, 0);
    } else if constexpr (NORM) {
        const int idx = threadIdx.x * 8;
#pragma unroll
        for (int m = 0; m < kNR; ++m) {
            xn4[m] = make_uint4(0, 0, 0, 0);
            if (m < p.m) {                             // workgroup-uniform: no load instructions for rows that do not exist
                xn4[m] = *reinterpret_cast<const uint4*>(p.x + (idx < p.k ? (size_t)m * p.ldx + idx : 0));
                if (idx >= p.k) xn4[m] = make_uint4(0, 0, 0, 0);
            }
        }
        if (p.norm_w) nw4 = *reinterpret_cast<const uint4*>(p.norm_w + (idx < p.k ? idx : 0));
    } else if constexpr (!I8) {
#pragma unroll
        for (int q = 0; q < XP; ++q) load_x(q, q);
    }
    // I8: the A operands of phase 0 and this wave's group constants of every phase, ahead of the weight ring
    v4i A[I8 ? 2 : 1][MB][3][2];
    uint4 craw[2 * MB];
    const __amdgpu_buffer_rsrc_t rpl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.planes), 0, I8 ? p.planes_bytes : 0, 0x00020000);
    const uint32_t kOob = 0x80000000u;               // a lane offset no descriptor covers: the load returns zero and moves nothing
    auto load_A = [&](int set, int ph) {             // set: static
        const uint32_t g = (uint32_t)((ph0 + ph) * kW + wave);
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            const uint32_t vo = (b * 16 + nrow < p.m && ph < P) ? (uint32_t)lane * 16u : kOob;
#pragma unroll
            for (int jm = 0; jm < 6; ++jm)
                A[set][b][jm >> 1][jm & 1] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rpl, vo, ((g * MB + b) * 6u + jm) * 1024u, 0));
        }
    };
    if constexpr (I8) {
        load_A(0, 0);
        const __amdgpu_buffer_rsrc_t rpc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pconsts), 0, p.pconsts_bytes, 0x00020000);
#pragma unroll
        for (int c = 0; c < 2 * MB; ++c) {           // 16 phases x MB blocks x 8 pieces of 16 bytes = 128 MB pieces per wave
            const int idx = c * 64 + lane, ph = idx / (8 * MB), within = idx % (8 * MB);
            const uint32_t g = (uint32_t)((ph0 + ph) * kW + wave);
            craw[c] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rpc, ph < P ? g * (uint32_t)(MB * 128) + within * 16u : kOob, 0, 0));
        }
    }
    // ROPE: what this thread's epilogue needs from memory (rotation table entries, the task's slot, buffer length and buffer
    // pointer; 16 m <= 512 outputs, one per thread) is requested here, next to the activations and ahead of the weight ring:
    // in the epilogue they were two dependent round trips at the very end of the launch (w4_i8p.hip does the same)
    float rp_c0 = 0.f, rp_s0 = 0.f, rp_c1 = 0.f, rp_s1 = 0.f;
    int rp_place = -1, rp_blen = 0;
    uint16_t* rp_kv = nullptr;
    if constexpr (ROPE) {
        if ((int)threadIdx.x < 16 * p.m) {
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
    // the plain / bias / residual epilogues: the operands of this thread's FIRST output (its only one up to 512 outputs per
    // workgroup) are requested here as well -- one more round trip at the very end of the launch otherwise
    float ep_res0 = 0.f, ep_bias0 = 0.f;
    bool ep_pref = false;
    if constexpr (!ROPE && KS == 1) {
        if (!(p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32))) {
            const int per_tile_ = 16 * p.m, o_ = threadIdx.x;
            if (o_ < R * per_tile_) {
                const int r_ = o_ / per_tile_, rem_ = o_ % per_tile_, tile_ = tile0 + r_;
                const int row_ = tile_ * 16 + (rem_ & 15);
                if (tile_ < p.tiles && row_ < p.n) {
                    if (p.epi & ZL_EPI_RESIDUAL) ep_res0 = (float)__builtin_bit_cast(_Float16, p.residual[(size_t)(rem_ >> 4) * p.ld_out + row_]);
                    if ((p.epi & ZL_EPI_BIAS) && p.bias) ep_bias0 = (float)__builtin_bit_cast(_Float16, p.bias[row_]);
                    ep_pref = true;
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- weight ring.  Item sequence of wave w: for phase: for r < R: (tile0 + r, 8 phase + w); the byte
    //      offset lives in SGPRs (buffer soffset), advanced by one of two strides; out-of-range tiles read
    //      zeros (buffer bounds), items past K meet zero activations.
    uint4 wq[D];
    uint32_t mt[D];
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.qw), 0, p.qw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.meta), 0, p.meta_bytes, 0x00020000);
    const uint32_t it0 = (uint32_t)tile0 * (uint32_t)p.groups + (uint32_t)(ph0 * kW) + (uint32_t)wave;
    uint32_t qs = it0 * 1024u, ms = it0 * 64u;
    const int tile_step = tile_stride * p.groups, phase_step = kW - (R - 1) * tile_stride * p.groups;
    const uint32_t q_off = (uint32_t)lane * 16u, m_off = (uint32_t)nrow * 4u;
    int iss_left = total - 1;
    // Exhausted stream: the remaining ring refills go through a zero-length descriptor -- out-of-range buffer loads return
    // zeros without touching memory.  (Re-reading the last item instead, the first cut, kept D - 1 cache-hit loads per
    // wave in flight at the end of the kernel, and a wave does not retire before its loads have landed.)
    const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.qw), 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rqc = total > 0 ? rq : rnull, rmc = total > 0 ? rm : rnull;
    auto issue = [&](int slot, int r_of_item) {      // both static
        wq[slot] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rqc, q_off, qs, 2 /* nt */));
        mt[slot] = __builtin_amdgcn_raw_buffer_load_b32(rmc, m_off, ms, 2);
        const bool more = iss_left > 0;
        const int adv = more ? 1 : 0;
        --iss_left;
        const int d = adv * (r_of_item == R - 1 ? phase_step : tile_step);
        qs += (uint32_t)d * 1024u;
        ms += (uint32_t)d * 64u;
        rqc = more ? rqc : rnull;
        rmc = more ? rmc : rnull;
    };
#pragma unroll
    for (int s = 0; s < D - 1; ++s) {
        issue(s, s % R);
        __builtin_amdgcn_sched_barrier(0);
    }
    mt[D - 1] = 0;                                   // the neutral "previous item" of the first step
    wq[D - 1] = make_uint4(0, 0, 0, 0);
    ZL_PPROBE(1);

    auto store_norm = [&](int ph) {                  // the quarter of the workgroup that holds phase ph's k range
        if ((int)(threadIdx.x >> 7) == ph) {
            uint16_t* dst = xs + (ph & 1) * kBuf + xcc;
#pragma unroll
            for (int m = 0; m < kNR; ++m) {
                if (m < p.m) *reinterpret_cast<uint4*>(dst + m * kXS) = xn4[m];
            }
        }
    };
    unsigned char* cw = smem + (size_t)wave * (16 * MB * 128);   // I8: this wave's group constants, [phase][block][16 rows][2]
    if constexpr (I8) {
#pragma unroll
        for (int c = 0; c < 2 * MB; ++c) *reinterpret_cast<uint4*>(cw + (size_t)(c * 64 + lane) * 16) = craw[c];
    } else
    if constexpr (NORM) {
      // (norm_w == null: the register-resident staging alone -- up to 4 rows it beats the per-phase loads: 4.9 vs 5.1 us
      //  on the o projection at one row)
      if constexpr (MERGE) {   // after the ring issue: the first weights are in flight while the splits merge
        // Thread t merges its own 8 halfs (head t / 16, d = 8 (t % 16) .. + 7) straight into the staging registers.
        // (A variant with 128 contiguous bytes per head per load instruction and an LDS hand-over into this assignment
        //  measured the same kernel time and 1-2 % less end to end: profiles/r01_attn_merge_ab.txt.)
        constexpr int kS = 16, kWS = 128 + 2;        // splits held at once; floats per split record
        const int idx = threadIdx.x * 8, heads = p.k >> 7;
        const bool act = idx < p.k;
        const int head = act ? (int)(threadIdx.x >> 4) : 0, d0 = (threadIdx.x & 15) * 8;
#pragma unroll
        for (int m = 0; m < kNR; ++m) {
            if (m < p.m) {                             // workgroup-uniform
                const int elen = min(p.buf_lens[m], p.mg_valid_lens[m]);
                const int ns = min((elen + p.mg_split_len - 1) / p.mg_split_len, kS);
                const float* src = p.mg_ws + ((size_t)m * heads + head) * p.mg_max_splits * kWS;
                float2 st[kS], v[kS][4];
#pragma unroll
                for (int u = 0; u < kS; ++u) {
                    st[u] = make_float2(-1e20f, 0.f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = make_float2(0.f, 0.f);
                    if (u < ns) {                      // uniform: no loads for splits that do not exist
                        st[u] = *reinterpret_cast<const float2*>(src + (size_t)u * kWS + 128);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[u][e] = *reinterpret_cast<const float2*>(src + (size_t)u * kWS + d0 + 2 * e);
                    }
                }
                float mn = -1e20f;
#pragma unroll
                for (int u = 0; u < kS; ++u) mn = fmaxf(mn, st[u].x);
                float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, z = 0.f;
#pragma unroll
                for (int u = 0; u < kS; ++u) {
                    if (u < ns) {
                        const float f = __expf(st[u].x - mn);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[2 * e] = __builtin_fmaf(v[u][e].x, f, a[2 * e]);
                            a[2 * e + 1] = __builtin_fmaf(v[u][e].y, f, a[2 * e + 1]);
                        }
                        z = __builtin_fmaf(st[u].y, f, z);
                    }
                }
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hv2 hh;
                    hh.x = zl_f32_to_f16(a[2 * e] / (z + 1e-20f));
                    hh.y = zl_f32_to_f16(a[2 * e + 1] / (z + 1e-20f));
                    o[e] = __builtin_bit_cast(uint32_t, hh);
                }
                if (act) xn4[m] = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
      }
      if constexpr (!MERGE) if (p.norm_w) {
        // sum of squares: per-thread chain, 64-lane butterfly, waves in order (zl_block_sum's order).  Instantiated per
        // live row count so that the rows' shuffle chains sit in one basic block and interleave (a batch-1 step pays
        // for one row; with a loop exit per row four rows cost 1.5 us).
        float* scratch = reinterpret_cast<float*>(smem + 2 * (size_t)kBuf * 2);
        auto norm_rows = [&](auto nr_tag) {
            constexpr int NRL = decltype(nr_tag)::value;
            float part[NRL];
#pragma unroll
            for (int m = 0; m < NRL; ++m) {
                const uint32_t u[4] = {xn4[m].x, xn4[m].y, xn4[m].z, xn4[m].w};
                float run = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const hv2 hh = __builtin_bit_cast(hv2, u[e]);
                    run = __builtin_fmaf((float)hh.x, (float)hh.x, run);
                    run = __builtin_fmaf((float)hh.y, (float)hh.y, run);
                }
                part[m] = run;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
                for (int m = 0; m < NRL; ++m) part[m] += __shfl_xor(part[m], off, 64);   // zl_wave_sum's order, rows interleaved
            }
            if (lane == 0) {
#pragma unroll
                for (int m = 0; m < NRL; ++m) scratch[m * kW + wave] = part[m];
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < NRL; ++m) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < kW; ++w) tot += scratch[m * kW + w];
                const float rs = zl_rsqrt_rn(tot / (float)p.k + p.norm_eps);
                uint32_t u[4] = {xn4[m].x, xn4[m].y, xn4[m].z, xn4[m].w};
                const uint32_t wu[4] = {nw4.x, nw4.y, nw4.z, nw4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const hv2 hh = __builtin_bit_cast(hv2, u[e]), ww = __builtin_bit_cast(hv2, wu[e]);
                    hv2 o;
                    o.x = zl_f32_to_f16((float)hh.x * rs * (float)ww.x);
                    o.y = zl_f32_to_f16((float)hh.y * rs * (float)ww.y);
                    u[e] = __builtin_bit_cast(uint32_t, o);
                }
                xn4[m] = make_uint4(u[0], u[1], u[2], u[3]);
            }
        };
        switch (p.m) {                                // workgroup-uniform
            case 1: norm_rows(std::integral_constant<int, 1>{}); break;
            case 2: norm_rows(std::integral_constant<int, 2>{}); break;
            case 3: norm_rows(std::integral_constant<int, 3>{}); break;
            case 4: norm_rows(std::integral_constant<int, 4>{}); break;
            case 5: norm_rows(std::integral_constant<int, 5>{}); break;
            case 6: norm_rows(std::integral_constant<int, 6>{}); break;
            case 7: norm_rows(std::integral_constant<int, 7>{}); break;
            default: norm_rows(std::integral_constant<int, 8>{}); break;
        }
      }
        store_norm(0);
    } else {
        store_x(0, 0);
    }
    if constexpr (!I8) __syncthreads();
    ZL_PPROBE(2);

    const uint32_t mask_lo = __builtin_amdgcn_readfirstlane(0x000f000fu);
    const uint32_t mask_hi = __builtin_amdgcn_readfirstlane(0x00f000f0u);
    uint32_t magic = 0x64006400u;
    asm volatile("" : "+v"(magic));
    // A fragment (activations): row m = 16 mb + (lane & 15), k = 128 wave + 32 t + 8 kq .. +7 of the phase
    const uint16_t* xl = xs + nrow * kXS + wave * 128 + 8 * kq;

    f4 acc[R][MB], accg_prev[MB];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[r][b] = (f4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < MB; ++b) accg_prev[b] = (f4){0.f, 0.f, 0.f, 0.f};

    auto finish_prev = [&](int rp, int pslot) {      // acc[rp] += scale * group sum of the previous item
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
    // one ring step (same order as k_w4a16_mfma: dequantise, scale-accumulate the previous item, MFMAs back to back)
    auto step = [&](int slot, int pslot, int rp, int r_issue, int ph) {
#ifdef ZL_EXP_NOCOMPUTE   // ablation: the stream structure alone (no LDS fragment reads, no dequant, no MFMA)
        {
            const uint32_t v = wq[slot].x ^ wq[slot].y ^ wq[slot].z ^ wq[slot].w ^ mt[slot];
            acc[rp][0][0] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, acc[rp][0][0]) ^ v);
            issue(pslot, r_issue);
            (void)ph;
            return;
        }
#endif
        const uint16_t* xb = xl + (ph & 1) * kBuf;
        uint4 bv[MB][4];
#pragma unroll
        for (int b = 0; b < MB; ++b) {
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[b][t] = *reinterpret_cast<const uint4*>(xb + b * 16 * kXS + 32 * t);
        }
#ifdef ZL_EXP_NODEQ       // ablation: LDS reads + MFMAs kept, the nibble extraction / zero subtraction dropped
        {
            h8 a0[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint4 u = make_uint4(wq[slot].x + t, wq[slot].y, wq[slot].z, wq[slot].w ^ mt[slot]);
                a0[t] = __builtin_bit_cast(h8, u);
            }
            finish_prev(rp, pslot);
            __builtin_amdgcn_sched_barrier(0);
            f4 accg0[MB];
#pragma unroll
            for (int b = 0; b < MB; ++b) accg0[b] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int b = 0; b < MB; ++b)
                    accg0[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, bv[b][t]), a0[t], accg0[b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < MB; ++b) accg_prev[b] = accg0[b];
            issue(pslot, r_issue);
            return;
        }
#endif
        const hv2 z1 = __builtin_bit_cast(hv2, __builtin_amdgcn_perm(mt[slot], mt[slot], 0x03020302u));
        const hv2 c960 = {(_Float16)960.f, (_Float16)960.f};
        const hv2 z16 = z1 + c960;
        const uint32_t wds[4] = {wq[slot].x, wq[slot].y, wq[slot].z, wq[slot].w};
        h8 a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = dequant_word(wds[t], z1, z16, mask_lo, mask_hi, magic);
        finish_prev(rp, pslot);
        __builtin_amdgcn_sched_barrier(0);
        f4 accg[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) accg[b] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int b = 0; b < MB; ++b)
                accg[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, bv[b][t]), a[t], accg[b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < MB; ++b) accg_prev[b] = accg[b];
        issue(pslot, r_issue);
    };
    // the integer step: same ring / accumulator discipline, the group value formed from exact digit sums
    float cst[MB][8];
    const uint32_t m4 = __builtin_amdgcn_readfirstlane(0x0f0f0f0fu);
    auto step_i8 = [&](int slot, int pslot, int rp, int r_issue, int set) {
        const uint4 w = wq[slot];
        const uint32_t mw = mt[slot];
        v4i b0, b1;
        b0[0] = (int)(w.x & m4); b0[1] = (int)((w.x >> 4) & m4); b0[2] = (int)(w.y & m4); b0[3] = (int)((w.y >> 4) & m4);
        b1[0] = (int)(w.z & m4); b1[1] = (int)((w.z >> 4) & m4); b1[2] = (int)(w.w & m4); b1[3] = (int)((w.w >> 4) & m4);
        finish_prev(rp, pslot);
        __builtin_amdgcn_sched_barrier(0);
        const v4i zero4 = (v4i){0, 0, 0, 0};
        v4i d[MB][3];
#pragma unroll
        for (int b = 0; b < MB; ++b) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                d[b][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[set][b][j][0], b0, zero4, 0, 0, 0);
                d[b][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[set][b][j][1], b1, d[b][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const hv2 sm = __builtin_bit_cast(hv2, mw);              // .x = scale, .y = -(1024 + zero)
        const float zf = (float)sm.y + 1024.f;                    // -zero, exact
#pragma unroll
        for (int b = 0; b < MB; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f12 = (float)((d[b][1][i] << 8) + d[b][2][i]);
                const float u = __builtin_fmaf((float)d[b][0][i], 65536.f, f12);
                accg_prev[b][i] = __builtin_fmaf(zf, cst[b][2 * i + 1], u * cst[b][2 * i]);
            }
        }
        // the B registers stay allocated until the results have been read (w4_i8p.hip: a VALU write one slot behind the MFMA
        // reached the rows of its last pass)
        asm volatile("" : "+v"(accg_prev[0][0]) : "v"(b0), "v"(b1));
        issue(pslot, r_issue);
    };
    // BODY steps starting at a phase boundary (k, ph); GUARD: the stream may end inside (wave-uniform tests)
    auto body = [&](int k, int ph, auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
#pragma unroll
        for (int s = 0; s < BODY; ++s) {
            if (GUARD && k + s >= total) break;
            const int r = s % R, j = s / R;          // static
            const int php = ph + j;
            if constexpr (I8) {
                if (r == 0) {
                    load_A((j + 1) & 1, php + 1);    // the other set: last read by phase php - 1
#pragma unroll
                    for (int b = 0; b < MB; ++b) {
                        const float4 c0 = *reinterpret_cast<const float4*>(cw + (size_t)(php * MB + b) * 128 + kq * 32);
                        const float4 c1 = *reinterpret_cast<const float4*>(cw + (size_t)(php * MB + b) * 128 + kq * 32 + 16);
                        cst[b][0] = c0.x; cst[b][1] = c0.y; cst[b][2] = c0.z; cst[b][3] = c0.w;
                        cst[b][4] = c1.x; cst[b][5] = c1.y; cst[b][6] = c1.z; cst[b][7] = c1.w;
                    }
                }
                step_i8(s % D, (s + D - 1) % D, (s + R - 1) % R, (s + D - 1) % R, j & 1);
                continue;
            }
            if (!NORM && r == 0) load_x(j % XP, php + XP);   // set (phase % XP): free since the end of phase php - 1
            step(s % D, (s + D - 1) % D, (s + R - 1) % R, (s + D - 1) % R, php);
#ifdef ZL_PHASE_PROBE
            if (k == 0 && s == 0) ZL_PPROBE(3);
#endif
#ifndef ZL_EXP_NOPHASE     // ablation: no per-phase staging / barrier
            if (r == R - 1) {
                if (php + 1 < P) {                    // workgroup-uniform
                    if constexpr (NORM) store_norm(php + 1);
                    else store_x((j + 1) % XP, php + 1);
                    __syncthreads();
                }
            }
#endif
        }
    };
    if (total >= BODY) {
        int k = 0, ph = 0;
#pragma unroll 1
        do {
            body(k, ph, NoGuard{});
            k += BODY;
            ph += BODY / R;
        } while (k + BODY <= total);
        body(k, ph, Guard{});
    } else {
        body(0, 0, Guard{});
    }
    {   // the last item's scale-accumulate (static register indices only)
        const int last = (total - 1) % BODY;
#pragma unroll
        for (int s = 0; s < BODY; ++s) {
            if (last == s) finish_prev(s % R, s % D);
        }
    }
    ZL_PPROBE(4);
    // DN: the row sums meet here -- 64-lane butterfly, then the two waves of a row quarter side by side in LDS (behind the
    // x / reduction area); the barriers below order them before the epilogue reads
    constexpr size_t kMainLds = (2 * (size_t)MB * 16 * kXS * 2 > (size_t)R * MB * kW * 64 * 16) ? 2 * (size_t)MB * 16 * kXS * 2 : (size_t)R * MB * kW * 64 * 16;
    float* dn_scr = reinterpret_cast<float*>(smem + kMainLds);       // [16 MB rows][2 waves]
    if constexpr (DN) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int c = 0; c < XC; ++c) ss[c] += __shfl_xor(ss[c], off, 64);
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < XC; ++c) dn_scr[(xrow0 + 4 * c) * 2 + (wave & 1)] = ss[c];
        }
    }
    auto dn_rs = [&](int m) -> float {               // rs of row m (DN only)
        return zl_rsqrt_rn((dn_scr[2 * m] + dn_scr[2 * m + 1]) / (float)p.k + p.norm_eps);
    };
    __syncthreads();                                  // every wave is done with the x buffers: reuse them

    // ---- park the partial C fragments, reduce over the 8 waves in fixed order, epilogue
    f4* red = reinterpret_cast<f4*>(smem);
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int b = 0; b < MB; ++b) red[((r * MB + b) * kW + wave) * 64 + lane] = acc[r][b];
    }
    __syncthreads();
    ZL_PPROBE(5);
    const float* redf = reinterpret_cast<const float*>(red);
    if constexpr (KS > 1) {
        // partials: agent-scope (write-through) stores; the barrier below waits for them (vmcnt(0)); one agent-scope
        // atomic per workgroup elects the last arriver, which reads all KS partials back with agent-scope loads
        auto total_of = [&](int r, int n_local, int m) {
            const int b = m >> 4, ln = ((m & 15) >> 2) * 16 + n_local, i = m & 3;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < kW; ++w) v += redf[(((size_t)(r * MB + b) * kW + w) * 64 + ln) * 4 + i];
            return v;
        };
        const int nouts = R * 16 * p.m;
        for (int o = threadIdx.x; o < nouts; o += kT) {
            const int r = o / (16 * p.m), rem = o % (16 * p.m);
            const int m = rem >> 4, n_local = rem & 15, col = (tile0 + r) * 16 + n_local;
            if (col < p.n) __hip_atomic_store(p.ks_ws + ((size_t)ksi * p.m + m) * p.n + col, total_of(r, n_local, m), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(p.ks_counter + tg, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old == KS - 1) __hip_atomic_store(p.ks_counter + tg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = old == KS - 1;
        }
        __syncthreads();
        if (!*flag) return;
        for (int o = threadIdx.x; o < nouts; o += kT) {
            const int r = o / (16 * p.m), rem = o % (16 * p.m);
            const int m = rem >> 4, n_local = rem & 15, row = (tile0 + r) * 16 + n_local;
            if (row >= p.n) continue;
            float v = 0.f;
#pragma unroll
            for (int sp = 0; sp < KS; ++sp)
                v += __hip_atomic_load(p.ks_ws + ((size_t)sp * p.m + m) * p.n + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const size_t orow = (size_t)m * p.ld_out;
            const float bb = ((p.epi & ZL_EPI_BIAS) && p.bias) ? (float)__builtin_bit_cast(_Float16, p.bias[row]) : 0.f;
            float ov;
            if (p.epi & ZL_EPI_ADD_C) ov = ((float)__builtin_bit_cast(_Float16, p.y[orow + row]) + v) + bb;
            else ov = v + bb;
            _Float16 y16 = zl_f32_to_f16(ov);
            if (p.epi & ZL_EPI_RESIDUAL)
                y16 = zl_f32_to_f16((float)__builtin_bit_cast(_Float16, p.residual[orow + row]) + (float)y16);
            p.y[orow + row] = __builtin_bit_cast(uint16_t, y16);
        }
        return;
    }
    if constexpr (ROPE) {
        const int half = p.d / 2;
        for (int o = threadIdx.x; o < 16 * p.m; o += kT) {
            const int m = o >> 4, n_local = o & 15;
            const int b = m >> 4, ln = ((m & 15) >> 2) * 16 + n_local, i = m & 3;
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int w = 0; w < kW; ++w) {
                v0 += redf[(((size_t)(0 * MB + b) * kW + w) * 64 + ln) * 4 + i];
                v1 += redf[(((size_t)(1 * MB + b) * kW + w) * 64 + ln) * 4 + i];
            }
            const int n0 = tile0 * 16 + n_local, n1 = n0 + half;       // columns of the fused qkv row
            if constexpr (DN) {
                const float rs = dn_rs(m);
                v0 *= rs;
                v1 *= rs;
            }
            if ((p.epi & ZL_EPI_BIAS) && p.bias) {
                v0 += (float)__builtin_bit_cast(_Float16, p.bias[n0]);
                v1 += (float)__builtin_bit_cast(_Float16, p.bias[n1]);
            }
            const float a = (float)zl_f32_to_f16(v0), bb = (float)zl_f32_to_f16(v1);   // the projection's fp16 outputs
            const int head = n0 / p.d, dcol = n0 % p.d;                 // dcol < half
            // (more than 32 outputs per thread never happens: 16 m <= 512; the prologue's table entries belong to o = threadIdx.x)
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
        ZL_PPROBE(6);
        return;
    }
    const bool silu = (p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
    const int per_tile = (silu ? 8 : 16) * p.m;
    const int nouts = R * per_tile;
    for (int o = threadIdx.x; o < nouts; o += kT) {
        const int r = o / per_tile, rem = o % per_tile;
        const int tile = tile0 + r;
        if (tile >= p.tiles) continue;
        auto total_of = [&](int n_local, int m) {
            const int b = m >> 4, ln = ((m & 15) >> 2) * 16 + n_local, i = m & 3;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < kW; ++w) v += redf[(((size_t)(r * MB + b) * kW + w) * 64 + ln) * 4 + i];
            return v;
        };
        if (!silu) {
            const int m = rem >> 4, n_local = rem & 15;
            const int row = tile * 16 + n_local;
            if (row < p.n) {
                float v = total_of(n_local, m);
                if constexpr (DN) v *= dn_rs(m);
                const size_t orow = (size_t)m * p.ld_out;
                const bool pre = ep_pref && o == (int)threadIdx.x;     // operands that came with the prologue
                const float bb = pre ? ep_bias0 : ((p.epi & ZL_EPI_BIAS) && p.bias) ? (float)__builtin_bit_cast(_Float16, p.bias[row]) : 0.f;
                float ov;
                if (p.epi & ZL_EPI_ADD_C) ov = ((float)__builtin_bit_cast(_Float16, p.y[orow + row]) + v) + bb;
                else ov = v + bb;
                _Float16 y16 = zl_f32_to_f16(ov);
                if (p.epi & ZL_EPI_RESIDUAL)
                    y16 = zl_f32_to_f16((pre ? ep_res0 : (float)__builtin_bit_cast(_Float16, p.residual[orow + row])) + (float)y16);
                p.y[orow + row] = __builtin_bit_cast(uint16_t, y16);
            }
        } else {
            const int m = rem >> 3, j = rem & 7;
            const int pr = tile * 8 + j;
            if (2 * pr + 1 < p.n) {
                float g = total_of(2 * j, m), u = total_of(2 * j + 1, m);
                if constexpr (DN) {
                    const float rs = dn_rs(m);
                    g *= rs;
                    u *= rs;
                }
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
    ZL_PPROBE(6);
}

template <int R, int MB, bool NORM, bool ROPE = false, int KS = 1, bool MERGE = false, bool I8 = false, bool DN = false>
int launch_phase(const PhaseParams& p, int grid, hipStream_t hs) {
    constexpr size_t x_bytes = I8 ? (size_t)kW * 16 * MB * 128 : 2 * (size_t)MB * 16 * kXS * 2 + (NORM ? 8 * kW * 4 : 0);
    constexpr size_t red_bytes = (size_t)R * MB * kW * 64 * 16;
    constexpr size_t lds = (x_bytes > red_bytes ? x_bytes : red_bytes) + (DN ? 16 * MB * 2 * 4 : 0);   // DN: the row sums behind both
    static_assert(lds <= 160 * 1024, "LDS");
    if (lds > 64 * 1024) {
        // every launch: the attribute is per device, and one process may drive several
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_w4a16_phase<R, MB, NORM, ROPE, KS, MERGE, I8, DN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ZL_ELIMIT;
    }
#ifdef ZL_PHASE_PROBE
    PhaseParams pp = p;
    pp.probe_id = zl_probe_seq++;     // host-side launch number (all instantiations share it)
    hipLaunchKernelGGL((k_w4a16_phase<R, MB, NORM, ROPE, KS, MERGE, I8, DN>), dim3(grid), dim3(kT), lds, hs, pp);
#else
    hipLaunchKernelGGL((k_w4a16_phase<R, MB, NORM, ROPE, KS, MERGE, I8, DN>), dim3(grid), dim3(kT), lds, hs, p);
#endif
    return zl_launch_status();
}

#ifdef ZL_EXPERIMENTAL
// ---- digit planes of an activation matrix (the producer side of the I8 instantiations; experimental build only) -------------
// One workgroup per row.  Octet o of the row (8 consecutive k) belongs to group o / 16; a DPP row of 16 lanes holds one group,
// so the group's largest magnitude and its sum of integers are row rotations.  Arithmetic = w4_i8p.hip's conversion slot:
// X = rint(x * 2^(36 - Ef)) (Ef = exponent field of the group's largest fp16 magnitude, |X| < 2^22), balanced byte digits
// X = 65536 b2 + 256 b1 + b0 read off Y = X + 0x808080; xscale = 2^(Ef - 36), NaN for a group that holds an inf / NaN.
// NORM: RMSNorm of the row first (LayerNorm::forward, src/nn/layernorm/layernorm.cu:10-42; T(x * rs * w) like the fused
// prologues), so the normalised row is never written anywhere.  NO = octets per thread (K <= 4096 NO).
template <int NO>
__global__ __launch_bounds__(512) void k_w4_planes(const uint16_t* __restrict__ x, int64_t ldx, int m, int k,
                                                   const uint16_t* __restrict__ norm_w, float norm_eps,
                                                   unsigned char* __restrict__ planes, float* __restrict__ pconsts, int mb) {
    __shared__ float red[8];
    const int row = blockIdx.x, b = row >> 4, rib = row & 15, groups = k >> 7;
    if (row >= m) {                                   // a row block's unused rows: neutral constants (their planes are never read)
        for (int g = threadIdx.x; g < groups; g += 512)
            *reinterpret_cast<float2*>(pconsts + (((size_t)g * mb + b) * 16 + rib) * 2) = make_float2(0.f, 0.f);
        return;
    }
    uint4 xr[NO], nw[NO];
    float part = 0.f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const int oct = threadIdx.x + 512 * o;
        const bool live = oct * 8 < k;
        xr[o] = *reinterpret_cast<const uint4*>(x + (live ? (size_t)row * ldx + oct * 8 : 0));
        if (!live) xr[o] = make_uint4(0, 0, 0, 0);
        nw[o] = norm_w ? *reinterpret_cast<const uint4*>(norm_w + (live ? oct * 8 : 0)) : make_uint4(0, 0, 0, 0);
    }
    if (norm_w) {                                     // kernel-uniform
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const uint32_t u[4] = {xr[o].x, xr[o].y, xr[o].z, xr[o].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const hv2 hh = __builtin_bit_cast(hv2, u[e]);
                part = __builtin_fmaf((float)hh.x, (float)hh.x, part);
                part = __builtin_fmaf((float)hh.y, (float)hh.y, part);
            }
        }
        const float rs = zl_rsqrt_rn(zl_block_sum(part, red) / (float)k + norm_eps);
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            uint32_t u[4] = {xr[o].x, xr[o].y, xr[o].z, xr[o].w};
            const uint32_t wu[4] = {nw[o].x, nw[o].y, nw[o].z, nw[o].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const hv2 hh = __builtin_bit_cast(hv2, u[e]), ww = __builtin_bit_cast(hv2, wu[e]);
                hv2 q;
                q.x = zl_f32_to_f16((float)hh.x * rs * (float)ww.x);
                q.y = zl_f32_to_f16((float)hh.y * rs * (float)ww.y);
                u[e] = __builtin_bit_cast(uint32_t, q);
            }
            xr[o] = make_uint4(u[0], u[1], u[2], u[3]);
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const int oct = threadIdx.x + 512 * o, g = oct >> 4, uo = oct & 15;
        const bool live = oct * 8 < k;                // whole DPP rows are live or not (k % 128 == 0)
        const uint32_t u[4] = {xr[o].x, xr[o].y, xr[o].z, xr[o].w};
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        const us2 m01 = __builtin_elementwise_max(__builtin_bit_cast(us2, u[0] & 0x7fff7fffu), __builtin_bit_cast(us2, u[1] & 0x7fff7fffu));
        const us2 m23 = __builtin_elementwise_max(__builtin_bit_cast(us2, u[2] & 0x7fff7fffu), __builtin_bit_cast(us2, u[3] & 0x7fff7fffu));
        const us2 mm = __builtin_elementwise_max(m01, m23);
        int am = max((int)mm.x, (int)mm.y);
        am = row16_max(am);
        const int ef = min(am >> 10, 30);
        const float up = __builtin_bit_cast(float, (uint32_t)(163 - ef) << 23);          // 2^(36 - Ef)
        const float xscale = am >= 0x7c00 ? __builtin_bit_cast(float, 0x7fc00000u) : __builtin_bit_cast(float, (uint32_t)(91 + ef) << 23);
        uint32_t Y[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const hv2 hh = __builtin_bit_cast(hv2, u[e]);
            Y[2 * e] = (uint32_t)__builtin_fmaf((float)hh.x, up, 8421504.f);
            Y[2 * e + 1] = (uint32_t)__builtin_fmaf((float)hh.y, up, 8421504.f);
        }
        int sx = (int)(((Y[0] + Y[1]) + (Y[2] + Y[3])) + ((Y[4] + Y[5]) + (Y[6] + Y[7]))) - 8 * 0x808080;
        sx = row16_sum(sx);
        // byte gather: k offsets [0 4 1 5 | 2 6 3 7] of the octet, the order (w & 0x0f0f0f0f | (w >> 4) & 0x0f0f0f0f) leaves the nibbles in
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
        if (live) {
            // octet uo: MFMA uo / 8, lane quarter kq = uo % 4, half (uo / 4) % 2 of the lane's 16 bytes
            unsigned char* dst = planes + ((((size_t)g * mb + b) * 6 + (uo >> 3)) * 64 + (uo & 3) * 16 + rib) * 16 + ((uo >> 2) & 1) * 8;
            *reinterpret_cast<uint2*>(dst) = make_uint2(a2, b2);                  // digit 2: tile pair 0
            *reinterpret_cast<uint2*>(dst + 2 * 1024) = make_uint2(a1, b1);       // digit 1
            *reinterpret_cast<uint2*>(dst + 4 * 1024) = make_uint2(a0, b0);       // digit 0
            if (uo == 0)
                *reinterpret_cast<float2*>(pconsts + (((size_t)g * mb + b) * 16 + rib) * 2) = make_float2(xscale, xscale * (float)sx);
        }
    }
}

#endif  // ZL_EXPERIMENTAL

}  // namespace

#ifdef ZL_PHASE_PROBE
extern "C" int zl_debug_probe_seq(void) { return zl_probe_seq; }
extern "C" int zl_debug_set_probe_p(void* p, int sel) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(zl_probe_p), &p, sizeof(p));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(zl_probe_sel), &sel, sizeof(sel));
    return (int)e;
}
#endif

// internal (called by zl_w4a16_gemm_mfma): 1 <= m <= 32; norm_w != null (fused RMSNorm): m <= 4 and k <= 4096.
// rounds_override: 0 = pick
int zl_w4a16_gemm_phase(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                        uint32_t meta_bytes, const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n,
                        int k, int groups, int tiles, int epilogue, int ld_out, const uint16_t* norm_w, float norm_eps,
                        const zl_w4_opts_t* opts, hipStream_t hs) {
    static const zl_w4_opts_t kNoOpts = {};
    const zl_w4_opts_t& o = opts ? *opts : kNoOpts;
    const int rounds_override = o.phase_rounds;
    if (norm_w && m <= 8 && k > 4096) return ZL_ESHAPE;        // register-resident staging; 9..32 rows: the deferred norm (DN), any K
    if (norm_w && m > 32) return ZL_ESHAPE;
    PhaseParams p = {};
    p.x = x; p.ldx = ldx; p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes;
    p.meta_bytes = meta_bytes; p.bias = bias; p.residual = residual; p.y = y; p.m = m; p.n = n; p.k = k;
    p.groups = groups; p.tiles = tiles; p.phases = (groups + kW - 1) / kW; p.epi = epilogue; p.ld_out = ld_out; p.norm_w = norm_w; p.norm_eps = norm_eps;
    p.cosv = p.sinv = nullptr; p.placement = p.buf_lens = nullptr; p.k_bufs = p.v_bufs = nullptr; p.q_out = nullptr;
    p.h = p.hkv = p.d = p.bshd = 0; p.pair_stride = 1;
    p.ks_ws = nullptr; p.ks_counter = nullptr;
    p.mg_ws = nullptr; p.mg_valid_lens = nullptr; p.mg_split_len = p.mg_max_splits = 0;
    {   // long K with 13..32 rows: K split over 2 or 4 adjacent workgroups (R = KS tiles each, same grid size); the fp32
        // partials and the arrival counters live in the caller's scratch (zl_w4_opts_t)
        const int ksplit = o.phase_ksplit ? o.phase_ksplit : 2;
        const bool plain = !(epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) && !norm_w;
        // (K = 14336: 16 rows 14.6 vs 15.9 us, 8 rows 13.4 vs 12.9 us -> from 13 rows on)
        const int ksplit_min_m = o.phase_ksplit_min_m > 0 ? o.phase_ksplit_min_m : 13;
        const int64_t need = ZL_SCRATCH_HEADER + (int64_t)ksplit * m * n * (int64_t)sizeof(float);
        if ((ksplit == 2 || ksplit == 4) && m >= ksplit_min_m && k > 8192 && plain && tiles / ksplit <= 16384 && o.scratch &&
            o.scratch_bytes >= need) {
            p.ks_counter = reinterpret_cast<int*>(o.scratch);
            p.ks_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(o.scratch) + ZL_SCRATCH_HEADER);
            const int grid = (tiles + ksplit - 1) / ksplit * ksplit;
            if (m <= 16) return ksplit == 2 ? launch_phase<2, 1, false, false, 2>(p, grid, hs) : launch_phase<4, 1, false, false, 4>(p, grid, hs);
            return ksplit == 2 ? launch_phase<2, 2, false, false, 2>(p, grid, hs) : launch_phase<4, 2, false, false, 4>(p, grid, hs);
        }
    }
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    // tiles per workgroup: one generation of workgroups when 8 tiles per CU suffice, else full-size workgroups
    int r = (tiles + cus - 1) / cus;
    if (r > 8) r = 8;
    if (rounds_override > 0 && rounds_override <= 8) r = rounds_override;
    const int grid = (tiles + r - 1) / r;
    const int mb = m <= 16 ? 1 : 2;
#define ZL_PH(RR)                                                                                              \
    case RR:                                                                                                   \
        if (norm_w && m > 8)                                                                                   \
            return mb == 1 ? launch_phase<RR, 1, false, false, 1, false, false, true>(p, grid, hs)             \
                           : launch_phase<RR, 2, false, false, 1, false, false, true>(p, grid, hs);            \
        if (norm_w || (m <= 4 && k <= 4096)) return launch_phase<RR, 1, true>(p, grid, hs);                    \
        return mb == 1 ? launch_phase<RR, 1, false>(p, grid, hs) : launch_phase<RR, 2, false>(p, grid, hs);
    switch (r) {
        ZL_PH(1) ZL_PH(2) ZL_PH(3) ZL_PH(4) ZL_PH(5) ZL_PH(6) ZL_PH(7) ZL_PH(8)
    }
#undef ZL_PH
    return ZL_EINVAL;
}

// internal (called by zl_w4a16_qkv_rope_scatter): the fused qkv projection of a decode step with the neox rotation and
// the KV scatter in the epilogue.  n = (h + 2 hkv) * d, d % 32 == 0, 1 <= m <= 32, norm_w != null: m <= 4 and k <= 4096.
int zl_w4a16_gemm_phase_rope(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                             uint32_t meta_bytes, const uint16_t* bias, int m, int n, int k, int groups, int tiles,
                             const uint16_t* norm_w, float norm_eps, const float* cosv, const float* sinv,
                             const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                             uint16_t* const* v_bufs, uint16_t* q_out, int h, int hkv, int d, int bshd, hipStream_t hs) {
    if (m < 1 || m > 32 || d % 32 != 0 || n != (h + 2 * hkv) * d || tiles * 16 != n) return ZL_ESHAPE;
    if (norm_w && m <= 8 && k > 4096) return ZL_ESHAPE;
    PhaseParams p = {};
    p.x = x; p.ldx = ldx; p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes;
    p.meta_bytes = meta_bytes; p.bias = bias; p.residual = nullptr; p.y = nullptr; p.m = m; p.n = n; p.k = k;
    p.groups = groups; p.tiles = tiles; p.phases = (groups + kW - 1) / kW; p.epi = bias ? ZL_EPI_BIAS : 0; p.ld_out = n;
    p.norm_w = norm_w; p.norm_eps = norm_eps;
    p.cosv = cosv; p.sinv = sinv; p.placement = placement; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs;
    p.q_out = q_out; p.h = h; p.hkv = hkv; p.d = d; p.bshd = bshd; p.pair_stride = d / 32;
    p.ks_ws = nullptr; p.ks_counter = nullptr;
    p.mg_ws = nullptr; p.mg_valid_lens = nullptr; p.mg_split_len = p.mg_max_splits = 0;
    const int grid = tiles / 2;
    if (norm_w && m > 8)                               // 9..32 rows: the deferred norm
        return m <= 16 ? launch_phase<2, 1, false, true, 1, false, false, true>(p, grid, hs)
                       : launch_phase<2, 2, false, true, 1, false, false, true>(p, grid, hs);
    if (norm_w || (m <= 4 && k <= 4096)) return launch_phase<2, 1, true, true>(p, grid, hs);
    return m <= 16 ? launch_phase<2, 1, false, true>(p, grid, hs) : launch_phase<2, 2, false, true>(p, grid, hs);
}

// internal (called by zl_w4a16_gemm_attn_merge): the attention output projection of a decode step reading the
// split-KV partials of zl_decode_attn_splits instead of a merged activation row.  m <= 4, k = heads * 128 <= 4096,
// max_splits <= 16, at most one generation of workgroups (tiles <= 2 * CUs).
int zl_w4a16_gemm_phase_merge(const float* ws, const int32_t* buf_lens, const int32_t* valid_lens, int split_len,
                              int max_splits, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes,
                              uint32_t meta_bytes, const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n,
                              int k, int groups, int tiles, int epilogue, hipStream_t hs) {
    if (m < 1 || m > 4 || k > 4096 || k % 128 != 0 || max_splits < 1 || max_splits > 16 || split_len < 1) return ZL_ESHAPE;
    if (epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) return ZL_ESHAPE;
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    const int r = (tiles + cus - 1) / cus;
    if (r > 2) return ZL_ESHAPE;
    PhaseParams p = {};
    p.x = nullptr; p.ldx = 0; p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes;
    p.meta_bytes = meta_bytes; p.bias = bias; p.residual = residual; p.y = y; p.m = m; p.n = n; p.k = k;
    p.groups = groups; p.tiles = tiles; p.phases = (groups + kW - 1) / kW; p.epi = epilogue; p.ld_out = n;
    p.norm_w = nullptr; p.norm_eps = 0.f;
    p.cosv = p.sinv = nullptr; p.placement = nullptr; p.buf_lens = buf_lens; p.k_bufs = p.v_bufs = nullptr; p.q_out = nullptr;
    p.h = p.hkv = p.d = p.bshd = 0; p.pair_stride = 1;
    p.ks_ws = nullptr; p.ks_counter = nullptr;
    p.mg_ws = ws; p.mg_valid_lens = valid_lens; p.mg_split_len = split_len; p.mg_max_splits = max_splits;
    const int grid = (tiles + r - 1) / r;
    return r == 1 ? launch_phase<1, 1, true, false, 1, true>(p, grid, hs) : launch_phase<2, 1, true, false, 1, true>(p, grid, hs);
}

#ifdef ZL_EXPERIMENTAL
// ---- digit planes: producer launch and the I8 instantiations ----------------------------------------------------------------
// planes buffer of an (m, k) activation matrix: [k / 128 groups][mb row blocks][6 KiB of A operands], then the constants
// [groups][mb][16 rows][2 floats]; mb = 1 up to 16 rows, 2 up to 32
int64_t zl_w4_planes_bytes_(int64_t m, int64_t k) {
    if (m < 1 || m > 32 || k < 128 || k % 128 != 0 || k > 16384) return ZL_ESHAPE;
    const int64_t mb = m <= 16 ? 1 : 2;
    return (k / 128) * mb * (6 * 1024 + 128);
}

int zl_w4_planes_launch(const uint16_t* x, int64_t ldx, int m, int k, const uint16_t* norm_w, float norm_eps, void* planes, hipStream_t hs) {
    if (zl_w4_planes_bytes_(m, k) < 0) return ZL_ESHAPE;
    const int mb = m <= 16 ? 1 : 2, groups = k / 128;
    unsigned char* pl = static_cast<unsigned char*>(planes);
    float* pc = reinterpret_cast<float*>(pl + (size_t)groups * mb * 6 * 1024);
    const int no = (k / 8 + 511) / 512;
    const dim3 grid(mb * 16), block(512);
    if (no <= 1) hipLaunchKernelGGL(k_w4_planes<1>, grid, block, 0, hs, x, ldx, m, k, norm_w, norm_eps, pl, pc, mb);
    else if (no <= 2) hipLaunchKernelGGL(k_w4_planes<2>, grid, block, 0, hs, x, ldx, m, k, norm_w, norm_eps, pl, pc, mb);
    else hipLaunchKernelGGL(k_w4_planes<4>, grid, block, 0, hs, x, ldx, m, k, norm_w, norm_eps, pl, pc, mb);
    return zl_launch_status();
}

static void planes_params(PhaseParams& p, const void* planes, int m, int k) {
    const int mb = m <= 16 ? 1 : 2, groups = k / 128;
    p.x = nullptr; p.ldx = 0;
    p.planes = static_cast<const unsigned char*>(planes);
    p.planes_bytes = (uint32_t)((size_t)groups * mb * 6 * 1024);
    p.pconsts = reinterpret_cast<const float*>(p.planes + p.planes_bytes);
    p.pconsts_bytes = (uint32_t)((size_t)groups * mb * 128);
}

// internal (called by zl_w4a16_gemm_planes): 5 <= m <= 32 rows as digit planes (zl_w4_planes_launch), k % 128 == 0, k <= 16384
int zl_w4a16_gemm_phase_planes(const void* planes, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                               const uint16_t* bias, const uint16_t* residual, uint16_t* y, int m, int n, int k, int groups, int tiles,
                               int epilogue, int ld_out, const zl_w4_opts_t* opts, hipStream_t hs) {
    static const zl_w4_opts_t kNoOpts = {};
    const zl_w4_opts_t& o = opts ? *opts : kNoOpts;
    if (zl_w4_planes_bytes_(m, k) < 0 || groups * 128 != k) return ZL_ESHAPE;
    PhaseParams p = {};
    planes_params(p, planes, m, k);
    p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes;
    p.meta_bytes = meta_bytes; p.bias = bias; p.residual = residual; p.y = y; p.m = m; p.n = n; p.k = k;
    p.groups = groups; p.tiles = tiles; p.phases = (groups + kW - 1) / kW; p.epi = epilogue; p.ld_out = ld_out; p.norm_w = nullptr; p.norm_eps = 0.f;
    p.cosv = p.sinv = nullptr; p.placement = p.buf_lens = nullptr; p.k_bufs = p.v_bufs = nullptr; p.q_out = nullptr;
    p.h = p.hkv = p.d = p.bshd = 0; p.pair_stride = 1;
    p.ks_ws = nullptr; p.ks_counter = nullptr;
    p.mg_ws = nullptr; p.mg_valid_lens = nullptr; p.mg_split_len = p.mg_max_splits = 0;
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    const int mb = m <= 16 ? 1 : 2;
    {   // few tiles and a long K (the down projection): K split over 2 / 4 workgroups of 2 / 4 tiles -- a workgroup pulls
        // 1 / KS of the planes through L2 and the grid keeps its size; partials and counters in the caller's scratch
        const int ksplit = o.phase_ksplit ? o.phase_ksplit : 4;
        const bool plain = !(epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32));
        const int64_t need = ZL_SCRATCH_HEADER + (int64_t)ksplit * m * n * (int64_t)sizeof(float);
        if ((ksplit == 2 || ksplit == 4) && k > 8192 && tiles <= 2 * cus && plain && o.scratch && o.scratch_bytes >= need) {
            p.ks_counter = reinterpret_cast<int*>(o.scratch);
            p.ks_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(o.scratch) + ZL_SCRATCH_HEADER);
            const int grid = (tiles + ksplit - 1) / ksplit * ksplit;
            if (mb == 1) return ksplit == 2 ? launch_phase<2, 1, false, false, 2, false, true>(p, grid, hs) : launch_phase<4, 1, false, false, 4, false, true>(p, grid, hs);
            return ksplit == 2 ? launch_phase<2, 2, false, false, 2, false, true>(p, grid, hs) : launch_phase<4, 2, false, false, 4, false, true>(p, grid, hs);
        }
    }
    int r = (tiles + cus - 1) / cus;
    if (r > 8) r = 8;
    if (o.phase_rounds > 0 && o.phase_rounds <= 8) r = o.phase_rounds;
    const int grid = (tiles + r - 1) / r;
#define ZL_PH8(RR)                                                                                             \
    case RR:                                                                                                   \
        return mb == 1 ? launch_phase<RR, 1, false, false, 1, false, true>(p, grid, hs) : launch_phase<RR, 2, false, false, 1, false, true>(p, grid, hs);
    switch (r) {
        ZL_PH8(1) ZL_PH8(2) ZL_PH8(3) ZL_PH8(4) ZL_PH8(5) ZL_PH8(6) ZL_PH8(7) ZL_PH8(8)
    }
#undef ZL_PH8
    return ZL_EINVAL;
}

// internal (called by zl_w4a16_qkv_rope_scatter_planes)
int zl_w4a16_gemm_phase_planes_rope(const void* planes, const uint32_t* qw, const uint32_t* meta, uint32_t qw_bytes, uint32_t meta_bytes,
                                    const uint16_t* bias, int m, int n, int k, int groups, int tiles, const float* cosv, const float* sinv,
                                    const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs, uint16_t* const* v_bufs,
                                    uint16_t* q_out, int h, int hkv, int d, int bshd, hipStream_t hs) {
    if (zl_w4_planes_bytes_(m, k) < 0 || groups * 128 != k || d % 32 != 0 || n != (h + 2 * hkv) * d || tiles * 16 != n) return ZL_ESHAPE;
    PhaseParams p = {};
    planes_params(p, planes, m, k);
    p.qw = reinterpret_cast<const uint4*>(qw); p.meta = meta; p.qw_bytes = qw_bytes;
    p.meta_bytes = meta_bytes; p.bias = bias; p.residual = nullptr; p.y = nullptr; p.m = m; p.n = n; p.k = k;
    p.groups = groups; p.tiles = tiles; p.phases = (groups + kW - 1) / kW; p.epi = bias ? ZL_EPI_BIAS : 0; p.ld_out = n;
    p.norm_w = nullptr; p.norm_eps = 0.f;
    p.cosv = cosv; p.sinv = sinv; p.placement = placement; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs;
    p.q_out = q_out; p.h = h; p.hkv = hkv; p.d = d; p.bshd = bshd; p.pair_stride = d / 32;
    p.ks_ws = nullptr; p.ks_counter = nullptr;
    p.mg_ws = nullptr; p.mg_valid_lens = nullptr; p.mg_split_len = p.mg_max_splits = 0;
    const int grid = tiles / 2;
    return m <= 16 ? launch_phase<2, 1, false, true, 1, false, true>(p, grid, hs) : launch_phase<2, 2, false, true, 1, false, true>(p, grid, hs);
}
#endif  // ZL_EXPERIMENTAL

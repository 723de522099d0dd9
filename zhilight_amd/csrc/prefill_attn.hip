// prefill_attn.hip -- causal ("encode part") attention of one task's prompt chunk on the matrix cores.
//
// Reference: attn_encode_group (src/nn/attention/attention.cpp:442-622) -> FlashDecoding::mha_fwd, i.e. the
// external flash-attn library (forbidden to borrow): out[q,h,:] = softmax_j<=pos(q)( scale * q.K_j ) . V with
// the K/V rows already in the task's buffer (the chunk's own rows are written before this runs, like
// copy_to_buffer in the reference), fp32 softmax, probabilities rounded to fp16 for the P.V product (what
// flash-attn does), fp32 accumulation.
//
// One workgroup = 64 query rows of one q head, 4 waves x 16 rows; key tiles of 64 rows go through LDS
// once per workgroup, both as stored (coalesced ds_write_b128); the V fragments of the P.V product (lane = column d,
// 8 keys) come out of the row-major tile through ds_read_b64_tr_b16 (a 16-lane group reads a 4-key x 16-d block, lane i
// gets column i: tools/ubench/tr_probe.hip) -- the first version transposed V with 32 two-byte LDS writes per thread
// and tile (53.7 -> 48.7 us at S = 1024); the next tile's loads ride one tile ahead in registers (another -10 %: a lone
// workgroup spends ~2.7 us per 64-key tile, mostly LDS + VALU: each of the 4 waves re-reads the whole K tile).  Query
// tiles rotated by the head index so that every XCD sees every tile length changed nothing.  Orientation: S^T = K.Q^T, so a lane's score registers belong to ONE
// query (lane & 15) -- row maxima need two cross-lane steps -- and, rounded to fp16, ARE the A operand
// of the P.V product (k index = key) with no transposition.  O accumulates in the C layout (rows 4 kq + i),
// its per-row rescale factors cross over through a 64-byte LDS strip per wave.
//
// Round 5: G wave GROUPS per workgroup (G x 4 waves).  All 512 workgroups of a 1 024-token prompt are resident at once, so the
// launch lasted as long as its longest workgroup: 16 key tiles in a row at ~2.7 us each (41 us, 0.08 of the MFMA peak) while the
// CUs holding the short ones idled.  Group g of a workgroup now walks the key tiles g, g + G, ... of the SAME 64 queries with its
// own K / V staging buffers and its own (O, m, l); the groups meet once at the end (LDS, flash-decoding's rescale).  The chain is
// cut G-fold; heavy query tiles are dispatched first.
#include <stdlib.h>
#include "zl_common.h"

namespace {

constexpr int kBQ = 64, kBK = 64, kD = 128;
constexpr int kKRow = kD + 8;       // K tile row (halfs), padded: conflict-free b128 reads
constexpr int kVRow = kD + 16;      // V tile row (halfs): 288 B, conflict-free transpose reads

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct PrefillParams {
    const uint16_t* q;      // (s_q, h, D)
    const uint16_t* k;      // task buffer, BSHD (len_buf, hkv, D) or BHSD (hkv, len_buf, D)
    const uint16_t* v;
    uint16_t* out;          // (s_q, h, D)
    int s_q, pos0, h, hkv, n_rep, len_buf, bshd;
    float scale;
    int nx, pair;           // query tiles; workgroup -> work item map (see the kernel)
};

template <int DT>
__device__ __forceinline__ f4 pf_mfma(uint4 a, uint4 b, f4 c) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    if constexpr (DT == ZL_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}

// DT = ZL_F16 / ZL_BF16: probabilities are rounded to T for the P.V product (flash-attention arithmetic)
constexpr int kGroupLds = (kBK * kKRow + kBK * kVRow) * 2;     // one group's K + V tile: 35 840 bytes (>= its 32.5 KB merge record)

template <int DT, int G>
__global__ __launch_bounds__(256 * G, G == 1 ? 2 : 4) void k_prefill_attn(const PrefillParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pf[];
    const int group = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));     // 4 waves each
    uint16_t* ks = reinterpret_cast<uint16_t*>(smem_pf + (size_t)group * kGroupLds);
    uint16_t* vt = ks + kBK * kKRow;
    float (*strip)[16] = reinterpret_cast<float (*)[16]>(smem_pf + (size_t)G * kGroupLds);     // [4 G waves][16]

    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, wave_all = threadIdx.x >> 6;
    const int nq = lane & 15, kq = lane >> 4;
    // work item k of the launch, longest first: query tile nx - 1 - k / h of head k % h.  Workgroup i takes item i (LPT order:
    // later workgroups fill the slots the short ones free) -- or, when every workgroup is resident at once (pair != 0: at most two
    // per CU), the first half takes the long items in order and the second half the short ones in REVERSE, so that the two
    // workgroups a CU receives (i and i + half in dispatch order) add up to the same number of key tiles: a 1 024-token prompt
    // gave CU c the query tile c % 16 twice -- 2 to 32 key tiles per CU, 43 us for the unlucky ones, 23 us of work on average
    const int total = p.nx * p.h;
    int item = (int)blockIdx.x;
    if (p.pair == 1 && item >= total / 2) item = total - 1 - (item - total / 2);
    int head = item % p.h, qt = p.nx - 1 - item / p.h;
    if (p.pair == 2) {                                    // the round-4 map (A/B): query tile fastest, in ascending order
        head = item / p.nx;
        qt = item % p.nx;
    }
    const int hk = head / p.n_rep;
    const int q0 = qt * kBQ;
    const int qrow = q0 + wave * 16 + nq;                 // the query this lane's scores belong to
    const int qpos = p.pos0 + qrow;                       // its position: keys 0 .. qpos are visible
    const size_t kv_stride = p.bshd ? (size_t)p.hkv * kD : (size_t)kD;
    const size_t kv_off = p.bshd ? (size_t)hk * kD : (size_t)hk * p.len_buf * kD;

    // Q fragments (B operand of S^T = K.Q^T): column q = lane & 15, k-chunk = d 32 t + 8 kq .. +7
    uint4 qf[4];
    {
        const int qr = qrow < p.s_q ? qrow : p.s_q - 1;
        const uint16_t* qp = p.q + ((size_t)qr * p.h + head) * kD + 8 * kq;
#pragma unroll
        for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const uint4*>(qp + 32 * t);
    }

    f4 o[8];                                              // O block: column d = 16 db + (lane & 15), rows q = 4 kq + i
#pragma unroll
    for (int db = 0; db < 8; ++db) o[db] = (f4){0.f, 0.f, 0.f, 0.f};
    float m_run = -1e20f, l_run = 0.f;                    // per query (lane & 15), replicated over kq

    const int last_q = min(q0 + kBQ, p.s_q) - 1;
    const int n_keys = p.pos0 + last_q + 1;               // keys any row of this block may see
    const int n_tiles = (n_keys + kBK - 1) / kBK;

    // K / V tile loads ride in registers one tile ahead (thread -> key = tid % 64, d chunks tid / 64 + 4 c): a workgroup
    // is one latency chain per tile otherwise (3 us per 64-key tile, ~2 of them waiting for the loads)
    uint4 kr0, kr1, kr2, kr3, vr0, vr1, vr2, vr3;         // (named: as arrays filled from two places they end up on the stack)
    const int skey = threadIdx.x & 63, sdch = (threadIdx.x >> 6) & 3;
#define ZL_PF_LOAD1(c_, kp_, vp_, dead_)                                                                   \
    kr##c_ = *reinterpret_cast<const uint4*>((kp_) + (sdch + 4 * c_) * 8);                               \
    vr##c_ = *reinterpret_cast<const uint4*>((vp_) + (sdch + 4 * c_) * 8);                               \
    if (dead_) vr##c_ = make_uint4(0, 0, 0, 0);           /* finite: its probability is exactly 0 */
#define ZL_PF_LOAD(tile_)                                                                                  \
    {                                                                                                      \
        const int kg_ = (tile_) * kBK + skey;                                                              \
        const int kc_ = kg_ < n_keys ? kg_ : n_keys - 1;                                                   \
        const uint16_t* kp_ = p.k + kv_off + (size_t)kc_ * kv_stride;                                     \
        const uint16_t* vp_ = p.v + kv_off + (size_t)kc_ * kv_stride;                                     \
        const bool dead_ = kg_ >= n_keys;                                                                  \
        ZL_PF_LOAD1(0, kp_, vp_, dead_) ZL_PF_LOAD1(1, kp_, vp_, dead_) ZL_PF_LOAD1(2, kp_, vp_, dead_) ZL_PF_LOAD1(3, kp_, vp_, dead_) \
    }
    ZL_PF_LOAD(group)

    // group g walks the tiles g, g + G, ...; the barriers are the workgroup's, so every group runs the same number of rounds
    // (a group past its last tile only keeps the barriers company)
    for (int tile = group; tile - group < n_tiles; tile += G) {
        const int key0 = tile * kBK;
        const bool live = tile < n_tiles;                 // group-uniform
        __syncthreads();                                  // previous tile fully consumed
#define ZL_PF_ST(c_)                                                                                       \
        *reinterpret_cast<uint4*>(&ks[skey * kKRow + (sdch + 4 * c_) * 8]) = kr##c_;                       \
        *reinterpret_cast<uint4*>(&vt[skey * kVRow + (sdch + 4 * c_) * 8]) = vr##c_;
        if (live) { ZL_PF_ST(0) ZL_PF_ST(1) ZL_PF_ST(2) ZL_PF_ST(3) }
#undef ZL_PF_ST
        __syncthreads();
        if (!live) continue;
        ZL_PF_LOAD(tile + G)                              // clamped past the end; in flight during this tile's MFMAs

        // ---- S^T = K . Q^T : 4 key blocks x 4 d steps; lane: query nq, keys key0 + 16 kb + 4 kq + i
        f4 st[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            st[kb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint4 a = *reinterpret_cast<const uint4*>(&ks[(kb * 16 + nq) * kKRow + 32 * t + 8 * kq]);
                st[kb] = pf_mfma<DT>(a, qf[t], st[kb]);
            }
        }
        // ---- scale, causal mask, online softmax for query nq (its 64 scores live in 4 lanes x 16 registers).  Scores are kept in
        //      the log2 domain (scale * log2 e folded into one multiply, v_exp_f32 is exp2); the causal select runs only on tiles
        //      that reach this wave's diagonal (wave-uniform: every other tile is fully visible)
        const float sl2 = p.scale * 1.44269504088896340736f;
        const bool diag = key0 + kBK - 1 > p.pos0 + q0 + wave * 16;      // some key of the tile lies beyond the wave's first query
        float mloc = -INFINITY;
        if (diag) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = key0 + kb * 16 + 4 * kq + i;
                    const float sv = key <= qpos ? st[kb][i] * sl2 : -INFINITY;
                    st[kb][i] = sv;
                    mloc = fmaxf(mloc, sv);
                }
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    st[kb][i] *= sl2;
                    mloc = fmaxf(mloc, st[kb][i]);
                }
            }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float lsum = 0.f;
        uint4 pf[2];                                      // P as the A operand of P.V: k-chunk kq of key group j
        auto cvt = [](float x) -> uint16_t {
            if constexpr (DT == ZL_F16) return __builtin_bit_cast(uint16_t, (_Float16)x);
            else {
                const uint32_t u = __builtin_bit_cast(uint32_t, x);
                return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            }
        };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t w[4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    const uint16_t p0 = cvt(__builtin_amdgcn_exp2f(st[2 * j + half][2 * i2] - m_new));
                    const uint16_t p1 = cvt(__builtin_amdgcn_exp2f(st[2 * j + half][2 * i2 + 1] - m_new));
                    lsum += ZT<DT>::to_f32(p0) + ZT<DT>::to_f32(p1);   // the normaliser sums what the product uses
                    w[half * 2 + i2] = (uint32_t)p0 | ((uint32_t)p1 << 16);
                }
            }
            pf[j] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        l_run = l_run * alpha + lsum;                     // per-lane partial; lanes of a query are merged at the end
        // ---- rescale O: the factor of row q = 4 kq + i comes from the lane whose nq is that row -- only when some row's maximum
        //      moved (wave-uniform; after the first tiles it rarely does: 32 multiplies and an LDS round trip per tile otherwise)
        if (!__all(alpha == 1.0f)) {
            if (kq == 0) strip[wave_all][nq] = alpha;
            __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the strip write has landed (wave-private)
            const f4 al = *reinterpret_cast<const f4*>(&strip[wave_all][4 * kq]);
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                o[db][0] *= al[0]; o[db][1] *= al[1]; o[db][2] *= al[2]; o[db][3] *= al[3];
            }
        }
        // ---- O += P . V : 2 key groups x 8 d blocks; B fragment (column d = 16 db + nq, keys 32 j + {4 kq + i, 16 + 4 kq + i})
        //      = two transposed 8-byte reads of the row-major V tile
        const uint16_t* vtr = vt + (4 * kq + (nq >> 2)) * kVRow + 4 * (nq & 3);
        typedef short s4 __attribute__((ext_vector_type(4)));
        typedef short s8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(vtr + (32 * j) * kVRow + 16 * db));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(vtr + (32 * j + 16) * kVRow + 16 * db));
                const s8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = pf_mfma<DT>(pf[j], __builtin_bit_cast(uint4, b), o[db]);
            }
        }
    }

#undef ZL_PF_LOAD
#undef ZL_PF_LOAD1
    // ---- the groups meet: groups 1 .. G - 1 park (O, m, l) in their own tile area, group 0 folds them in with the usual
    //      rescale (scores live in the log2 domain: exp2).  l is still the lane's partial sum; the lanes of a query share m.
    if constexpr (G > 1) {
        __syncthreads();                                  // every group is done with its tiles
        float* rec = reinterpret_cast<float*>(smem_pf + (size_t)group * kGroupLds);     // [wave][16 rows][128] O, then [wave][64 lanes] m, l
        if (group > 0) {
#pragma unroll
            for (int db = 0; db < 8; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) rec[(wave * 16 + 4 * kq + i) * kD + 16 * db + nq] = o[db][i];
            rec[4 * 16 * kD + wave * 64 + lane] = m_run;
            rec[4 * 16 * kD + 4 * 64 + wave * 64 + lane] = l_run;
        }
        __syncthreads();
        if (group > 0) return;
#pragma unroll 1
        for (int g = 1; g < G; ++g) {
            const float* rg = reinterpret_cast<const float*>(smem_pf + (size_t)g * kGroupLds);
            const float mg = rg[4 * 16 * kD + wave * 64 + lane], lg = rg[4 * 16 * kD + 4 * 64 + wave * 64 + lane];
            const float m_new = fmaxf(m_run, mg);
            const float a0 = __builtin_amdgcn_exp2f(m_run - m_new), ag = __builtin_amdgcn_exp2f(mg - m_new);
            m_run = m_new;
            l_run = l_run * a0 + lg * ag;
            if (kq == 0) {
                strip[wave_all][nq] = a0;
                strip[4 + wave_all][nq] = ag;            // group 0's waves are 0 .. 3: the strips of waves 4 .. 7 are free now
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const f4 f0 = *reinterpret_cast<const f4*>(&strip[wave_all][4 * kq]);
            const f4 fg = *reinterpret_cast<const f4*>(&strip[4 + wave_all][4 * kq]);
#pragma unroll
            for (int db = 0; db < 8; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    o[db][i] = o[db][i] * f0[i] + rg[(wave * 16 + 4 * kq + i) * kD + 16 * db + nq] * fg[i];
            __builtin_amdgcn_s_waitcnt(0xc07f);           // the strip is read before the next round overwrites it (wave-private)
        }
    }
    // ---- normalise: l of query nq = sum over its 4 lanes; bring 1/l to the C layout through the strip
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (kq == 0) strip[wave_all][nq] = 1.0f / (l_run + 1e-20f);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const f4 inv = *reinterpret_cast<const f4*>(&strip[wave_all][4 * kq]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = q0 + wave * 16 + 4 * kq + i;
        if (row >= p.s_q) continue;
        uint16_t* op = p.out + ((size_t)row * p.h + head) * kD + nq;
#pragma unroll
        for (int db = 0; db < 8; ++db) op[db * 16] = ZT<DT>::from_f32(o[db][i] * inv[i]);
    }
}

}  // namespace

extern "C" int zl_prefill_attn_ex(const uint16_t* q, const uint16_t* k_buf, const uint16_t* v_buf, uint16_t* out, int64_t s_q,
                                  int64_t pos0, int64_t h, int64_t hkv, int64_t d, float scale, int64_t len_buf, int bshd,
                                  int dtype, int groups, zl_stream_t s);

extern "C" int zl_prefill_attn(const uint16_t* q, const uint16_t* k_buf, const uint16_t* v_buf, uint16_t* out, int64_t s_q,
                               int64_t pos0, int64_t h, int64_t hkv, int64_t d, float scale, int64_t len_buf, int bshd,
                               int dtype, zl_stream_t s) {
    return zl_prefill_attn_ex(q, k_buf, v_buf, out, s_q, pos0, h, hkv, d, scale, len_buf, bshd, dtype, 0, s);
}

template <int DT, int G>
static int launch_prefill(const PrefillParams& p, dim3 grid, hipStream_t hs) {
    const size_t lds = (size_t)G * kGroupLds + (size_t)4 * G * 16 * sizeof(float);       // tiles + one 64-byte strip per wave
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prefill_attn<DT, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ZL_ELIMIT;
    }
    hipLaunchKernelGGL((k_prefill_attn<DT, G>), grid, dim3(256 * G), lds, hs, p);
    return zl_launch_status();
}

// groups: wave groups per workgroup (1 / 2 / 4: the key tiles of a query tile are dealt to them round-robin); 0 = the launcher's
// choice: as many as the longest query tile has key tiles to hand out, up to 4
extern "C" int zl_prefill_attn_ex(const uint16_t* q, const uint16_t* k_buf, const uint16_t* v_buf, uint16_t* out, int64_t s_q,
                                  int64_t pos0, int64_t h, int64_t hkv, int64_t d, float scale, int64_t len_buf, int bshd,
                                  int dtype, int groups, zl_stream_t s) {
    ZL_CHECK_ARG(q && k_buf && v_buf && out && s_q > 0 && pos0 >= 0 && h > 0 && hkv > 0 && len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(h % hkv == 0 && pos0 + s_q <= len_buf, ZL_ESHAPE);
    ZL_CHECK_ARG(d == kD, ZL_ESHAPE);          // other head sizes: the mask form of zl_decode_attn
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    ZL_CHECK_ARG(h <= 65535, ZL_ELIMIT);
    const bool plain_map = groups == -1;               // A/B switch: one group, work items in plain dispatch order (the round-4 launch)
    if (plain_map) groups = 1;
    ZL_CHECK_ARG(groups == 0 || groups == 1 || groups == 2 || groups == 4, ZL_EINVAL);
    PrefillParams p;
    p.q = q; p.k = k_buf; p.v = v_buf; p.out = out;
    p.s_q = (int)s_q; p.pos0 = (int)pos0; p.h = (int)h; p.hkv = (int)hkv; p.n_rep = (int)(h / hkv);
    p.len_buf = (int)len_buf; p.bshd = bshd; p.scale = scale;
    p.nx = (int)((s_q + kBQ - 1) / kBQ);
    ZL_CHECK_ARG((int64_t)p.nx * h < ((int64_t)1 << 31), ZL_ELIMIT);
    const dim3 grid((unsigned)(p.nx * (int)h));
    // measured (profiles/r05_prefill_attn.txt): more wave groups make the launch SLOWER (43 -> 72 us per layer at 1 024 tokens with
    // four): a CU is throughput-bound from two 4-wave workgroups on, what a long query tile lacked was a short neighbour, not waves
    if (groups == 0) groups = 1;
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    const int total = p.nx * (int)h;
    p.pair = (groups == 1 && total % 2 == 0 && total <= 2 * cus) ? 1 : 0;
    if (plain_map) p.pair = 2;
    hipStream_t hs = (hipStream_t)s;
#define ZL_PF_G(GG) return dtype == ZL_F16 ? launch_prefill<ZL_F16, GG>(p, grid, hs) : launch_prefill<ZL_BF16, GG>(p, grid, hs);
    switch (groups) {
        case 1: ZL_PF_G(1)
        case 2: ZL_PF_G(2)
        default: ZL_PF_G(4)
    }
#undef ZL_PF_G
}

// attention.hip -- decode ("search") attention over per-task ragged KV buffers.  SURVEY 8a row a15.
//
// Reference semantics (src/nn/attention/attention_kernel.cu:673-725 default kernel, :729-923 split-KV):
//   out[b,q,h,:] = sum_j p_j V_j,  p = softmax_j( mask ? scale * q.K_j : -inf ), fp32 throughout,
//   max initialised to -1e20, normaliser to 1e-20; K/V addressed through per-task device pointers,
//   BSHD (len_buf, Hkv, D) or BHSD (Hkv, len_buf, D).
//
// MI355X design (KV streaming is HBM-bound at large batch, latency-bound at batch 1):
//  * GQA-aware: a workgroup serves ONE kv head and all of its n_rep*len_q query rows (RT per pass),
//    so K and V are read once per kv head instead of once per q head as the reference's default
//    kernel does.
//  * flash-decoding split over the KV length: grid (splits, Hkv, B*passes); a split is a multiple of
//    128 keys; 4 wavefronts per workgroup take 32-key chunks round-robin.  Nothing is sized by
//    max_len_buf in LDS (the reference materialises all logits in shared memory).
//  * inside a wavefront a key row (D halfs) is spread over D/8 lanes x 16 B, so one
//    global_load_dwordx4 per lane fetches 64/(D/8) whole keys; 8 K loads + 8 V loads are issued
//    per chunk before any math.  q.k partials are v_dot2_f32_f16, reduced over the D/8 lanes with
//    xor shuffles; softmax is online per 32-key chunk (one rescale per chunk); P.V accumulates in
//    fp32 registers per lane-group and is merged group->wave->workgroup at the end.
//  * partial (acc[D], m, l) per split goes to a small fp32 workspace; a second tiny kernel merges
//    the splits:  out = sum_s acc_s e^{m_s-M} / (sum_s l_s e^{m_s-M} + 1e-20).
#include <algorithm>
#include "zl_common.h"

namespace {

constexpr int kSteps = 8;  // key-steps per chunk
constexpr int kMaxSplits = 512;   // splits a merge can hold (k_decode_attn_combine's kMaxS)

struct AttnParams {
    const uint16_t* q;
    const int32_t* buf_lens;
    const uint16_t* const* k_bufs;
    const uint16_t* const* v_bufs;
    const int8_t* mask;
    const int32_t* valid_lens;
    uint16_t* out;
    float* ws;
    int b, len_q, h, hkv, n_rep, rows;  // rows = len_q * n_rep
    int passes, split_len, max_splits;
    float scale;
    int bshd;
    // fused decode front end (FUSE): q/k/v come from the fused qkv rows and are rotated in-kernel
    const uint16_t* qkv;        // (B, (H + 2 Hkv) * D)
    const float* cosv;          // (B, D)
    const float* sinv;
    const int32_t* placement;   // (B) slot of the new token in its task's buffers
    uint16_t* const* k_bufs_w;  // writable views of k_bufs / v_bufs
    uint16_t* const* v_bufs_w;
    int neox;
    // INT8 cache (k_decode_attn_partial_q8): k_bufs / v_bufs hold u8 codes, one fp32 scale per (key, kv head)
    const float* const* k_scales;
    const float* const* v_scales;
    // k_decode_attn_mfma: 1 = partials leave as NORMALISED fp16 rows [vh][split][128] + fp32 (max, sum) pairs behind them
    // (zl_decode_attn_splits_h: 264 instead of 520 bytes per (head, split) for the merging projection to re-read)
    int half_partials;
    // k_decode_attn_mfma, last-arriver merge (zl_decode_attn_la): la = 1 -> every workgroup publishes its split record write-through,
    // counts its arrival on la_cnt[task * hkv + kv head] and the LAST one of the pair merges the pair's records into `out`
    // (no merge launch, no merging prologue); la_cnt is all zero between launches (the last arriver resets its word)
    int la;
    int* la_cnt;
};

typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

// split length: multiple of 128 keys, grown so that about >= 1024 workgroups exist when possible
static inline int attn_split_len(int64_t b, int64_t hkv, int64_t max_len) {
    // about 1024 workgroups (4 resident per CU = one generation).  Short splits pay their fixed cost twice when the
    // grid spills into a second generation, so they are rounded up instead (batch 32, 1088-key buffers: 3 splits of
    // 384 keys instead of 5 of 256: 34.7 vs 37.6 us per layer); long splits keep the round-down (8192 keys, batch 8:
    // 512-key splits 61.4 us, 640-key splits 65.5)
    int64_t want = (max_len * b * hkv) / 1024;
    int64_t ls = (want / 128) * 128;
    if (ls < 128) ls = 128;
    if (ls <= 256 && ((max_len + ls - 1) / ls) * b * hkv > 1024) ls += 128;
    if (ls > 2048) ls = 2048;
    // the merge (k_decode_attn_combine, the attn_out prologue) holds at most kMaxSplits splits: few (task, kv head) pairs
    // with a very long buffer (batch 1, one local kv head under ATTN_KV_REP_TP, 128 k keys) would otherwise exceed it and
    // silently drop keys.  Beyond kMaxSplits * 2048 keys the launchers return ZL_ELIMIT.
    while ((max_len + ls - 1) / ls > kMaxSplits && ls < (int64_t)1 << 30) ls += 128;
    return (int)ls;
}

template <int DT>
__device__ __forceinline__ float dot8(const uint4& a, const uint4& b) {
    const uint32_t au[4] = {a.x, a.y, a.z, a.w}, bu[4] = {b.x, b.y, b.z, b.w};
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (DT == ZL_F16) {
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(hv2, au[e]), __builtin_bit_cast(hv2, bu[e]), acc, false);
        } else {
            acc = __builtin_fmaf(__builtin_bit_cast(float, au[e] << 16), __builtin_bit_cast(float, bu[e] << 16), acc);
            acc = __builtin_fmaf(__builtin_bit_cast(float, au[e] & 0xffff0000u),
                                 __builtin_bit_cast(float, bu[e] & 0xffff0000u), acc);
        }
    }
    return acc;
}

template <int DT>
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = ZT<DT>::to_f32((uint16_t)(u[e] & 0xffff));
        f[2 * e + 1] = ZT<DT>::to_f32((uint16_t)(u[e] >> 16));
    }
}

// rotate 8 consecutive head-dim elements (a) with their rotation partners (b): one rounding to T,
// same fp32 expression as rope_one_value (src/nn/position/rope_common.cuh:14-34)
template <int DT>
__device__ __forceinline__ uint4 rope8(const uint4& a, const uint4& b, const float* c, const float* s, int d0, int half,
                                       int neox) {
    float af[8], bf[8];
    unpack8<DT>(a, af);
    unpack8<DT>(b, bf);
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        float r[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = e + t;
            bool minus;
            float partner;
            if (neox) {
                minus = (d0 + i) < half;
                partner = bf[i];
            } else {
                minus = (i & 1) == 0;
                partner = af[i ^ 1];
            }
            r[t] = minus ? __builtin_fmaf(-partner, s[i], af[i] * c[i]) : __builtin_fmaf(partner, s[i], af[i] * c[i]);
        }
        o[e / 2] = (uint32_t)ZT<DT>::from_f32(r[0]) | ((uint32_t)ZT<DT>::from_f32(r[1]) << 16);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// sum over the LPK lanes that share a key row, result in every one of them.  DPP only (VALU, no LDS
// crossbar): xor 1 / xor 2 by quad_perm, then row_half_mirror (quad pairs) and row_mirror (halves of the
// 16-lane row); a key row wider than one DPP row (D = 256) needs one real cross-row shuffle.
template <int LPK>
__device__ __forceinline__ float key_row_sum(float v) {
#ifdef ZL_ATTN_NO_DPP
#pragma unroll
    for (int off = LPK / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
#endif
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    if (LPK >= 8)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
    if (LPK >= 16)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
    if (LPK >= 32) v += __shfl_xor(v, 16, 64);
    return v;
}

// MASKED = false: prefix visibility (valid_lens), the decode fast path -- no per-key visibility work at all
// because the split range is already clipped to the visible prefix.
template <int DT, int D, int RT, bool FUSE, bool MASKED>
__global__ __launch_bounds__(256) void k_decode_attn_partial(const AttnParams p) {
    constexpr int LPK = D / 8;       // lanes per key row
    constexpr int KPS = 64 / LPK;    // keys per wave-step
    constexpr int CHUNK = KPS * kSteps;
    __shared__ float xw[4][RT][D + 2];

    const int b = blockIdx.z / p.passes, pass = blockIdx.z % p.passes;
    const int hk = blockIdx.y, split = blockIdx.x;
    // all scalar inputs are fetched before the first use so the dependent-load chain is one level deep
    const int len = p.buf_lens[b];
    const int vlen_in = MASKED ? 0x7fffffff : p.valid_lens[b];
    const uint16_t* kbase = p.k_bufs[b];
    const uint16_t* vbase = p.v_bufs[b];
    const int place = FUSE ? p.placement[b] : -1;
    const int elen = MASKED ? len : min(len, vlen_in);
    const int t0 = split * p.split_len;
    // FUSE: the workgroup whose split covers the new token's slot must store it even if that slot is
    // not (yet) visible, so it may not leave before the store below
    const bool owns_new = FUSE && place >= t0 && place < t0 + p.split_len && place < len;
    if (t0 >= elen && !owns_new) return;
    const int t1 = min(elen, t0 + p.split_len);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPK, dp = lane % LPK;
    const size_t kv_stride = p.bshd ? (size_t)p.hkv * D : (size_t)D;
    const size_t kv_off = (p.bshd ? (size_t)hk * D : (size_t)hk * len * D) + dp * 8;

    // ---- first KV chunk of this wave: issued before anything else (batch-1 decode is a latency chain).
    // Addresses are clamped into the buffer, so the loads need no branch even for a wave without work.
    uint4 kk[kSteps], vv[kSteps];
    int c0 = t0 + wave * CHUNK;
    const int last_key = t1 > 0 ? t1 - 1 : 0;
#define ZL_LOAD_CHUNK(base)                                                                      \
    _Pragma("unroll") for (int st = 0; st < kSteps; ++st) {                                      \
        const int key_ = (base) + st * KPS + grp;                                                \
        const int kc_ = key_ < last_key ? key_ : last_key;                                       \
        kk[st] = *reinterpret_cast<const uint4*>(kbase + kv_off + (size_t)kc_ * kv_stride);     \
        vv[st] = *reinterpret_cast<const uint4*>(vbase + kv_off + (size_t)kc_ * kv_stride);     \
    }
    if (len <= 0) return;
    ZL_LOAD_CHUNK(c0)

    size_t mask_off = 0;
    if (MASKED) {
        for (int i = 0; i < b; ++i) mask_off += (size_t)p.buf_lens[i];
        mask_off *= p.len_q;
    }

    uint4 qv[RT];
    int q_of_row[RT];
    uint4 k_new = make_uint4(0, 0, 0, 0), v_new = make_uint4(0, 0, 0, 0);
    if constexpr (FUSE) {
        // q rows, the new k row and the new v row of this kv head straight from the fused qkv vector
        constexpr int half = D / 2;
        const int d0 = dp * 8;
        const int pd0 = p.neox ? (d0 < half ? d0 + half : d0 - half) : d0;
        float c[8], sn[8];
        const float4* cp = reinterpret_cast<const float4*>(p.cosv + (size_t)b * D + d0);
        const float4* sp = reinterpret_cast<const float4*>(p.sinv + (size_t)b * D + d0);
        const float4 c0v = cp[0], c1v = cp[1], s0v = sp[0], s1v = sp[1];
        c[0] = c0v.x; c[1] = c0v.y; c[2] = c0v.z; c[3] = c0v.w; c[4] = c1v.x; c[5] = c1v.y; c[6] = c1v.z; c[7] = c1v.w;
        sn[0] = s0v.x; sn[1] = s0v.y; sn[2] = s0v.z; sn[3] = s0v.w; sn[4] = s1v.x; sn[5] = s1v.y; sn[6] = s1v.z; sn[7] = s1v.w;
        const uint16_t* row = p.qkv + (size_t)b * (p.h + 2 * p.hkv) * D;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            qv[i] = make_uint4(0, 0, 0, 0);
            q_of_row[i] = 0;
            const int rr = pass * RT + i;
            if (rr < p.rows) {
                const uint16_t* src = row + (size_t)(hk * p.n_rep + rr) * D;
                qv[i] = rope8<DT>(*reinterpret_cast<const uint4*>(src + d0), *reinterpret_cast<const uint4*>(src + pd0), c, sn,
                                  d0, half, p.neox);
            }
        }
        const uint16_t* ksrc = row + (size_t)(p.h + hk) * D;
        k_new = rope8<DT>(*reinterpret_cast<const uint4*>(ksrc + d0), *reinterpret_cast<const uint4*>(ksrc + pd0), c, sn, d0,
                          half, p.neox);
        v_new = *reinterpret_cast<const uint4*>(row + (size_t)(p.h + p.hkv + hk) * D + d0);
        // the workgroup whose split owns the new token's slot stores the row (copy_to_rag_buffer2 semantics)
        if (owns_new && pass == 0 && wave == 0 && grp == 0) {
            *reinterpret_cast<uint4*>(p.k_bufs_w[b] + kv_off + (size_t)place * kv_stride) = k_new;
            *reinterpret_cast<uint4*>(p.v_bufs_w[b] + kv_off + (size_t)place * kv_stride) = v_new;
        }
        if (t0 >= elen) return;  // wave-uniform: nothing visible in this split
    } else {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int rr = pass * RT + i;
            qv[i] = make_uint4(0, 0, 0, 0);
            q_of_row[i] = 0;
            if (rr < p.rows) {
                const int qi = rr / p.n_rep, head = hk * p.n_rep + rr % p.n_rep;
                q_of_row[i] = qi;
                qv[i] = *reinterpret_cast<const uint4*>(p.q + (((size_t)b * p.len_q + qi) * p.h + head) * D + dp * 8);
            }
        }
    }

    float m[RT], l[RT], acc[RT][8];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        m[i] = -1e20f;
        l[i] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    }

    while (c0 < t1) {
        float sc[RT][kSteps];
#pragma unroll
        for (int st = 0; st < kSteps; ++st) {
            const int key = c0 + st * KPS + grp;
            if (FUSE && key == place) {  // the row written by this launch: take it from registers
                kk[st] = k_new;
                vv[st] = v_new;
            }
            const bool inb = key < t1;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const float d = key_row_sum<LPK>(dot8<DT>(qv[i], kk[st]));
                bool vis = inb;
                if (MASKED) {
                    if (vis) vis = p.mask[mask_off + (size_t)q_of_row[i] * len + key] != 0;
                }
                sc[i][st] = vis ? d * p.scale : -INFINITY;
            }
            if (!inb) vv[st] = make_uint4(0, 0, 0, 0);  // clamped duplicate: contributes p = 0, keep it finite
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float mc = sc[i][0];
#pragma unroll
            for (int st = 1; st < kSteps; ++st) mc = fmaxf(mc, sc[i][st]);
            const float mn = fmaxf(m[i], mc);
            const float alpha = __expf(m[i] - mn);
            m[i] = mn;
            l[i] *= alpha;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[i][e] *= alpha;
        }
#pragma unroll
        for (int st = 0; st < kSteps; ++st) {
            float vf[8];
            unpack8<DT>(vv[st], vf);
            if (MASKED) {
                // a key no query row sees must not leak a (possibly non-finite) V row: p = 0 times inf is NaN
                bool any = false;
#pragma unroll
                for (int i = 0; i < RT; ++i) any = any || (sc[i][st] != -INFINITY);
                if (!any) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vf[e] = 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const float pr = __expf(sc[i][st] - m[i]);
                l[i] += pr;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][e] = __builtin_fmaf(pr, vf[e], acc[i][e]);
            }
        }
        c0 += 4 * CHUNK;
        if (c0 >= t1) break;
        ZL_LOAD_CHUNK(c0)
    }
#undef ZL_LOAD_CHUNK

    // ---- merge the KPS lane-groups of the wave
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float m2 = __shfl_xor(m[i], off, 64), l2 = __shfl_xor(l[i], off, 64);
            const float mn = fmaxf(m[i], m2);
            const float f1 = __expf(m[i] - mn), f2 = __expf(m2 - mn);
            l[i] = l[i] * f1 + l2 * f2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a2 = __shfl_xor(acc[i][e], off, 64);
                acc[i][e] = acc[i][e] * f1 + a2 * f2;
            }
            m[i] = mn;
        }
    }
    // ---- merge the 4 waves through LDS
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xw[wave][i][dp * 8 + e] = acc[i][e];
            if (dp == 0) {
                xw[wave][i][D] = m[i];
                xw[wave][i][D + 1] = l[i];
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < RT * D; idx += 256) {
        const int i = idx / D, d = idx % D;
        const int rr = pass * RT + i;
        if (rr >= p.rows) continue;
        float mn = xw[0][i][D];
#pragma unroll
        for (int w = 1; w < 4; ++w) mn = fmaxf(mn, xw[w][i][D]);
        float a = 0.f, lt = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __expf(xw[w][i][D] - mn);
            a = __builtin_fmaf(xw[w][i][d], f, a);
            lt = __builtin_fmaf(xw[w][i][D + 1], f, lt);
        }
        const int qi = rr / p.n_rep, head = hk * p.n_rep + rr % p.n_rep;
        float* dst = p.ws + ((((size_t)b * p.len_q + qi) * p.h + head) * p.max_splits + split) * (D + 2);
        dst[d] = a;
        if (d == 0) {
            dst[D] = mn;
            dst[D + 1] = lt;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// INT8 KV cache.  Reference: KERNEL_mqa_rag_buffer_split_kv_quant (attention_kernel.cu:802-878) with
// quant_attention.cuh:39-123:  logit_j = scale * sk_j * q.(K_j - 128),  out = sum_j p_j sv_j (V_j - 128).
// Same split/merge structure as the fp16 kernel; a key row is D bytes, so a lane's 16 B hold 16 codes,
// a row takes D/16 lanes and a wave-step covers twice the keys (HBM bytes per key: 2 D + 8 instead of 4 D).
// codes -> exact fp16 (code - 128) without a conversion instruction: byte | 0x6400 is the half 1024 + code,
// one packed subtract of 1152 later it is (code - 128); q.k is then v_dot2_f32_f16 with fp32 accumulation
// (the reference multiplies and adds in fp16 there).  P.V converts the codes to fp32.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void codes_to_h2(const uint4& w, hv2 (&h)[8]) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
    const hv2 bias = {(_Float16)1152.f, (_Float16)1152.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[2 * e] = __builtin_bit_cast(hv2, __builtin_amdgcn_perm(0x64646464u, u[e], 0x04010400u)) - bias;
        h[2 * e + 1] = __builtin_bit_cast(hv2, __builtin_amdgcn_perm(0x64646464u, u[e], 0x04030402u)) - bias;
    }
}
__device__ __forceinline__ void codes_to_f32(const uint4& w, float (&f)[16]) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[4 * e + j] = (float)((u[e] >> (8 * j)) & 0xffu) - 128.f;
    }
}

template <int DT, int D, int RT, bool MASKED>
__global__ __launch_bounds__(256) void k_decode_attn_partial_q8(const AttnParams p) {
    constexpr int LPK = D / 16;      // lanes per key row (16 codes each)
    constexpr int KPS = 64 / LPK;    // keys per wave-step
    constexpr int CHUNK = KPS * kSteps;
    constexpr int QW = DT == ZL_F16 ? 8 : 16;
    __shared__ float xw[4][RT][D + 2];

    const int b = blockIdx.z / p.passes, pass = blockIdx.z % p.passes;
    const int hk = blockIdx.y, split = blockIdx.x;
    const int len = p.buf_lens[b];
    const int vlen_in = MASKED ? 0x7fffffff : p.valid_lens[b];
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k_bufs[b]);
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v_bufs[b]);
    const float* ksc = p.k_scales[b];
    const float* vsc = p.v_scales[b];
    const int elen = MASKED ? len : min(len, vlen_in);
    const int t0 = split * p.split_len;
    if (t0 >= elen || len <= 0) return;
    const int t1 = min(elen, t0 + p.split_len);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPK, dp = lane % LPK;
    const size_t kv_stride = p.bshd ? (size_t)p.hkv * D : (size_t)D;
    const size_t kv_off = (p.bshd ? (size_t)hk * D : (size_t)hk * len * D) + dp * 16;
    const size_t sc_stride = p.bshd ? (size_t)p.hkv : 1;
    const size_t sc_off = p.bshd ? (size_t)hk : (size_t)hk * len;

    uint4 kk[kSteps], vv[kSteps];
    float ks[kSteps], vs[kSteps];
    int c0 = t0 + wave * CHUNK;
    const int last_key = t1 - 1;
#define ZL_LOAD_CHUNK_Q8(base)                                                                   \
    _Pragma("unroll") for (int st = 0; st < kSteps; ++st) {                                      \
        const int key_ = (base) + st * KPS + grp;                                                \
        const int kc_ = key_ < last_key ? key_ : last_key;                                       \
        kk[st] = *reinterpret_cast<const uint4*>(kbase + kv_off + (size_t)kc_ * kv_stride);     \
        vv[st] = *reinterpret_cast<const uint4*>(vbase + kv_off + (size_t)kc_ * kv_stride);     \
        ks[st] = ksc[sc_off + (size_t)kc_ * sc_stride];                                          \
        vs[st] = vsc[sc_off + (size_t)kc_ * sc_stride];                                          \
    }
    ZL_LOAD_CHUNK_Q8(c0)

    size_t mask_off = 0;
    if (MASKED) {
        for (int i = 0; i < b; ++i) mask_off += (size_t)p.buf_lens[i];
        mask_off *= p.len_q;
    }

    // q: fp16 stays packed (dot2), bf16 is widened to fp32
    hv2 qh[RT][8];
    float qf[RT][QW];
    int q_of_row[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int rr = pass * RT + i;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        q_of_row[i] = 0;
        if (rr < p.rows) {
            const int qi = rr / p.n_rep, head = hk * p.n_rep + rr % p.n_rep;
            q_of_row[i] = qi;
            const uint16_t* qp = p.q + (((size_t)b * p.len_q + qi) * p.h + head) * D + dp * 16;
            lo = *reinterpret_cast<const uint4*>(qp);
            hi = *reinterpret_cast<const uint4*>(qp + 8);
        }
        if constexpr (DT == ZL_F16) {
            const uint32_t u[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) qh[i][e] = __builtin_bit_cast(hv2, u[e]);
        } else {
            float a[8], c[8];
            unpack8<DT>(lo, a);
            unpack8<DT>(hi, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qf[i][e] = a[e];
                qf[i][8 + e] = c[e];
            }
        }
    }

    float m[RT], l[RT], acc[RT][16];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        m[i] = -1e20f;
        l[i] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    }

    while (c0 < t1) {
        float sc[RT][kSteps];
#pragma unroll
        for (int st = 0; st < kSteps; ++st) {
            const int key = c0 + st * KPS + grp;
            const bool inb = key < t1;
            const float lscale = ks[st] * p.scale;
            hv2 kh[8];
            float kf[16];
            if constexpr (DT == ZL_F16) codes_to_h2(kk[st], kh);
            else codes_to_f32(kk[st], kf);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                float d = 0.f;
                if constexpr (DT == ZL_F16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) d = __builtin_amdgcn_fdot2(qh[i][e], kh[e], d, false);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) d = __builtin_fmaf(qf[i][e], kf[e], d);
                }
                d = key_row_sum<LPK>(d);
                bool vis = inb;
                if (MASKED) {
                    if (vis) vis = p.mask[mask_off + (size_t)q_of_row[i] * len + key] != 0;
                }
                sc[i][st] = vis ? d * lscale : -INFINITY;
            }
            if (!inb) vs[st] = 0.f;   // clamped duplicate: weight 0
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float mc = sc[i][0];
#pragma unroll
            for (int st = 1; st < kSteps; ++st) mc = fmaxf(mc, sc[i][st]);
            const float mn = fmaxf(m[i], mc);
            const float alpha = __expf(m[i] - mn);
            m[i] = mn;
            l[i] *= alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] *= alpha;
        }
#pragma unroll
        for (int st = 0; st < kSteps; ++st) {
            float vf[16];
            codes_to_f32(vv[st], vf);
            float sv = vs[st];
            if (MASKED) {
                // a key no row sees: its scale may be uninitialised memory (inf/nan) -- weight it by exactly 0
                bool any = false;
#pragma unroll
                for (int i = 0; i < RT; ++i) any = any || (sc[i][st] != -INFINITY);
                if (!any) sv = 0.f;
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const float pr = __expf(sc[i][st] - m[i]);
                l[i] += pr;
                const float w = pr * sv;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = __builtin_fmaf(w, vf[e], acc[i][e]);
            }
        }
        c0 += 4 * CHUNK;
        if (c0 >= t1) break;
        ZL_LOAD_CHUNK_Q8(c0)
    }
#undef ZL_LOAD_CHUNK_Q8

#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float m2 = __shfl_xor(m[i], off, 64), l2 = __shfl_xor(l[i], off, 64);
            const float mn = fmaxf(m[i], m2);
            const float f1 = __expf(m[i] - mn), f2 = __expf(m2 - mn);
            l[i] = l[i] * f1 + l2 * f2;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float a2 = __shfl_xor(acc[i][e], off, 64);
                acc[i][e] = acc[i][e] * f1 + a2 * f2;
            }
            m[i] = mn;
        }
    }
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) xw[wave][i][dp * 16 + e] = acc[i][e];
            if (dp == 0) {
                xw[wave][i][D] = m[i];
                xw[wave][i][D + 1] = l[i];
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < RT * D; idx += 256) {
        const int i = idx / D, d = idx % D;
        const int rr = pass * RT + i;
        if (rr >= p.rows) continue;
        float mn = xw[0][i][D];
#pragma unroll
        for (int w = 1; w < 4; ++w) mn = fmaxf(mn, xw[w][i][D]);
        float a = 0.f, lt = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __expf(xw[w][i][D] - mn);
            a = __builtin_fmaf(xw[w][i][d], f, a);
            lt = __builtin_fmaf(xw[w][i][D + 1], f, lt);
        }
        const int qi = rr / p.n_rep, head = hk * p.n_rep + rr % p.n_rep;
        float* dst = p.ws + ((((size_t)b * p.len_q + qi) * p.h + head) * p.max_splits + split) * (D + 2);
        dst[d] = a;
        if (d == 0) {
            dst[D] = mn;
            dst[D + 1] = lt;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Matrix-core decode attention (fp16, D = 128, prefix visibility, len_q * n_rep <= 16 query rows per kv head).
// The VALU kernel above is instruction-issue bound once the batch fills the chip (batch 32, seq 1024: 35 us
// with the KV cache-resident vs 49 us from HBM; ~660 VALU instructions per 32-key chunk and wave).  Here a
// 32-key chunk costs 24 MFMAs + ~100 VALU:
//   S^T = K . Q^T      K rows are the A operand straight from global memory (lane = key, 8 d per k-chunk),
//                      Q^T the B operand (rows beyond len_q * n_rep are zero); C: lane = (query m = lane & 15,
//                      keys 4 kq + i of each 16-key block) -- a lane's scores all belong to ONE query row.
//   softmax            per lane + two cross-lane steps for the row maximum; fp32 probabilities enter the product
//                      as a hi + lo pair of fp16 values (the reference's decode kernel multiplies fp32
//                      probabilities; a single fp16 rounding of p moved INT8-route logits by 1e-2).
//   O^T = V^T . P^T    P^T (this lane's own eight probabilities) IS the B operand; V^T comes from a wave-private
//                      row-major LDS copy of the V chunk through ds_read_b64_tr_b16 (a 16-lane group reads a
//                      4-key x 16-d block, lane i receives column i: tools/ubench/tr_probe.hip).  C: lane =
//                      (query m, d = 16 db + 4 kq + i): rescaling by e^{m_old - m_new} is lane-local.
// Waves are independent inside the loop (no barrier); splits, workspace format and the combine kernel are
// those of the VALU kernel.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
constexpr int kMD = 128;
constexpr int kMVS = kMD + 16;      // LDS V row stride (halfs): 288 B, conflict-free transpose reads

// DT: ZL_F16 / ZL_BF16 (v_mfma_f32_16x16x32_bf16; the hi + lo probability pair then carries 16 mantissa bits)
template <int DT>
__device__ __forceinline__ f4v mfma_t(uint4 a, uint4 b, f4v c) {
    typedef __bf16 b8v __attribute__((ext_vector_type(8)));
    if constexpr (DT == ZL_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
}

// ---- last-arriver merge (zl_decode_attn_la) ------------------------------------------------------------------------------
// The tail of k_decode_attn_mfma when the split merge happens INSIDE the launch.  xw holds the nw waves' (O[128], m, l) of the
// workgroup's <= 16 query rows.  The workgroup folds them into the split's record, publishes the record with write-through
// (sc1) 8-byte stores, waits for the stores (vmcnt(0)) and counts its arrival on the (task, kv head) pair's word with one
// agent-scope atomic; the workgroup that draws the last ticket re-reads the pair's records with sc1 loads (another CU's
// write-through store is in memory by the time its ticket is; nothing here is cached in this CU's L1) and writes the merged
// attention rows -- k_decode_attn_combine's arithmetic in its order (fp32 records: bit-identical to zl_decode_attn up to 16
// splits), or the merging projection's (half records: w4_i8p.hip MERGE).  The last arriver puts the word back to zero: the
// counters need zeroing once, when the workspace is made.  A pair with a single live split skips all of it and writes its rows.
// Records: fp32 [vh][split][128 acc | max | sum], or (half) fp16 [vh][split][128] normalised rows + fp32 (max, sum) pairs behind them.
#ifndef ZL_LA_BATCH
#define ZL_LA_BATCH 16
#endif
constexpr int kLaBatch = ZL_LA_BATCH;   // split records a last arriver holds in registers at once
// ---- optional timeline probe of the in-launch-merge kernel (build a variant with -DZL_ATTN_PROBE; tools/ubench/probe_attn_la.py)
#ifdef ZL_ATTN_PROBE
__device__ unsigned long long* zl_aprobe_p = nullptr;   // [workgroups * 8 waves][8] wall-clock ticks (100 MHz)
#define ZL_APROBE_INIT()                                                                                           \
    unsigned long long* ap_ = nullptr;                                                                             \
    {                                                                                                              \
        const unsigned wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                        \
        if (zl_aprobe_p && (threadIdx.x & 63) == 0 && wg_ < 8192) ap_ = zl_aprobe_p + ((size_t)wg_ * 8 + (threadIdx.x >> 6)) * 8; \
    }
#define ZL_APROBE(slot) do { if (ap_) ap_[slot] = wall_clock64(); } while (0)
#else
#define ZL_APROBE_INIT() do {} while (0)
#define ZL_APROBE(slot) do {} while (0)
#endif
constexpr int kLaMaxSplits = 64;      // splits per task the launcher allows (the last arriver walks them 16 at a time)

template <int DT>
__device__ __forceinline__ void attn_tail_la(const AttnParams& p, float* xw, int b, int hk, int split, int nw, int ns, unsigned long long* ap_) {
    (void)ap_;
    constexpr int WS = 16 * (kMD + 2);
    const int nthr = nw * 64;
    const bool direct = ns == 1;
    const size_t stat0 = (size_t)p.b * p.len_q * p.h * p.max_splits * (kMD / 2);      // half records: floats in front of the pairs
    // ---- this split's record: thread -> (row i, 4 consecutive d)
    for (int it = threadIdx.x; it < p.rows * (kMD / 4); it += nthr) {
        const int i = it / (kMD / 4), d0 = (it % (kMD / 4)) * 4;
        const float* src = xw + (size_t)i * (kMD + 2);
        float mn = src[kMD];
        for (int w = 1; w < nw; ++w) mn = fmaxf(mn, src[w * WS + kMD]);
        float a[4] = {0.f, 0.f, 0.f, 0.f}, lt = 0.f;
        for (int w = 0; w < nw; ++w) {
            const float f = __expf(src[w * WS + kMD] - mn);
            // (two 8-byte reads: a row is kMD + 2 = 130 floats, so odd rows start 8 bytes off a 16-byte boundary -- ADVICE r05)
            typedef float f2v __attribute__((ext_vector_type(2)));
            const f2v v01 = *reinterpret_cast<const f2v*>(src + w * WS + d0), v23 = *reinterpret_cast<const f2v*>(src + w * WS + d0 + 2);
            const f4v v = (f4v){v01[0], v01[1], v23[0], v23[1]};
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = __builtin_fmaf(v[e], f, a[e]);
            lt = __builtin_fmaf(src[w * WS + kMD + 1], f, lt);
        }
        const int qi = i / p.n_rep, head = hk * p.n_rep + i % p.n_rep;
        const size_t vh = ((size_t)b * p.len_q + qi) * p.h + head;
        const size_t rec = vh * p.max_splits + split;
        if (direct) {                                   // what the merge makes of a single record
            uint16_t o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float ov;
                if (p.half_partials) ov = ((float)(_Float16)(a[e] / lt) * lt) * (1.0f / (lt + 1e-20f));
                else ov = a[e] / (lt + 1e-20f);
                o4[e] = ZT<DT>::from_f32(ov);
            }
            *reinterpret_cast<uint2*>(p.out + vh * kMD + d0) = make_uint2((uint32_t)o4[0] | ((uint32_t)o4[1] << 16), (uint32_t)o4[2] | ((uint32_t)o4[3] << 16));
            continue;
        }
        auto st8 = [](void* dst, uint32_t lo, uint32_t hi) {     // one write-through 8-byte store (global_store_dwordx2 sc1)
            __hip_atomic_store(reinterpret_cast<uint64_t*>(dst), (uint64_t)lo | ((uint64_t)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if (p.half_partials) {
            uint32_t w2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
                w2[e] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)(a[2 * e] / lt)) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)(a[2 * e + 1] / lt)) << 16);
            st8(reinterpret_cast<uint16_t*>(p.ws) + rec * kMD + d0, w2[0], w2[1]);
            if (d0 == 0) st8(p.ws + stat0 + rec * 2, __builtin_bit_cast(uint32_t, mn), __builtin_bit_cast(uint32_t, lt));
        } else {
            float* dst = p.ws + rec * (kMD + 2) + d0;   // 130 floats per record: 8-byte aligned
            st8(dst, __builtin_bit_cast(uint32_t, a[0]), __builtin_bit_cast(uint32_t, a[1]));
            st8(dst + 2, __builtin_bit_cast(uint32_t, a[2]), __builtin_bit_cast(uint32_t, a[3]));
            if (d0 == 0) st8(dst + kMD, __builtin_bit_cast(uint32_t, mn), __builtin_bit_cast(uint32_t, lt));
        }
    }
    if (direct) return;
    // ---- arrival.  The stores above are write-through; once vmcnt is zero they are where an sc1 load of any CU finds them.
    //      (gfx9-family reasoning -- stores counted by vmcnt, sc1 = write-through to the level every CU reads with sc1 -- not the
    //      HIP memory model's: relaxed agent-scope atomics carry the data, the drain orders it.  Pinned to the target below.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "attn_tail_la's hand-off is written for gfx950 (vmcnt-tracked write-through stores); revisit it for any other target"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // ... for every wave of the workgroup; xw is free from here on
    ZL_APROBE(5);
    int* flag = reinterpret_cast<int*>(xw);
    if (threadIdx.x == 0) {
        int* cnt = p.la_cnt + (size_t)b * p.hkv + hk;
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == ns - 1) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = old == ns - 1;
    }
    __syncthreads();
    ZL_APROBE(6);
    if (!*flag) return;
    // ---- the pair's last arriver: thread -> (row i, 4 consecutive d).  Statistics AND record slices of up to 16 splits are
    //      requested together (one memory round trip; the statistics are the same 8 bytes for the 32 threads of a row); more
    //      than kLaBatch splits continue in further batches with the running maximum carried along (flash-decoding's rescale)
    for (int it = threadIdx.x; it < p.rows * (kMD / 4); it += nthr) {
        const int i = it / (kMD / 4), d0 = (it % (kMD / 4)) * 4;
        const int qi = i / p.n_rep, head = hk * p.n_rep + i % p.n_rep;
        const size_t vh = ((size_t)b * p.len_q + qi) * p.h + head;
        float mn = -1e20f, a[4] = {0.f, 0.f, 0.f, 0.f}, z = 0.f;
        for (int u0 = 0; u0 < ns; u0 += kLaBatch) {
            uint64_t lo[kLaBatch], hi[kLaBatch], ml[kLaBatch];
#pragma unroll
            for (int j = 0; j < kLaBatch; ++j) {
                const size_t rec = vh * p.max_splits + min(u0 + j, ns - 1);
                if (p.half_partials) {
                    lo[j] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint16_t*>(p.ws) + (rec * kMD + d0)),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hi[j] = 0;
                    ml[j] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p.ws + stat0 + rec * 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const uint64_t* src = reinterpret_cast<const uint64_t*>(p.ws + rec * (kMD + 2) + d0);
                    lo[j] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hi[j] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ml[j] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p.ws + rec * (kMD + 2) + kMD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            float mb = mn;
#pragma unroll
            for (int j = 0; j < kLaBatch; ++j)
                if (u0 + j < ns) mb = fmaxf(mb, __builtin_bit_cast(float, (uint32_t)ml[j]));
            if (u0 > 0) {                               // (never taken up to 16 splits: the arithmetic there is the merge kernel's)
                const float r = __expf(mn - mb);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] *= r;
                z *= r;
            }
            mn = mb;
#pragma unroll
            for (int j = 0; j < kLaBatch; ++j) {
                if (u0 + j < ns) {
                    const float ms = __builtin_bit_cast(float, (uint32_t)ml[j]), ls = __builtin_bit_cast(float, (uint32_t)(ml[j] >> 32));
                    if (p.half_partials) {              // w4_i8p.hip MERGE: weight l e^(m - M) on the normalised row
                        const float f = ls * __expf(ms - mn);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            a[e] = __builtin_fmaf((float)__builtin_bit_cast(_Float16, (uint16_t)(lo[j] >> (16 * e))), f, a[e]);
                        z += f;
                    } else {                            // k_decode_attn_combine
                        const float f = __expf(ms - mn);
                        a[0] = __builtin_fmaf(__builtin_bit_cast(float, (uint32_t)lo[j]), f, a[0]);
                        a[1] = __builtin_fmaf(__builtin_bit_cast(float, (uint32_t)(lo[j] >> 32)), f, a[1]);
                        a[2] = __builtin_fmaf(__builtin_bit_cast(float, (uint32_t)hi[j]), f, a[2]);
                        a[3] = __builtin_fmaf(__builtin_bit_cast(float, (uint32_t)(hi[j] >> 32)), f, a[3]);
                        z = __builtin_fmaf(ls, f, z);
                    }
                }
            }
        }
        uint16_t o4[4];
        const float zi = 1.0f / (z + 1e-20f);
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = ZT<DT>::from_f32(p.half_partials ? a[e] * zi : a[e] / (z + 1e-20f));
        *reinterpret_cast<uint2*>(p.out + vh * kMD + d0) = make_uint2((uint32_t)o4[0] | ((uint32_t)o4[1] << 16), (uint32_t)o4[2] | ((uint32_t)o4[3] << 16));
    }
    ZL_APROBE(7);
}

#ifndef ZL_ATTN8_WAVES_PER_SIMD
#define ZL_ATTN8_WAVES_PER_SIMD 2      // 8-wave instantiation: 2 = one workgroup per CU (135 VGPRs), 4 = two (128 VGPRs, a few spilled)
#endif
#define ZL_ATTN8_OCC(NW_) ((NW_) == 4 ? 4 : ZL_ATTN8_WAVES_PER_SIMD)
// NW: waves per workgroup the instantiation is built for (4: every launcher; 8: zl_decode_attn_la's long splits -- a whole 1 088-slot
// buffer per workgroup at batch 32, where one workgroup per (task, kv head) fills the chip and needs no split, no record, no merge)
// LA: the in-launch merge (zl_decode_attn_la) is compiled in and the wave count comes from the launch; without it the kernel is the
// round-4 one -- four waves as a compile-time constant, no tail code (the batch-1 step's launch: with the runtime wave count and the
// tail behind a branch it measured 6.95 us against 5.90, profiles/r05_decode_kernel_stats.csv history in DESIGN 5.R5)
// PF: chunks requested ahead of the one in use.  1 = rounds 4-5 (the next chunk's loads go out when this chunk's V rows are in LDS).
// 2 (round 6, the in-launch-merge launches of decode batches): a second K / V register set, the chunk after next in flight as well --
// at batch >= 8 a workgroup has its CU to itself (256 workgroups of 4 waves: one wave per SIMD), so the 64 extra VGPRs cost no
// occupancy, and what bounded the launch was the one-chunk-ahead dependency chain, not the bytes
// MASKED (one query row per task, the reference's int8 visibility row instead of a prefix length -- what
// multi_query_attention_rag_buffer is handed, attention_kernel.cu:1252-1457): every key of the buffer is walked, a chunk's 32
// visibility bytes travel with its K / V loads (one byte per lane) and become a wave-uniform 32-bit word by ballot; an invisible
// key scores -inf AND its V row is zeroed before the product (a buffer's tail is not initialised: 0 x NaN must not reach the MFMA).
template <int DT, int NW = 4, bool LA = false, int PF = 1, bool MASKED = false>
__global__ __launch_bounds__(64 * NW, PF >= 3 ? 1 : (PF == 2 || MASKED) ? 2 : ZL_ATTN8_OCC(NW)) void k_decode_attn_mfma(const AttnParams p) {
    // (MASKED: two workgroups per SIMD set -- under the prefix form's 128-register budget the visibility word sent the V set to scratch)
    static_assert(!MASKED || (!LA && PF == 1), "the mask form exists for the two-launch route only");
    __shared__ __attribute__((aligned(16))) uint16_t vs[NW][32 * kMVS];     // 9 KB per wave (36 / 72 KB); reused for the wave merge
    const int b = blockIdx.z, hk = blockIdx.y, split = blockIdx.x;
    const int len = p.buf_lens[b];
    const int vlen_in = MASKED ? 0x7fffffff : p.valid_lens[b];
    const int8_t* mrow = nullptr;
    if constexpr (MASKED) {
        size_t mask_off = 0;
        for (int i = 0; i < b; ++i) mask_off += (size_t)p.buf_lens[i];
        mrow = p.mask + mask_off;                        // len_q == 1: one row of len entries per task
    }
    const uint16_t* kbase = p.k_bufs[b];
    const uint16_t* vbase = p.v_bufs[b];
    const int elen = min(len, vlen_in);
    const int t0 = split * p.split_len;
    const int nw = LA ? __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6)) : NW;   // LA: 1 / 2 / 4 (8) waves, as launched
    ZL_APROBE_INIT();
    ZL_APROBE(0);
    if (t0 >= elen || len <= 0) {
        // LA: a task without a visible key has no arriver at all -- split 0 leaves the rows the merge launch would (zeros)
        if constexpr (LA) {
            if (split == 0)
                for (int idx = threadIdx.x; idx < p.rows * kMD; idx += nw * 64) {
                    const int i = idx / kMD, qi = i / p.n_rep, head = hk * p.n_rep + i % p.n_rep;
                    p.out[(((size_t)b * p.len_q + qi) * p.h + head) * kMD + idx % kMD] = 0;
                }
        }
        return;
    }
    const int t1 = min(elen, t0 + p.split_len);
    const int last_key = t1 - 1;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const size_t kv_stride = p.bshd ? (size_t)p.hkv * kMD : (size_t)kMD;
    const size_t kv_off = p.bshd ? (size_t)hk * kMD : (size_t)hk * len * kMD;

    uint4 kkA[2][4], kkB[PF >= 2 ? 2 : 1][4], kkC[PF >= 3 ? 2 : 1][4];
    uint4 vA0, vA1, vA2, vA3, vA4, vA5, vA6, vA7, vB0, vB1, vB2, vB3, vB4, vB5, vB6, vB7, vC0, vC1, vC2, vC3, vC4, vC5, vC6, vC7;
    int c0 = t0 + wave * 32;
    const int cstep = nw * 32;
    // clamped, branch-free: past the end of the split the loads re-read its last row (cache hits)
#define ZL_MFMA_LOAD_K(KK, base)                                                                                  \
    _Pragma("unroll") for (int blk_ = 0; blk_ < 2; ++blk_) {                                                   \
        const int key_ = (base) + 16 * blk_ + r;                                                               \
        const uint16_t* src_ = kbase + kv_off + (size_t)(key_ < last_key ? key_ : last_key) * kv_stride + 8 * kq; \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) KK[blk_][t_] = *reinterpret_cast<const uint4*>(src_ + 32 * t_); \
    }
    // (the V rows as eight NAMED registers per set -- prefix VS: as an indexed array, or as a struct, the compiler left them on the stack)
#define ZL_MFMA_V1(VS, j_, base)                                                                                 \
    {                                                                                                          \
        const int key_ = (base) + 4 * j_ + (lane >> 4);                                                        \
        VS##j_ = *reinterpret_cast<const uint4*>(vbase + kv_off + (size_t)(key_ < last_key ? key_ : last_key) * kv_stride + (lane & 15) * 8); \
    }
#define ZL_MFMA_LOAD_V(VS, base)                                                                               \
    ZL_MFMA_V1(VS, 0, base) ZL_MFMA_V1(VS, 1, base) ZL_MFMA_V1(VS, 2, base) ZL_MFMA_V1(VS, 3, base) ZL_MFMA_V1(VS, 4, base) \
    ZL_MFMA_V1(VS, 5, base) ZL_MFMA_V1(VS, 6, base) ZL_MFMA_V1(VS, 7, base)
    int mbA = 1, mbX = 1;                                 // MASKED: this lane's visibility byte of the set's chunk (lanes 0..31)
#define ZL_MFMA_LOAD_M(MB, base)                                                                               \
    if constexpr (MASKED) {                                                                                    \
        const int key_ = (base) + (lane & 31);                                                                 \
        MB = mrow[key_ < last_key ? key_ : last_key];                                                          \
    }
    ZL_MFMA_LOAD_M(mbA, c0)                               // first: loads return in order, the ballot must not wait for K and V
    ZL_MFMA_LOAD_K(kkA, c0)
    ZL_MFMA_LOAD_V(vA, c0)
    if constexpr (PF >= 2) {
        ZL_MFMA_LOAD_K(kkB, c0 + cstep)
        ZL_MFMA_LOAD_V(vB, c0 + cstep)
    }
    if constexpr (PF >= 3) {
        ZL_MFMA_LOAD_K(kkC, c0 + 2 * cstep)
        ZL_MFMA_LOAD_V(vC, c0 + 2 * cstep)
    }
    ZL_APROBE(1);

    // Q^T fragments: query row m = r -> (qi, head)
    uint4 qf[4];
    {
        uint4 z = make_uint4(0, 0, 0, 0);
        const bool live = r < p.rows;
        const int rr = live ? r : 0;
        const int qi = rr / p.n_rep, head = hk * p.n_rep + rr % p.n_rep;
        const uint16_t* qp = p.q + (((size_t)b * p.len_q + qi) * p.h + head) * kMD + 8 * kq;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint4 v = *reinterpret_cast<const uint4*>(qp + 32 * t);
            qf[t] = live ? v : z;
        }
    }

    f4v o[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) o[db] = (f4v){0.f, 0.f, 0.f, 0.f};
    float m_run = -1e20f, l_run = 0.f;
    uint16_t* vsw = vs[wave];
    const uint16_t* vtr = vsw + (4 * kq + (r >> 2)) * kMVS + 4 * (r & 3);   // transpose-read address of this lane

    if (c0 >= t1) c0 = -1;                             // a wave without keys: skip the loop, keep the merge
    auto chunk = [&](uint4 (&KK)[2][4], uint4& v0, uint4& v1, uint4& v2, uint4& v3, uint4& v4, uint4& v5, uint4& v6, uint4& v7,
                     int& MB, const int cur) __attribute__((always_inline)) {
        uint32_t vbits = 0xffffffffu;                  // bit j: key cur + j is visible
        if constexpr (MASKED) {
            vbits = (uint32_t)__ballot(lane < 32 && MB != 0 && cur + lane < t1);
            const int vsh = lane >> 4;
#define ZL_MFMA_VZ(j_) if (!((vbits >> (4 * j_ + vsh)) & 1u)) v##j_ = make_uint4(0, 0, 0, 0);
            ZL_MFMA_VZ(0) ZL_MFMA_VZ(1) ZL_MFMA_VZ(2) ZL_MFMA_VZ(3) ZL_MFMA_VZ(4) ZL_MFMA_VZ(5) ZL_MFMA_VZ(6) ZL_MFMA_VZ(7)
#undef ZL_MFMA_VZ
        }
        // ---- S^T = K . Q^T
        f4v st[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            st[blk] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t)
                st[blk] = mfma_t<DT>(KK[blk][t], qf[t], st[blk]);
        }
        // ---- the V chunk to LDS (row-major), then the NEXT chunk's loads: they fly during softmax and P.V
        {
            uint16_t* vdst = vsw + (lane >> 4) * kMVS + (lane & 15) * 8;
#define ZL_MFMA_VST(j_) *reinterpret_cast<uint4*>(vdst + 4 * j_ * kMVS) = v##j_;
            ZL_MFMA_VST(0) ZL_MFMA_VST(1) ZL_MFMA_VST(2) ZL_MFMA_VST(3) ZL_MFMA_VST(4) ZL_MFMA_VST(5) ZL_MFMA_VST(6) ZL_MFMA_VST(7)
#undef ZL_MFMA_VST
        }
        ZL_MFMA_LOAD_M(MB, cur + PF * cstep)
        ZL_MFMA_LOAD_K(KK, cur + PF * cstep)          // this set's next chunk: PF chunks on
        ZL_MFMA_LOAD_V(v, cur + PF * cstep)
        // ---- online softmax of query row r over this lane's 8 keys (+ the 3 other lanes of the row)
        float sv[2][4];
        float mloc = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = cur + 16 * blk + 4 * kq + i;
                bool vis = key < t1;
                if constexpr (MASKED) vis = (vbits >> (16 * blk + 4 * kq + i)) & 1u;      // the ballot already holds key < t1
                sv[blk][i] = vis ? st[blk][i] * p.scale : -INFINITY;
                mloc = fmaxf(mloc, sv[blk][i]);
            }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __expf(m_run - m_new);
#ifdef ZL_ATTN_PROBE
        if (m_run == -1e20f && m_new > -1e19f) ZL_APROBE(2);     // the wave's first scores are in hand
#endif
        m_run = m_new;
        // probabilities as hi + lo fp16 parts (two MFMAs per V fragment): the product sees p to ~2^-22, i.e. the fp32
        // probabilities of the reference's decode kernel, not flash-attention's fp16 ones
        float pr[2][4];
        float lsum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pr[blk][i] = __expf(sv[blk][i] - m_new);
                lsum += pr[blk][i];
            }
        }
        auto cvt = [](float x) -> uint16_t {             // plain RNE conversion (one v_cvt for fp16)
            if constexpr (DT == ZL_F16) return __builtin_bit_cast(uint16_t, (_Float16)x);
            else {
                uint32_t u = __builtin_bit_cast(uint32_t, x);
                return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            }
        };
        auto hi_lo = [&](float a, float b2, uint32_t& hw, uint32_t& lw) {   // two probabilities -> packed hi word, lo word
            const uint16_t ah = cvt(a), bh = cvt(b2);
            hw = (uint32_t)ah | ((uint32_t)bh << 16);
            lw = (uint32_t)cvt(a - ZT<DT>::to_f32(ah)) | ((uint32_t)cvt(b2 - ZT<DT>::to_f32(bh)) << 16);
        };
        uint4 pf, pl;
        hi_lo(pr[0][0], pr[0][1], pf.x, pl.x);
        hi_lo(pr[0][2], pr[0][3], pf.y, pl.y);
        hi_lo(pr[1][0], pr[1][1], pf.z, pl.z);
        hi_lo(pr[1][2], pr[1][3], pf.w, pl.w);
        l_run = l_run * alpha + lsum;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(vtr + 16 * db));
            const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(vtr + 16 * kMVS + 16 * db));
            typedef short s8v __attribute__((ext_vector_type(8)));
            const s8v a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            o[db] = mfma_t<DT>(__builtin_bit_cast(uint4, a), pf, o[db]);
            o[db] = mfma_t<DT>(__builtin_bit_cast(uint4, a), pl, o[db]);
        }
    };
    while (c0 >= 0) {
        chunk(kkA, vA0, vA1, vA2, vA3, vA4, vA5, vA6, vA7, mbA, c0);
        c0 += cstep;
        if (c0 >= t1) break;
        if constexpr (PF >= 2) {
            chunk(kkB, vB0, vB1, vB2, vB3, vB4, vB5, vB6, vB7, mbX, c0);
            c0 += cstep;
            if (c0 >= t1) break;
        }
        if constexpr (PF >= 3) {
            chunk(kkC, vC0, vC1, vC2, vC3, vC4, vC5, vC6, vC7, mbX, c0);
            c0 += cstep;
            if (c0 >= t1) break;
        }
    }
#undef ZL_MFMA_LOAD_K
#undef ZL_MFMA_LOAD_V
#undef ZL_MFMA_LOAD_M
#undef ZL_MFMA_V1

    // ---- merge: the row's normaliser lives in 4 lanes; then the waves through LDS (aliases the V staging)
    ZL_APROBE(3);
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    __syncthreads();
    float* xw = reinterpret_cast<float*>(&vs[0][0]);       // [wave][16 rows][D + 2]
    if (r < p.rows) {
        float* dst = xw + ((size_t)wave * 16 + r) * (kMD + 2);
#pragma unroll
        for (int db = 0; db < 8; ++db) *reinterpret_cast<f4v*>(dst + 16 * db + 4 * kq) = o[db];
        if (kq == 0) {
            dst[kMD] = m_run;
            dst[kMD + 1] = l_run;
        }
    }
    __syncthreads();
    if constexpr (LA) {
        ZL_APROBE(4);
#ifdef ZL_ATTN_PROBE
        attn_tail_la<DT>(p, xw, b, hk, split, nw, (elen + p.split_len - 1) / p.split_len, ap_);
#else
        attn_tail_la<DT>(p, xw, b, hk, split, nw, (elen + p.split_len - 1) / p.split_len, nullptr);
#endif
        return;
    }
    for (int idx = threadIdx.x; idx < p.rows * kMD; idx += NW * 64) {
        const int i = idx / kMD, d = idx % kMD;
        const float* src = xw + (size_t)i * (kMD + 2);
        constexpr int WS = 16 * (kMD + 2);
        float mn = src[kMD];
#pragma unroll
        for (int w = 1; w < NW; ++w) mn = fmaxf(mn, src[w * WS + kMD]);
        float a = 0.f, lt = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float f = __expf(src[w * WS + kMD] - mn);
            a = __builtin_fmaf(src[w * WS + d], f, a);
            lt = __builtin_fmaf(src[w * WS + kMD + 1], f, lt);
        }
        const int qi = i / p.n_rep, head = hk * p.n_rep + i % p.n_rep;
        const size_t rec = (((size_t)b * p.len_q + qi) * p.h + head) * p.max_splits + split;
        if (p.half_partials) {
            uint16_t* hp = reinterpret_cast<uint16_t*>(p.ws);
            // lt >= 1: the split's largest score gives exp(0) -- except, in the mask form, a split without a visible key (lt = 0)
            hp[rec * kMD + d] = __builtin_bit_cast(uint16_t, (_Float16)((MASKED && lt == 0.f) ? 0.f : a / lt));
            if (d == 0) {
                float* st = p.ws + (size_t)p.b * p.len_q * p.h * p.max_splits * (kMD / 2) + rec * 2;
                st[0] = mn;
                st[1] = lt;
            }
            continue;
        }
        float* dst = p.ws + rec * (kMD + 2);
        dst[d] = a;
        if (d == 0) {
            dst[kMD] = mn;
            dst[kMD + 1] = lt;
        }
    }
}

// INT8 KV cache on the matrix cores: the kernel above with u8 codes in HBM (half the bytes per key).  A lane's 16-byte
// K load holds 16 consecutive d of its key = the A fragments of TWO MFMA steps (the contraction order over d is
// permuted accordingly: step 2u + h of lane quarter kq covers d = 16 (kq + 4u) + 8h .. +7, for K and for Q alike);
// codes become exact fp16 (code - 128) through the 0x6400 trick (one v_perm + one v_pk_add per pair); the key's k-scale
// multiplies its score, its v-scale its probability (before the hi + lo split; the normaliser sums the unscaled
// probabilities); V chunks are converted in registers and staged in LDS as fp16, so the P.V half is unchanged.
__global__ __launch_bounds__(256, 3) void k_decode_attn_mfma_q8(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t vs[4][32 * kMVS];
    const int b = blockIdx.z, hk = blockIdx.y, split = blockIdx.x;
    const int len = p.buf_lens[b];
    const int vlen_in = p.valid_lens[b];
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k_bufs[b]);
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v_bufs[b]);
    const float* ksc = p.k_scales[b];
    const float* vsc = p.v_scales[b];
    const int elen = min(len, vlen_in);
    const int t0 = split * p.split_len;
    if (t0 >= elen || len <= 0) return;
    const int t1 = min(elen, t0 + p.split_len);
    const int last_key = t1 - 1;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const size_t kv_stride = p.bshd ? (size_t)p.hkv * kMD : (size_t)kMD;          // bytes
    const size_t kv_off = p.bshd ? (size_t)hk * kMD : (size_t)hk * len * kMD;
    const size_t sc_stride = p.bshd ? (size_t)p.hkv : 1;
    const size_t sc_off = p.bshd ? (size_t)hk : (size_t)hk * len;

    uint4 kk[2][2], vv0, vv1, vv2, vv3;
    float ksv[2][4], vsv[2][4];
    int c0 = t0 + wave * 32;
#define ZL_Q8_LOAD_K(base)                                                                                     \
    _Pragma("unroll") for (int blk_ = 0; blk_ < 2; ++blk_) {                                                   \
        const int key_ = (base) + 16 * blk_ + r;                                                               \
        const uint8_t* src_ = kbase + kv_off + (size_t)(key_ < last_key ? key_ : last_key) * kv_stride + 16 * kq; \
        kk[blk_][0] = *reinterpret_cast<const uint4*>(src_);                                                   \
        kk[blk_][1] = *reinterpret_cast<const uint4*>(src_ + 64);                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                     \
            const int ks_ = (base) + 16 * blk_ + 4 * kq + i_;                                                  \
            const size_t so_ = sc_off + (size_t)(ks_ < last_key ? ks_ : last_key) * sc_stride;                 \
            ksv[blk_][i_] = ksc[so_];                                                                          \
            vsv[blk_][i_] = vsc[so_];                                                                          \
        }                                                                                                      \
    }
    // V: lane loads 16 codes of key (base + 8 j + lane / 8), d = 16 (lane % 8) .. +15
#define ZL_Q8_V1(j_, base)                                                                                     \
    {                                                                                                          \
        const int key_ = (base) + 8 * j_ + (lane >> 3);                                                        \
        vv##j_ = *reinterpret_cast<const uint4*>(vbase + kv_off + (size_t)(key_ < last_key ? key_ : last_key) * kv_stride + (lane & 7) * 16); \
    }
#define ZL_Q8_LOAD_V(base) ZL_Q8_V1(0, base) ZL_Q8_V1(1, base) ZL_Q8_V1(2, base) ZL_Q8_V1(3, base)
    ZL_Q8_LOAD_K(c0)
    ZL_Q8_LOAD_V(c0)

    // Q^T fragments in the permuted d order: step t = 2u + h: d = 16 (kq + 4u) + 8h .. +7
    h8v qf[4];
    {
        uint4 z = make_uint4(0, 0, 0, 0);
        const bool live = r < p.rows;
        const int rr = live ? r : 0;
        const int qi = rr / p.n_rep, head = hk * p.n_rep + rr % p.n_rep;
        const uint16_t* qp = p.q + (((size_t)b * p.len_q + qi) * p.h + head) * kMD + 16 * kq;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint4 v = *reinterpret_cast<const uint4*>(qp + 64 * (t >> 1) + 8 * (t & 1));
            qf[t] = __builtin_bit_cast(h8v, live ? v : z);
        }
    }

    f4v o[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) o[db] = (f4v){0.f, 0.f, 0.f, 0.f};
    float m_run = -1e20f, l_run = 0.f;
    uint16_t* vsw = vs[wave];
    const uint16_t* vtr = vsw + (4 * kq + (r >> 2)) * kMVS + 4 * (r & 3);
    const hv2 bias = {(_Float16)1152.f, (_Float16)1152.f};
    // 8 codes (two words) -> 8 halfs of (code - 128)
    auto cvt8 = [&](uint32_t w0, uint32_t w1) {
        uint4 o4;
        o4.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(hv2, __builtin_amdgcn_perm(0x64646464u, w0, 0x04010400u)) - bias);
        o4.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(hv2, __builtin_amdgcn_perm(0x64646464u, w0, 0x04030402u)) - bias);
        o4.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(hv2, __builtin_amdgcn_perm(0x64646464u, w1, 0x04010400u)) - bias);
        o4.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(hv2, __builtin_amdgcn_perm(0x64646464u, w1, 0x04030402u)) - bias);
        return o4;
    };

    if (c0 >= t1) c0 = -1;
    while (c0 >= 0) {
        // ---- S^T = (K - 128) . Q^T
        f4v st[2];
        float ks_cur[2][4], vs_cur[2][4];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            st[blk] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint4 c = kk[blk][u];
                st[blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, cvt8(c.x, c.y)), qf[2 * u], st[blk], 0, 0, 0);
                st[blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, cvt8(c.z, c.w)), qf[2 * u + 1], st[blk], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ks_cur[blk][i] = ksv[blk][i];
                vs_cur[blk][i] = vsv[blk][i];
            }
        }
        // ---- the V chunk, converted, to LDS as fp16 (row-major); then the NEXT chunk's loads
        {
            uint16_t* vdst = vsw + (lane >> 3) * kMVS + (lane & 7) * 16;
#define ZL_Q8_VST(j_)                                                                                          \
            *reinterpret_cast<uint4*>(vdst + 8 * j_ * kMVS) = cvt8(vv##j_.x, vv##j_.y);                        \
            *reinterpret_cast<uint4*>(vdst + 8 * j_ * kMVS + 8) = cvt8(vv##j_.z, vv##j_.w);
            ZL_Q8_VST(0) ZL_Q8_VST(1) ZL_Q8_VST(2) ZL_Q8_VST(3)
#undef ZL_Q8_VST
        }
        const int cur = c0;
        c0 += 4 * 32;
        ZL_Q8_LOAD_K(c0)
        ZL_Q8_LOAD_V(c0)
        // ---- online softmax
        float sv[2][4];
        float mloc = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = cur + 16 * blk + 4 * kq + i;
                sv[blk][i] = key < t1 ? st[blk][i] * (ks_cur[blk][i] * p.scale) : -INFINITY;
                mloc = fmaxf(mloc, sv[blk][i]);
            }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        h8v pf, pl;
        float lsum = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = cur + 16 * blk + 4 * kq + i;
                const float pr = __expf(sv[blk][i] - m_new);
                lsum += pr;
                const float pw = key < t1 ? pr * vs_cur[blk][i] : 0.f;   // a clamped duplicate's scale is a real one; weight 0 anyway
                const _Float16 ph = (_Float16)pw;
                pf[blk * 4 + i] = ph;
                pl[blk * 4 + i] = (_Float16)(pw - (float)ph);
            }
        }
        l_run = l_run * alpha + lsum;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
        }
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(vtr + 16 * db));
            const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(vtr + 16 * kMVS + 16 * db));
            typedef short s8v __attribute__((ext_vector_type(8)));
            const s8v a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), pf, o[db], 0, 0, 0);
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), pl, o[db], 0, 0, 0);
        }
        if (c0 >= t1) break;
    }
#undef ZL_Q8_LOAD_K
#undef ZL_Q8_LOAD_V
#undef ZL_Q8_V1

    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    __syncthreads();
    float* xw = reinterpret_cast<float*>(&vs[0][0]);
    if (r < p.rows) {
        float* dst = xw + ((size_t)wave * 16 + r) * (kMD + 2);
#pragma unroll
        for (int db = 0; db < 8; ++db) *reinterpret_cast<f4v*>(dst + 16 * db + 4 * kq) = o[db];
        if (kq == 0) {
            dst[kMD] = m_run;
            dst[kMD + 1] = l_run;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < p.rows * kMD; idx += 256) {
        const int i = idx / kMD, d = idx % kMD;
        const float* src = xw + (size_t)i * (kMD + 2);
        constexpr int WS = 16 * (kMD + 2);
        float mn = src[kMD];
#pragma unroll
        for (int w = 1; w < 4; ++w) mn = fmaxf(mn, src[w * WS + kMD]);
        float a = 0.f, lt = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __expf(src[w * WS + kMD] - mn);
            a = __builtin_fmaf(src[w * WS + d], f, a);
            lt = __builtin_fmaf(src[w * WS + kMD + 1], f, lt);
        }
        const int qi = i / p.n_rep, head = hk * p.n_rep + i % p.n_rep;
        float* dst = p.ws + ((((size_t)b * p.len_q + qi) * p.h + head) * p.max_splits + split) * (kMD + 2);
        dst[d] = a;
        if (d == 0) {
            dst[kMD] = mn;
            dst[kMD + 1] = lt;
        }
    }
}

// grid (B*len_q*H), block D.  Split statistics go through LDS once; the per-d accumulation then issues
// independent loads back to back (the first version walked the splits with dependent loads: 6 us).
template <int DT, int D>
__global__ void k_decode_attn_combine(const AttnParams p) {
    constexpr int kMaxS = 512;
    __shared__ float sf[kMaxS];
    __shared__ float red[16];
    const int vh = blockIdx.x;
    const int b = vh / (p.len_q * p.h);
    const int len = p.buf_lens[b];
    const int vlen_in = p.valid_lens ? p.valid_lens[b] : 0x7fffffff;
    const int elen = p.mask ? len : min(len, vlen_in);
    const int ns = min((elen + p.split_len - 1) / p.split_len, kMaxS);
    const int d = threadIdx.x;
    const float* src = p.ws + (size_t)vh * p.max_splits * (D + 2);
    if (ns <= 16) {
        // few splits (batch-1 decode): ONE memory round trip -- the accumulator loads do not wait for the
        // statistics, both are in flight together; the statistics reach every thread through LDS
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(size_t)min(u, max(ns - 1, 0)) * (D + 2) + d];
        if (d < ns) {
            sf[d] = src[(size_t)d * (D + 2) + D];
            sf[64 + d] = src[(size_t)d * (D + 2) + D + 1];
        }
        __syncthreads();
        float mn = -1e20f;
        for (int s = 0; s < ns; ++s) mn = fmaxf(mn, sf[s]);
        float a = 0.f, z = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u < ns) {
                const float f = __expf(sf[u] - mn);
                a = __builtin_fmaf(v[u], f, a);
                z = __builtin_fmaf(sf[64 + u], f, z);
            }
        }
        p.out[(size_t)vh * D + d] = ZT<DT>::from_f32(a / (z + 1e-20f));
        return;
    }
    float mloc = -1e20f;
    for (int s = d; s < ns; s += D) {
        const float ms = src[(size_t)s * (D + 2) + D];
        sf[s] = ms;
        mloc = fmaxf(mloc, ms);
    }
    const float mn = zl_block_max(mloc, red);
    float zloc = 0.f;
    for (int s = d; s < ns; s += D) {
        const float f = __expf(sf[s] - mn);
        sf[s] = f;
        zloc = __builtin_fmaf(src[(size_t)s * (D + 2) + D + 1], f, zloc);
    }
    const float z = zl_block_sum(zloc, red);  // also orders the sf[] writes before the reads below
    float a = 0.f;
    int s = 0;
    for (; s + 4 <= ns; s += 4) {
        const float v0 = src[(size_t)(s + 0) * (D + 2) + d], v1 = src[(size_t)(s + 1) * (D + 2) + d];
        const float v2 = src[(size_t)(s + 2) * (D + 2) + d], v3 = src[(size_t)(s + 3) * (D + 2) + d];
        a = __builtin_fmaf(v0, sf[s], a);
        a = __builtin_fmaf(v1, sf[s + 1], a);
        a = __builtin_fmaf(v2, sf[s + 2], a);
        a = __builtin_fmaf(v3, sf[s + 3], a);
    }
    for (; s < ns; ++s) a = __builtin_fmaf(src[(size_t)s * (D + 2) + d], sf[s], a);
    p.out[(size_t)vh * D + d] = ZT<DT>::from_f32(a / (z + 1e-20f));
}

// The merge of HALF-PRECISION split records (zl_decode_attn_splits_h[_mask]'s format) as a launch of its own: the arithmetic and the
// order of the merging projection's prologue (w4_i8p.hip MERGE), so a caller whose projection did not take the records over gets the
// rows that projection would have read.  grid (B * H), block 128; p.valid_lens may be null (mask form: every split of the buffer).
__global__ __launch_bounds__(kMD) void k_decode_attn_combine_h(const AttnParams p) {
    const int vh = blockIdx.x, b = vh / p.h, d = threadIdx.x;
    const int len = p.buf_lens[b];
    const int elen = p.valid_lens ? min(len, p.valid_lens[b]) : len;
    const int ns = min((elen + p.split_len - 1) / p.split_len, p.max_splits);
    const uint16_t* part = reinterpret_cast<const uint16_t*>(p.ws) + (size_t)vh * p.max_splits * kMD;
    const float2* stat = reinterpret_cast<const float2*>(p.ws + (size_t)p.b * p.h * p.max_splits * (kMD / 2)) + (size_t)vh * p.max_splits;
    float mn = -1e20f;
    for (int u = 0; u < ns; ++u) mn = fmaxf(mn, stat[u].x);
    float a = 0.f, z = 0.f;
    for (int u = 0; u < ns; ++u) {
        const float2 st = stat[u];
        const float f = st.y * __expf(st.x - mn);
        a = __builtin_fmaf((float)__builtin_bit_cast(_Float16, part[(size_t)u * kMD + d]), f, a);
        z += f;
    }
    const float zi = 1.0f / (z + 1e-20f);
    p.out[(size_t)vh * kMD + d] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(a * zi));
}

template <int DT, int D, bool FUSE>
int launch_d(const AttnParams& p, hipStream_t st) {
    dim3 grid((unsigned)p.max_splits, (unsigned)p.hkv, (unsigned)(p.b * p.passes));
    const int rt = p.rows >= 8 ? 8 : (p.rows >= 4 ? 4 : (p.rows >= 2 ? 2 : 1));
#define ZL_ATTN_RT(RT)                                                                                          \
    if (p.mask) hipLaunchKernelGGL((k_decode_attn_partial<DT, D, RT, FUSE, true>), grid, dim3(256), 0, st, p);      \
    else hipLaunchKernelGGL((k_decode_attn_partial<DT, D, RT, FUSE, false>), grid, dim3(256), 0, st, p);
    switch (rt) {
        case 1: ZL_ATTN_RT(1) break;
        case 2: ZL_ATTN_RT(2) break;
        case 4: ZL_ATTN_RT(4) break;
        default: ZL_ATTN_RT(8) break;
    }
#undef ZL_ATTN_RT
    int e = zl_launch_status();
    if (e) return e;
    hipLaunchKernelGGL((k_decode_attn_combine<DT, D>), dim3((unsigned)(p.b * p.len_q * p.h)), dim3(D), 0, st, p);
    return zl_launch_status();
}

template <int DT, int D>
int launch_q8(const AttnParams& p, hipStream_t st) {
    dim3 grid((unsigned)p.max_splits, (unsigned)p.hkv, (unsigned)(p.b * p.passes));
    const int rt = p.rows >= 4 ? 4 : (p.rows >= 2 ? 2 : 1);
#define ZL_ATTN_RT(RT)                                                                                       \
    if (p.mask) hipLaunchKernelGGL((k_decode_attn_partial_q8<DT, D, RT, true>), grid, dim3(256), 0, st, p);  \
    else hipLaunchKernelGGL((k_decode_attn_partial_q8<DT, D, RT, false>), grid, dim3(256), 0, st, p);
    switch (rt) {
        case 1: ZL_ATTN_RT(1) break;
        case 2: ZL_ATTN_RT(2) break;
        default: ZL_ATTN_RT(4) break;
    }
#undef ZL_ATTN_RT
    int e = zl_launch_status();
    if (e) return e;
    hipLaunchKernelGGL((k_decode_attn_combine<DT, D>), dim3((unsigned)(p.b * p.len_q * p.h)), dim3(D), 0, st, p);
    return zl_launch_status();
}

}  // namespace

extern "C" {

int64_t zl_decode_attn_workspace_bytes(int64_t b, int64_t len_q, int64_t h, int64_t d, int64_t max_len_buf) {
    if (b <= 0 || len_q <= 0 || h <= 0 || d <= 0 || max_len_buf <= 0) return ZL_EINVAL;
    // the split length depends on hkv, which is not known here: size for the smallest split (128)
    int64_t splits = (max_len_buf + 127) / 128;
    return b * len_q * h * splits * (d + 2) * 4;
}

int zl_decode_attn(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                   const uint16_t* const* v_bufs, const int8_t* mask, const int32_t* valid_lens, uint16_t* out,
                   void* workspace, int64_t b, int64_t len_q, int64_t h, int64_t hkv, int64_t d, float scale,
                   int64_t max_len_buf, int bshd, int dtype, zl_stream_t s) {
    return zl_decode_attn_ex(q, buf_lens, k_bufs, v_bufs, mask, valid_lens, out, workspace, b, len_q, h, hkv, d, scale, max_len_buf,
                             bshd, dtype, 0, s);
}

int zl_decode_attn_ex(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                      const uint16_t* const* v_bufs, const int8_t* mask, const int32_t* valid_lens, uint16_t* out,
                      void* workspace, int64_t b, int64_t len_q, int64_t h, int64_t hkv, int64_t d, float scale,
                      int64_t max_len_buf, int bshd, int dtype, int algo, zl_stream_t s) {
    ZL_CHECK_ARG(q && buf_lens && k_bufs && v_bufs && out && workspace, ZL_EINVAL);
    ZL_CHECK_ARG(mask || valid_lens, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && len_q > 0 && h > 0 && hkv > 0 && d > 0 && max_len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(h % hkv == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    AttnParams p;
    p.q = q; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs; p.mask = mask; p.valid_lens = valid_lens;
    p.out = out; p.ws = (float*)workspace;
    p.b = (int)b; p.len_q = (int)len_q; p.h = (int)h; p.hkv = (int)hkv; p.n_rep = (int)(h / hkv);
    p.rows = p.len_q * p.n_rep;
    const int rt = p.rows >= 8 ? 8 : (p.rows >= 4 ? 4 : (p.rows >= 2 ? 2 : 1));
    p.passes = (p.rows + rt - 1) / rt;
    p.split_len = attn_split_len(b, hkv, max_len_buf);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    ZL_CHECK_ARG(p.max_splits <= kMaxSplits, ZL_ELIMIT);
    p.scale = scale; p.bshd = bshd;
    ZL_CHECK_ARG((int64_t)p.b * p.passes <= 65535 && hkv <= 65535, ZL_ELIMIT);
    hipStream_t hs = (hipStream_t)s;
    p.qkv = nullptr; p.cosv = p.sinv = nullptr; p.placement = nullptr; p.k_bufs_w = p.v_bufs_w = nullptr; p.neox = 1;
    p.k_scales = p.v_scales = nullptr; p.half_partials = 0; p.la = 0; p.la_cnt = nullptr;
    {   // decode fast path on the matrix cores: all query rows of a kv head in one 16-row MFMA block
        // with the reference's visibility mask instead of prefix lengths: the same kernel's mask form, for one query row per task
        if (algo != 1 && (!mask || (len_q == 1 && !valid_lens)) && d == kMD && p.rows <= 16) {
            p.passes = 1;
            const dim3 grid((unsigned)p.max_splits, (unsigned)hkv, (unsigned)b);
            if (mask) {
                if (dtype == ZL_F16) hipLaunchKernelGGL((k_decode_attn_mfma<ZL_F16, 4, false, 1, true>), grid, dim3(256), 0, hs, p);
                else hipLaunchKernelGGL((k_decode_attn_mfma<ZL_BF16, 4, false, 1, true>), grid, dim3(256), 0, hs, p);
            } else if (dtype == ZL_F16) hipLaunchKernelGGL(k_decode_attn_mfma<ZL_F16>, grid, dim3(256), 0, hs, p);
            else hipLaunchKernelGGL(k_decode_attn_mfma<ZL_BF16>, grid, dim3(256), 0, hs, p);
            int e = zl_launch_status();
            if (e) return e;
            if (dtype == ZL_F16) hipLaunchKernelGGL((k_decode_attn_combine<ZL_F16, kMD>), dim3((unsigned)(b * len_q * h)), dim3(kMD), 0, hs, p);
            else hipLaunchKernelGGL((k_decode_attn_combine<ZL_BF16, kMD>), dim3((unsigned)(b * len_q * h)), dim3(kMD), 0, hs, p);
            return zl_launch_status();
        }
    }
#define ZL_ATTN_D(DT, FUSE)                                    \
    switch (d) {                                               \
        case 64: return launch_d<DT, 64, FUSE>(p, hs);         \
        case 128: return launch_d<DT, 128, FUSE>(p, hs);       \
        case 256: return launch_d<DT, 256, FUSE>(p, hs);       \
        default: return ZL_ESHAPE;                             \
    }
    if (dtype == ZL_F16) { ZL_ATTN_D(ZL_F16, false) }
    ZL_ATTN_D(ZL_BF16, false)
}

int64_t zl_decode_attn_split_len(int64_t b, int64_t hkv, int64_t max_len_buf) {
    if (b <= 0 || hkv <= 0 || max_len_buf <= 0) return ZL_EINVAL;
    return attn_split_len(b, hkv, max_len_buf);
}

int zl_decode_attn_splits(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                          const uint16_t* const* v_bufs, const int32_t* valid_lens, void* workspace, int64_t b, int64_t h,
                          int64_t hkv, int64_t d, float scale, int64_t max_len_buf, int bshd, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(q && buf_lens && k_bufs && v_bufs && valid_lens && workspace, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && h > 0 && hkv > 0 && d > 0 && max_len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(h % hkv == 0 && d == kMD && h / hkv <= 16 && b <= 65535 && hkv <= 65535, ZL_ESHAPE);   // the matrix-core kernel
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    AttnParams p;
    p.q = q; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs; p.mask = nullptr; p.valid_lens = valid_lens;
    p.out = nullptr; p.ws = (float*)workspace;
    p.b = (int)b; p.len_q = 1; p.h = (int)h; p.hkv = (int)hkv; p.n_rep = (int)(h / hkv);
    p.rows = p.n_rep; p.passes = 1;
    p.split_len = attn_split_len(b, hkv, max_len_buf);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    ZL_CHECK_ARG(p.max_splits <= kMaxSplits, ZL_ELIMIT);
    p.scale = scale; p.bshd = bshd;
    p.qkv = nullptr; p.cosv = p.sinv = nullptr; p.placement = nullptr; p.k_bufs_w = p.v_bufs_w = nullptr; p.neox = 1;
    p.k_scales = p.v_scales = nullptr; p.half_partials = 0; p.la = 0; p.la_cnt = nullptr;
    const dim3 grid((unsigned)p.max_splits, (unsigned)hkv, (unsigned)b);
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_decode_attn_mfma<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, p);
    else hipLaunchKernelGGL(k_decode_attn_mfma<ZL_BF16>, grid, dim3(256), 0, (hipStream_t)s, p);
    return zl_launch_status();
}

int zl_decode_attn_splits_h(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                            const uint16_t* const* v_bufs, const int32_t* valid_lens, void* workspace, int64_t b, int64_t h,
                            int64_t hkv, int64_t d, float scale, int64_t max_len_buf, int bshd, zl_stream_t s) {
    ZL_CHECK_ARG(q && buf_lens && k_bufs && v_bufs && valid_lens && workspace, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && h > 0 && hkv > 0 && d > 0 && max_len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(h % hkv == 0 && d == kMD && h / hkv <= 16 && b <= 65535 && hkv <= 65535, ZL_ESHAPE);   // the matrix-core kernel
    AttnParams p;
    p.q = q; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs; p.mask = nullptr; p.valid_lens = valid_lens;
    p.out = nullptr; p.ws = (float*)workspace;
    p.b = (int)b; p.len_q = 1; p.h = (int)h; p.hkv = (int)hkv; p.n_rep = (int)(h / hkv);
    p.rows = p.n_rep; p.passes = 1;
    p.split_len = attn_split_len(b, hkv, max_len_buf);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    ZL_CHECK_ARG(p.max_splits <= kMaxSplits, ZL_ELIMIT);
    p.scale = scale; p.bshd = bshd;
    p.qkv = nullptr; p.cosv = p.sinv = nullptr; p.placement = nullptr; p.k_bufs_w = p.v_bufs_w = nullptr; p.neox = 1;
    p.k_scales = p.v_scales = nullptr; p.half_partials = 1; p.la = 0; p.la_cnt = nullptr;
    const dim3 grid((unsigned)p.max_splits, (unsigned)hkv, (unsigned)b);
    hipLaunchKernelGGL(k_decode_attn_mfma<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, p);
    return zl_launch_status();
}

int zl_decode_attn_splits_h_mask(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs,
                                 const uint16_t* const* v_bufs, const int8_t* mask, void* workspace, int64_t b, int64_t h,
                                 int64_t hkv, int64_t d, float scale, int64_t max_len_buf, int bshd, zl_stream_t s) {
    ZL_CHECK_ARG(q && buf_lens && k_bufs && v_bufs && mask && workspace, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && h > 0 && hkv > 0 && d > 0 && max_len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(h % hkv == 0 && d == kMD && h / hkv <= 16 && b <= 65535 && hkv <= 65535, ZL_ESHAPE);   // the matrix-core kernel
    AttnParams p;
    p.q = q; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs; p.mask = mask; p.valid_lens = nullptr;
    p.out = nullptr; p.ws = (float*)workspace;
    p.b = (int)b; p.len_q = 1; p.h = (int)h; p.hkv = (int)hkv; p.n_rep = (int)(h / hkv);
    p.rows = p.n_rep; p.passes = 1;
    p.split_len = attn_split_len(b, hkv, max_len_buf);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    ZL_CHECK_ARG(p.max_splits <= kMaxSplits, ZL_ELIMIT);
    p.scale = scale; p.bshd = bshd;
    p.qkv = nullptr; p.cosv = p.sinv = nullptr; p.placement = nullptr; p.k_bufs_w = p.v_bufs_w = nullptr; p.neox = 1;
    p.k_scales = p.v_scales = nullptr; p.half_partials = 1; p.la = 0; p.la_cnt = nullptr;
    const dim3 grid((unsigned)p.max_splits, (unsigned)hkv, (unsigned)b);
    hipLaunchKernelGGL((k_decode_attn_mfma<ZL_F16, 4, false, 1, true>), grid, dim3(256), 0, (hipStream_t)s, p);
    return zl_launch_status();
}

int zl_decode_attn_combine_h(const void* workspace, const int32_t* buf_lens, const int32_t* valid_lens, uint16_t* out, int64_t b,
                             int64_t h, int64_t hkv, int64_t max_len_buf, zl_stream_t s) {
    ZL_CHECK_ARG(workspace && buf_lens && out, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && h > 0 && hkv > 0 && max_len_buf > 0, ZL_EINVAL);
    AttnParams p;
    p.q = nullptr; p.buf_lens = buf_lens; p.k_bufs = p.v_bufs = nullptr; p.mask = nullptr; p.valid_lens = valid_lens;
    p.out = out; p.ws = (float*)const_cast<void*>(workspace);
    p.b = (int)b; p.len_q = 1; p.h = (int)h; p.hkv = (int)hkv; p.n_rep = (int)(h / hkv); p.rows = p.n_rep; p.passes = 1;
    p.split_len = attn_split_len(b, hkv, max_len_buf);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    ZL_CHECK_ARG(p.max_splits <= kMaxSplits, ZL_ELIMIT);
    p.scale = 0.f; p.bshd = 1;
    p.qkv = nullptr; p.cosv = p.sinv = nullptr; p.placement = nullptr; p.k_bufs_w = p.v_bufs_w = nullptr; p.neox = 1;
    p.k_scales = p.v_scales = nullptr; p.half_partials = 1; p.la = 0; p.la_cnt = nullptr;
    hipLaunchKernelGGL(k_decode_attn_combine_h, dim3((unsigned)(b * h)), dim3(kMD), 0, (hipStream_t)s, p);
    return zl_launch_status();
}

// ---- decode attention with the split merge inside the launch (last-arriver; k_decode_attn_mfma + attn_tail_la) -------------
static inline int la_split_len(int64_t b, int64_t hkv, int64_t max_len) {
    // the two-launch path's split length (attn_split_len: multiples of 128 keys, 4 waves per workgroup) -- with it the records and
    // the merge are the ones of zl_decode_attn, bit for bit.  Finer splits (32 / 64 keys: every CU pulls a share of a batch-1
    // history) were measured and LOSE: the last arriver's serial chain (write-through drain, ticket, re-read) grows with the
    // record count faster than the per-CU pull shrinks (profiles/r05_attn_la_ab.txt)
    int64_t ls = attn_split_len(b, hkv, max_len);
    // Many (task, kv head) pairs -- a quarter of the CU count or more (batch 8 / 16 / 32 with 8 kv heads): as few splits as give every
    // CU one workgroup -- 4, 2, and NONE at batch 32, where the workgroup walks its whole buffer, writes the attention rows itself
    // and there is no record, no arrival, no re-read.  The per-CU pull is the same either way; what shrinks is the ~4 us tail (fewer
    // records) and the fixed cost per workgroup.  Measured on one box (profiles/r05_attn_la_ab.txt, call 10), tokens/s of the step:
    // batch 32: 8 698 (three 384-key splits) -> 8 966 unsplit with 4 waves (8 833 with 8 waves per workgroup); batch 16: 5 720 -> 5 858
    // (two 576-key splits); batch 8: 3 480 -> 3 501 .. 3 535 (288- / 384-key splits).
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    const int64_t pairs = b * hkv;
    if (pairs * 4 >= cus && max_len <= 16384) {
        const int64_t splits = pairs >= cus ? 1 : (cus + pairs - 1) / pairs;
        ls = ((max_len + splits - 1) / splits + 31) / 32 * 32;
        if (ls < 128) ls = 128;
    }
    while ((max_len + ls - 1) / ls > kLaMaxSplits) ls += 128;
    return (int)ls;
}

int64_t zl_decode_attn_la_split_len(int64_t b, int64_t hkv, int64_t max_len_buf) {
    if (b <= 0 || hkv <= 0 || max_len_buf <= 0) return ZL_EINVAL;
    return la_split_len(b, hkv, max_len_buf);
}

static inline int64_t la_counter_bytes(int64_t b, int64_t hkv) { return (b * hkv * 4 + 255) / 256 * 256; }

int64_t zl_decode_attn_la_workspace_bytes(int64_t b, int64_t h, int64_t hkv, int64_t max_len_buf, int64_t split_len) {
    if (b <= 0 || h <= 0 || hkv <= 0 || max_len_buf <= 0 || split_len < 0 || split_len % 32 != 0) return ZL_EINVAL;
    // split_len == 0: any split length the launcher may pick.  It never runs more than kLaMaxSplits splits (la_split_len grows the
    // length until they fit), so that is the record count to provide for -- not max_len_buf / 32 (545 MB at 32 tasks
    // x 32K keys where 35 MB are used)
    int64_t splits = split_len ? (max_len_buf + split_len - 1) / split_len : std::min<int64_t>(kLaMaxSplits, (max_len_buf + 31) / 32);
    if (splits > kLaMaxSplits) return ZL_ELIMIT;      // an explicit split length the launcher would refuse
    return la_counter_bytes(b, hkv) + b * h * splits * (kMD + 2) * 4;
}

int zl_decode_attn_la(const uint16_t* q, const int32_t* buf_lens, const uint16_t* const* k_bufs, const uint16_t* const* v_bufs,
                      const int32_t* valid_lens, uint16_t* out, void* workspace, int64_t b, int64_t h, int64_t hkv, int64_t d,
                      float scale, int64_t max_len_buf, int bshd, int dtype, int64_t split_len, int half_partials, zl_stream_t s) {
    ZL_CHECK_ARG(q && buf_lens && k_bufs && v_bufs && valid_lens && out && workspace, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && h > 0 && hkv > 0 && d > 0 && max_len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(h % hkv == 0 && d == kMD && h / hkv <= 16 && b <= 65535 && hkv <= 65535, ZL_ESHAPE);   // the matrix-core kernel
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    ZL_CHECK_ARG(!(half_partials & 1) || dtype == ZL_F16, ZL_EDTYPE);
    ZL_CHECK_ARG(split_len >= 0 && split_len % 32 == 0, ZL_EINVAL);
    AttnParams p;
    p.q = q; p.buf_lens = buf_lens; p.k_bufs = k_bufs; p.v_bufs = v_bufs; p.mask = nullptr; p.valid_lens = valid_lens;
    p.out = out;
    p.la = 1; p.la_cnt = reinterpret_cast<int*>(workspace);
    p.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + la_counter_bytes(b, hkv));
    p.b = (int)b; p.len_q = 1; p.h = (int)h; p.hkv = (int)hkv; p.n_rep = (int)(h / hkv);
    p.rows = p.n_rep; p.passes = 1;
    p.split_len = split_len ? (int)split_len : la_split_len(b, hkv, max_len_buf);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    ZL_CHECK_ARG(p.max_splits <= kLaMaxSplits, ZL_ELIMIT);
    ZL_CHECK_ARG((int64_t)b * h * p.max_splits * (kMD + 2) * 4 < ((int64_t)1 << 31), ZL_ELIMIT);
    p.scale = scale; p.bshd = bshd;
    p.qkv = nullptr; p.cosv = p.sinv = nullptr; p.placement = nullptr; p.k_bufs_w = p.v_bufs_w = nullptr; p.neox = 1;
    p.k_scales = p.v_scales = nullptr; p.half_partials = (half_partials & 1) ? 1 : 0;
    // bits 8.. of the flags word: waves per workgroup (0 = 8 from 256-key splits on, 4 from 128, else one per 32 keys) -- A/B only
    int nw = p.split_len >= 128 ? 4 : p.split_len / 32;      // (8 waves per workgroup: measured behind 4 at every batch, on request only)
    const int nw_req = (half_partials >> 8) & 0xff;
    if (nw_req == 1 || nw_req == 2 || nw_req == 4 || nw_req == 8) nw = nw_req * 32 <= p.split_len ? nw_req : nw;
    const dim3 grid((unsigned)p.max_splits, (unsigned)hkv, (unsigned)b);
    if (nw == 8) {
        if (dtype == ZL_F16) hipLaunchKernelGGL((k_decode_attn_mfma<ZL_F16, 8, true>), grid, dim3(512), 0, (hipStream_t)s, p);
        else hipLaunchKernelGGL((k_decode_attn_mfma<ZL_BF16, 8, true>), grid, dim3(512), 0, (hipStream_t)s, p);
    } else {
#ifndef ZL_ATTN_LA_PF
#define ZL_ATTN_LA_PF 2        // (tools/ubench/variant.sh ... -DZL_ATTN_LA_PF=1 rebuilds the one-chunk-ahead launch for A/B runs)
#endif
        if (dtype == ZL_F16) hipLaunchKernelGGL((k_decode_attn_mfma<ZL_F16, 4, true, ZL_ATTN_LA_PF>), grid, dim3(64 * nw), 0, (hipStream_t)s, p);
        else hipLaunchKernelGGL((k_decode_attn_mfma<ZL_BF16, 4, true, ZL_ATTN_LA_PF>), grid, dim3(64 * nw), 0, (hipStream_t)s, p);
    }
    return zl_launch_status();
}

int zl_decode_attn_fused(const float* cosv, const float* sinv, const uint16_t* qkv, const int32_t* placement,
                         const int32_t* buf_lens, const int32_t* valid_lens, uint16_t* const* k_bufs,
                         uint16_t* const* v_bufs, uint16_t* out, void* workspace, int64_t b, int64_t h, int64_t hkv,
                         int64_t d, float scale, int64_t max_len_buf, int neox, int bshd, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(cosv && sinv && qkv && placement && buf_lens && valid_lens && k_bufs && v_bufs && out && workspace, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && h > 0 && hkv > 0 && d > 0 && max_len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(h % hkv == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    AttnParams p;
    p.q = nullptr; p.buf_lens = buf_lens;
    p.k_bufs = reinterpret_cast<const uint16_t* const*>(k_bufs);
    p.v_bufs = reinterpret_cast<const uint16_t* const*>(v_bufs);
    p.mask = nullptr; p.valid_lens = valid_lens; p.out = out; p.ws = (float*)workspace;
    p.b = (int)b; p.len_q = 1; p.h = (int)h; p.hkv = (int)hkv; p.n_rep = (int)(h / hkv);
    p.rows = p.n_rep;
    const int rt = p.rows >= 8 ? 8 : (p.rows >= 4 ? 4 : (p.rows >= 2 ? 2 : 1));
    p.passes = (p.rows + rt - 1) / rt;
    p.split_len = attn_split_len(b, hkv, max_len_buf);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    ZL_CHECK_ARG(p.max_splits <= kMaxSplits, ZL_ELIMIT);
    p.scale = scale; p.bshd = bshd;
    p.qkv = qkv; p.cosv = cosv; p.sinv = sinv; p.placement = placement; p.k_bufs_w = k_bufs; p.v_bufs_w = v_bufs; p.neox = neox;
    p.k_scales = p.v_scales = nullptr; p.half_partials = 0; p.la = 0; p.la_cnt = nullptr;
    ZL_CHECK_ARG((int64_t)p.b * p.passes <= 65535 && hkv <= 65535, ZL_ELIMIT);
    hipStream_t hs = (hipStream_t)s;
    if (dtype == ZL_F16) { ZL_ATTN_D(ZL_F16, true) }
    ZL_ATTN_D(ZL_BF16, true)
#undef ZL_ATTN_D
}

int zl_decode_attn_quant(const uint16_t* q, const int32_t* buf_lens, const uint8_t* const* k_bufs,
                         const uint8_t* const* v_bufs, const float* const* k_scales, const float* const* v_scales,
                         const int8_t* mask, const int32_t* valid_lens, uint16_t* out, void* workspace, int64_t b,
                         int64_t len_q, int64_t h, int64_t hkv, int64_t d, float scale, int64_t max_len_buf, int bshd,
                         int dtype, zl_stream_t s) {
    return zl_decode_attn_quant_ex(q, buf_lens, k_bufs, v_bufs, k_scales, v_scales, mask, valid_lens, out, workspace, b, len_q, h, hkv,
                                   d, scale, max_len_buf, bshd, dtype, 0, s);
}

int zl_decode_attn_quant_ex(const uint16_t* q, const int32_t* buf_lens, const uint8_t* const* k_bufs,
                         const uint8_t* const* v_bufs, const float* const* k_scales, const float* const* v_scales,
                         const int8_t* mask, const int32_t* valid_lens, uint16_t* out, void* workspace, int64_t b,
                         int64_t len_q, int64_t h, int64_t hkv, int64_t d, float scale, int64_t max_len_buf, int bshd,
                         int dtype, int algo, zl_stream_t s) {
    ZL_CHECK_ARG(q && buf_lens && k_bufs && v_bufs && k_scales && v_scales && out && workspace, ZL_EINVAL);
    ZL_CHECK_ARG(mask || valid_lens, ZL_EINVAL);
    ZL_CHECK_ARG(b > 0 && len_q > 0 && h > 0 && hkv > 0 && d > 0 && max_len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(h % hkv == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    AttnParams p;
    p.q = q; p.buf_lens = buf_lens;
    p.k_bufs = reinterpret_cast<const uint16_t* const*>(k_bufs);
    p.v_bufs = reinterpret_cast<const uint16_t* const*>(v_bufs);
    p.k_scales = k_scales; p.v_scales = v_scales; p.half_partials = 0; p.la = 0; p.la_cnt = nullptr;
    p.mask = mask; p.valid_lens = valid_lens; p.out = out; p.ws = (float*)workspace;
    p.b = (int)b; p.len_q = (int)len_q; p.h = (int)h; p.hkv = (int)hkv; p.n_rep = (int)(h / hkv);
    p.rows = p.len_q * p.n_rep;
    const int rt = p.rows >= 4 ? 4 : (p.rows >= 2 ? 2 : 1);
    p.passes = (p.rows + rt - 1) / rt;
    p.split_len = attn_split_len(b, hkv, max_len_buf);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    ZL_CHECK_ARG(p.max_splits <= kMaxSplits, ZL_ELIMIT);
    p.scale = scale; p.bshd = bshd;
    ZL_CHECK_ARG((int64_t)p.b * p.passes <= 65535 && hkv <= 65535, ZL_ELIMIT);
    p.qkv = nullptr; p.cosv = p.sinv = nullptr; p.placement = nullptr; p.k_bufs_w = p.v_bufs_w = nullptr; p.neox = 1;
    hipStream_t hs = (hipStream_t)s;
    {   // decode fast path on the matrix cores
        if (algo != 1 && !mask && dtype == ZL_F16 && d == kMD && p.len_q * p.n_rep <= 16) {
            p.passes = 1;
            hipLaunchKernelGGL(k_decode_attn_mfma_q8, dim3((unsigned)p.max_splits, (unsigned)hkv, (unsigned)b), dim3(256), 0, hs, p);
            int e = zl_launch_status();
            if (e) return e;
            hipLaunchKernelGGL((k_decode_attn_combine<ZL_F16, kMD>), dim3((unsigned)(b * len_q * h)), dim3(kMD), 0, hs, p);
            return zl_launch_status();
        }
    }
    switch (d) {
        case 64: return dtype == ZL_F16 ? launch_q8<ZL_F16, 64>(p, hs) : launch_q8<ZL_BF16, 64>(p, hs);
        case 128: return dtype == ZL_F16 ? launch_q8<ZL_F16, 128>(p, hs) : launch_q8<ZL_BF16, 128>(p, hs);
        case 256: return dtype == ZL_F16 ? launch_q8<ZL_F16, 256>(p, hs) : launch_q8<ZL_BF16, 256>(p, hs);
        default: return ZL_ESHAPE;
    }
}

}  // extern "C"

#ifdef ZL_ATTN_PROBE
extern "C" int zl_debug_set_attn_probe(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(zl_aprobe_p), &p, sizeof(p)); }
#endif

// mla_attn.hip -- f4: multi-head latent attention (DeepSeek MLA, config 5) for decode rows over the COMPRESSED cache.
//
// Reference: Attention::impl::MLAImpl with LATENT_CACHE=1 (src/nn/attention/multi_head_latent_attention.cpp): the cache holds
// one latent row of kv_lora_rank + qk_rope_head_dim values per key (512 + 64), q arrives "absorbed" (q_nope . W_UK | q_rope,
// :1022-1050) and every head attends to the same rows -- as key with all 576 values, as value with the first 512 (:836-872 the
// open gemm + attn_softmax + gemm route; :877-1004 FlashMLA, a closed binary):
//     out[b, h, :512] = softmax_j( scale * q_adj[b, h, :] . kv_b[j, :] ) . kv_b[j, :512],     j < min(buf_len, valid_len)
// Flash-decoding on the VALU, correctness first: scores and probabilities stay in fp32 (the open route rounds both to T; the
// oracle's R flavour restates that, E is fp64), fp32 accumulation, one rounding to T.  A workgroup = 16 heads x one split of the
// keys, wave = 4 heads: a lane is a KEY for the scores (its 576-value row against the heads' q rows broadcast from LDS) and an
// octet of the 512 output columns for the P.V product (probabilities cross over by lane shuffles); partial (max, sum, O) per
// (task, head, split) go through the workspace, k_mla_combine merges them.  128 heads share every cache row, so the bytes are
// small (1.2 MB for 1024 keys) and the kernel is arithmetic- and latency-bound.
//
// k_mla_decode_mfma (the default) is the same split on the matrix cores, built like k_decode_attn_mfma (attention.hip): a wave
// owns 16 keys of a 64-key chunk, S^T = KV . Q^T in 18 k-steps of v_mfma_f32_16x16x32 (the latent rows are the A operand straight
// from global memory, the 16 heads' q rows the B operand, resident in registers), the C layout of S^T is the B layout of P^T, the
// value part of the SAME row fragments goes row-major into wave-private LDS and comes back transposed (ds_read_tr16_b64) as the A
// operand of O^T += V^T . P^T (32 accumulator tiles of 16 output columns, v_mfma_f32_16x16x16); probabilities as hi + lo halves
// (two MFMAs per tile) so that the product sees them to ~2^-22; the four waves merge through LDS.  zl_mla_decode_attn_ex(algo = 1)
// keeps the VALU kernel reachable (tests compare the two).
#include "zl_common.h"

namespace {

constexpr int kRank = 512, kRope = 64, kCD = kRank + kRope;    // latent row: value part | rope part
constexpr int kHG = 16;                                        // heads per workgroup (4 per wave)
constexpr int kRec = kRank + 2;                                // partial record: O[512], max, sum

struct MlaParams {
    const uint16_t* q;                 // (B, H, 576)
    const int32_t* buf_lens;
    const int32_t* valid_lens;         // nullable
    const uint16_t* const* kv_bufs;    // (B) -> (len_buf, 576)
    uint16_t* out;                     // (B, H, 512)
    float* ws;                         // (B, H, max_splits, kRec)
    int b, h, split_len, max_splits;
    float scale;
    // paged form (FlashMLA's cache): rows of task b live in pages of `page` keys (a multiple of 64) of kcache, page i of the task =
    // block_table[b * max_blocks + i]; kv_bufs is unused then
    const uint16_t* kcache;
    const int32_t* block_table;
    int page, max_blocks;
    float* lse;                        // optional (B, H): log-sum-exp of the scaled scores
    int dbg;                           // ablation bits of the wide kernel (zl_debug_mla): 1 no P.V, 2 no scores, 4 no staging waits, 8 no records
};

template <int DT>
__device__ __forceinline__ void unpack8f(const uint4& v, float (&f)[8]) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = ZT<DT>::to_f32((uint16_t)(u[e] & 0xffffu));
        f[2 * e + 1] = ZT<DT>::to_f32((uint16_t)(u[e] >> 16));
    }
}

template <int DT>
__global__ __launch_bounds__(256) void k_mla_decode_partial(const MlaParams p) {
    __shared__ __attribute__((aligned(16))) float qs[kHG * kCD];          // 36.9 KB: the workgroup's 16 q rows in fp32
    const int split = blockIdx.x, hg = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int len = p.valid_lens ? min(p.buf_lens[b], p.valid_lens[b]) : p.buf_lens[b];
    const int t0 = split * p.split_len, t1 = min(len, t0 + p.split_len);
    if (t0 >= len) return;                                                // (workgroup-uniform) the combine skips unwritten splits
    const uint16_t* kv = p.block_table ? nullptr : p.kv_bufs[b];
    for (int i = threadIdx.x; i < kHG * kCD / 8; i += 256) {
        const int hl = i / (kCD / 8), c = i % (kCD / 8), head = hg * kHG + hl;
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (head < p.h) unpack8f<DT>(*reinterpret_cast<const uint4*>(p.q + ((size_t)b * p.h + head) * kCD + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) qs[hl * kCD + c * 8 + e] = f[e];
    }
    __syncthreads();

    float m_run[4], l_run[4], acc[4][8];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
        m_run[hh] = -1e20f;
        l_run[hh] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[hh][e] = 0.f;
    }
    const float* q0 = qs + (wave * 4) * kCD;
    for (int key0 = t0; key0 < t1; key0 += 64) {
        // ---- scores: lane = key key0 + lane; its row against the four q rows
        // a chunk of 64 keys never straddles a page (split_len and page are multiples of 64): one table lookup per chunk
        const uint16_t* chunk = p.block_table ? p.kcache + ((size_t)p.block_table[(size_t)b * p.max_blocks + key0 / p.page] * p.page + key0 % p.page) * kCD
                                              : kv + (size_t)key0 * kCD;
        const int key = key0 + lane;
        const bool live = key < t1;
        const uint4* kp = reinterpret_cast<const uint4*>(chunk + (size_t)(live ? lane : t1 - 1 - key0) * kCD);
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < kCD / 8; ++c) {
            float kf[8];
            unpack8f<DT>(kp[c], kf);
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                const float4 qa = *reinterpret_cast<const float4*>(q0 + hh * kCD + c * 8);
                const float4 qb = *reinterpret_cast<const float4*>(q0 + hh * kCD + c * 8 + 4);
                float a = s[hh];
                a = __builtin_fmaf(kf[0], qa.x, a); a = __builtin_fmaf(kf[1], qa.y, a);
                a = __builtin_fmaf(kf[2], qa.z, a); a = __builtin_fmaf(kf[3], qa.w, a);
                a = __builtin_fmaf(kf[4], qb.x, a); a = __builtin_fmaf(kf[5], qb.y, a);
                a = __builtin_fmaf(kf[6], qb.z, a); a = __builtin_fmaf(kf[7], qb.w, a);
                s[hh] = a;
            }
        }
        // ---- online softmax per head over the 64 keys of the chunk
        float pj[4], alpha[4];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            const float sv = live ? s[hh] * p.scale : -INFINITY;
            const float m_new = fmaxf(m_run[hh], zl_wave_max(sv));
            alpha[hh] = __expf(m_run[hh] - m_new);
            pj[hh] = live ? __expf(sv - m_new) : 0.f;
            l_run[hh] = l_run[hh] * alpha[hh] + zl_wave_sum(pj[hh]);
            m_run[hh] = m_new;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[hh][e] *= alpha[hh];
        }
        // ---- O += P . V: lane = output columns 8 lane .. 8 lane + 7; key j's probability comes from lane j
        const int nk = min(64, t1 - key0);
        const uint16_t* vbase = chunk + lane * 8;
#pragma unroll 4
        for (int j = 0; j < nk; ++j) {
            float vf[8];
            unpack8f<DT>(*reinterpret_cast<const uint4*>(vbase + (size_t)j * kCD), vf);
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                const float pw = __shfl(pj[hh], j, 64);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[hh][e] = __builtin_fmaf(pw, vf[e], acc[hh][e]);
            }
        }
    }
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
        const int head = hg * kHG + wave * 4 + hh;
        if (head >= p.h) continue;
        float* rec = p.ws + (((size_t)b * p.h + head) * p.max_splits + split) * kRec;
        *reinterpret_cast<float4*>(rec + lane * 8) = make_float4(acc[hh][0], acc[hh][1], acc[hh][2], acc[hh][3]);
        *reinterpret_cast<float4*>(rec + lane * 8 + 4) = make_float4(acc[hh][4], acc[hh][5], acc[hh][6], acc[hh][7]);
        if (lane == 0) {
            rec[kRank] = m_run[hh];
            rec[kRank + 1] = l_run[hh];
        }
    }
}

// ---- the matrix-core kernel ------------------------------------------------------------------------------------------------------
typedef float mf4 __attribute__((ext_vector_type(4)));
typedef short ms4 __attribute__((ext_vector_type(4)));
constexpr int kVS = kRank + 16;                                // LDS row stride of the staged value rows (halfs): 1056 B = 8 banks mod 64
constexpr int kMlaStage = 4 * 16 * kVS * 2;                    // bytes of the four waves' staged value rows
constexpr int kMS = kRank + 4;                                 // row stride of the wave merge (floats): 16-byte rows
constexpr int kMlaLds = 4 * 16 * kMS * 4;                      // the wave merge (fp32 [4][16][516]) is the larger use (staging + q: 86 016 B)
static_assert(kMlaStage + 18 * 64 * 16 <= kMlaLds, "the merge buffer covers the loop's LDS");

template <int DT>
__device__ __forceinline__ mf4 mla_mfma32(uint4 a, uint4 b, mf4 c) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    if constexpr (DT == ZL_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ mf4 mla_mfma16(ms4 a, uint2 b, mf4 c) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    if constexpr (DT == ZL_F16) return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, a), __builtin_bit_cast(h4, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(ms4, b), c, 0, 0, 0);
}

// grid (splits, ceil(H / 16), B), 256 threads; dynamic LDS kMlaLds
template <int DT>
__global__ __launch_bounds__(256) void k_mla_decode_mfma(const MlaParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mla_smem[];
    const int split = blockIdx.x, hg = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int len = p.valid_lens ? min(p.buf_lens[b], p.valid_lens[b]) : p.buf_lens[b];
    const int t0 = split * p.split_len, t1 = min(len, t0 + p.split_len);
    if (t0 >= len) {                                                      // (workgroup-uniform) the combine skips unwritten splits
        if (p.max_splits == 1) {                                          // ... and without a combine launch a task without keys gets its
            for (int j = 0; j < 4; ++j) {                                 // zero rows (and lse = -inf) here, as k_mla_combine leaves them
                const int head = hg * kHG + wave * 4 + j;
                if (head >= p.h) break;
                *reinterpret_cast<uint4*>(p.out + ((size_t)b * p.h + head) * kRank + lane * 8) = make_uint4(0, 0, 0, 0);
                if (p.lse && lane == 0) p.lse[(size_t)b * p.h + head] = -INFINITY;
            }
        }
        return;
    }
    const int last_key = t1 - 1;
    const uint16_t* kv = p.block_table ? nullptr : p.kv_bufs[b];

    // Q^T fragments (B operand): lane = (head 16 hg + r, dims 32 t + 8 kq .. + 7); heads past H are zero rows.  They are the same for
    // the four waves and for every piece: fragment t sits in LDS as 64 consecutive 16-byte lane slots (18 KB behind the staging; 72
    // registers otherwise, next to 72 of row fragments and 128 accumulators)
    uint4* qs = reinterpret_cast<uint4*>(mla_smem + kMlaStage);
    {
        const int head = hg * kHG + r;
        const bool live = head < p.h;
        const uint16_t* qp = p.q + ((size_t)b * p.h + (live ? head : 0)) * kCD + 8 * kq;
        uint4 v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = *reinterpret_cast<const uint4*>(qp + 32 * min(wave + 4 * i, 17));     // (one round trip, not five)
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (wave + 4 * i < 18) qs[(wave + 4 * i) * 64 + lane] = live ? v[i] : make_uint4(0, 0, 0, 0);
    }
    // latent-row fragments (A operand): lane = (key c0 + r, dims 32 t + 8 kq .. + 7).  The 16 keys of a wave's piece lie in one
    // 64-key block, hence in one page; a key past the split re-reads its last row (the clamp never leaves the piece)
    uint4 kk[18];
    int c0 = t0 + 16 * wave;
    auto load_rows = [&](int base) {                                      // base: wave-uniform, < t1
        const uint16_t* piece;
        if (p.block_table) {
            const int pg = __builtin_amdgcn_readfirstlane(base / p.page);                  // (scalar: one s_load per piece)
            piece = p.kcache + ((size_t)p.block_table[(size_t)b * p.max_blocks + pg] * p.page + (base - pg * p.page)) * kCD;
        } else {
            piece = kv + (size_t)base * kCD;
        }
        const uint16_t* src = piece + (size_t)(min(base + r, last_key) - base) * kCD + 8 * kq;
        // (the buffer pointer comes out of a table in memory: say that it is global memory, or every load is a flat_load)
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        typedef const u4v __attribute__((address_space(1)))* gptr4;
        gptr4 g = (gptr4)reinterpret_cast<const u4v*>(src);
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            const u4v v = g[4 * t];
            kk[t] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    if (c0 < t1) load_rows(c0);
    else c0 = -1;                                                          // a wave without keys: skip the loop, keep the merge
    __syncthreads();                                                       // the q fragments are in place

    mf4 o[32];
#pragma unroll
    for (int db = 0; db < 32; ++db) o[db] = (mf4){0.f, 0.f, 0.f, 0.f};
    float m_run = -1e20f, l_run = 0.f;
    uint16_t* vsw = reinterpret_cast<uint16_t*>(mla_smem) + (size_t)wave * 16 * kVS;       // this wave's 16 staged value rows
    const uint16_t* vtr = vsw + (4 * kq + (r >> 2)) * kVS + 4 * (r & 3);                    // transpose-read address of this lane

    while (c0 >= 0) {
        // ---- S^T = KV . Q^T: rows = keys, columns = heads
        mf4 st = (mf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 18; ++t) st = mla_mfma32<DT>(kk[t], qs[t * 64 + lane], st);
        // ---- the value part of the same rows to LDS (row-major), then the NEXT piece's loads: they fly during softmax and P.V
        {
            uint16_t* vdst = vsw + r * kVS + 8 * kq;
#pragma unroll
            for (int t = 0; t < 16; ++t) *reinterpret_cast<uint4*>(vdst + 32 * t) = kk[t];
        }
        const int cur = c0;
        c0 += 64;
        if (c0 < t1) load_rows(c0);
        // ---- online softmax of head r over this lane's 4 keys (+ the 3 other lanes of the head)
        float sv[4], mloc = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sv[i] = cur + 4 * kq + i < t1 ? st[i] * p.scale : -INFINITY;
            mloc = fmaxf(mloc, sv[i]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float pr[4], lsum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pr[i] = __expf(sv[i] - m_new);
            lsum += pr[i];
        }
        auto cvt = [](float x) -> uint16_t { return ZT<DT>::from_f32(x); };
        auto hi_lo = [&](float a, float b2, uint32_t& hw, uint32_t& lw) {   // two probabilities -> packed hi word, lo word
            const uint16_t ah = cvt(a), bh = cvt(b2);
            hw = (uint32_t)ah | ((uint32_t)bh << 16);
            lw = (uint32_t)cvt(a - ZT<DT>::to_f32(ah)) | ((uint32_t)cvt(b2 - ZT<DT>::to_f32(bh)) << 16);
        };
        uint2 pf, pl;
        hi_lo(pr[0], pr[1], pf.x, pl.x);
        hi_lo(pr[2], pr[3], pf.y, pl.y);
        // the 128 accumulators live in AGPRs (the VALU cannot touch them without a round trip): rescale only when some head's
        // maximum moved while it had anything accumulated -- otherwise every lane's alpha is exactly 1 (or its column is all zero)
        const bool moved = alpha != 1.0f && l_run > 0.f;
        l_run = l_run * alpha + lsum;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
            for (int db = 0; db < 32; ++db) {
                o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
            }
        }
        // ---- O^T += V^T . P^T: rows = output columns 16 db .. + 15, columns = heads, k = the piece's 16 keys
#pragma unroll
        for (int db = 0; db < 32; ++db) {
            const ms4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ms4*)(vtr + 16 * db));
            o[db] = mla_mfma16<DT>(a, pf, o[db]);
            o[db] = mla_mfma16<DT>(a, pl, o[db]);
        }
        if (c0 >= t1) break;
    }

    // ---- merge: the head's normaliser lives in 4 lanes; then the 4 waves through LDS (aliases the staging)
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    __syncthreads();
    float* xw = reinterpret_cast<float*>(mla_smem);                       // [wave][16 heads][kMS]
    {
        float* dst = xw + ((size_t)wave * 16 + r) * kMS;
#pragma unroll
        for (int db = 0; db < 32; ++db) *reinterpret_cast<mf4*>(dst + 16 * db + 4 * kq) = o[db];
        if (kq == 0) {
            dst[kRank] = m_run;
            dst[kRank + 1] = l_run;
        }
    }
    __syncthreads();
    // wave w writes the records of heads 4 w .. 4 w + 3: lane = 8 output columns, the four waves' rows merged in wave order
    constexpr int WS = 16 * kMS;
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const int i = wave * 4 + j, head = hg * kHG + i;
        if (head >= p.h) break;                                           // (wave-uniform)
        const float* src = xw + (size_t)i * kMS;
        float f[4], mn = src[kRank], lt = 0.f;
#pragma unroll
        for (int w = 1; w < 4; ++w) mn = fmaxf(mn, src[w * WS + kRank]);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            f[w] = __expf(src[w * WS + kRank] - mn);
            lt = __builtin_fmaf(src[w * WS + kRank + 1], f[w], lt);
        }
        mf4 a0 = (mf4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const mf4 x0 = *reinterpret_cast<const mf4*>(src + w * WS + lane * 8), x1 = *reinterpret_cast<const mf4*>(src + w * WS + lane * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] = __builtin_fmaf(x0[e], f[w], a0[e]);
                a1[e] = __builtin_fmaf(x1[e], f[w], a1[e]);
            }
        }
        if (p.max_splits == 1) {
            // ONE split per (task, head) -- batch 32 x 1024 keys: 256 workgroups without any -- needs no record and no combine launch:
            // what k_mla_combine makes of a single record (weight e^0 = 1: the sums are the record's own), written here.  Round 6:
            // 8.4 MB of records out and in again + 4 096 64-lane workgroups for nothing
            const float inv = 1.0f / (lt + 1e-20f);
            if (p.lse && lane == 0) p.lse[(size_t)b * p.h + head] = mn + logf(lt);
            uint32_t w4[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                w4[e] = (uint32_t)ZT<DT>::from_f32(a0[2 * e] * inv) | ((uint32_t)ZT<DT>::from_f32(a0[2 * e + 1] * inv) << 16);
                w4[2 + e] = (uint32_t)ZT<DT>::from_f32(a1[2 * e] * inv) | ((uint32_t)ZT<DT>::from_f32(a1[2 * e + 1] * inv) << 16);
            }
            *reinterpret_cast<uint4*>(p.out + ((size_t)b * p.h + head) * kRank + lane * 8) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            continue;
        }
        float* rec = p.ws + (((size_t)b * p.h + head) * p.max_splits + split) * kRec;
        *reinterpret_cast<float4*>(rec + lane * 8) = make_float4(a0[0], a0[1], a0[2], a0[3]);
        *reinterpret_cast<float4*>(rec + lane * 8 + 4) = make_float4(a1[0], a1[1], a1[2], a1[3]);
        if (lane == 0) {
            rec[kRank] = mn;
            rec[kRank + 1] = lt;
        }
    }
}


// ---- the wide kernel: ALL 128 heads of a task in one workgroup (round 4) ------------------------------------------------------------
// k_mla_decode_mfma gives a workgroup 16 heads, so eight workgroups pull every latent row of a task through their CU's memory path:
// at batch 32 each CU moves 1.18 MB for 59 us (20 GB/s, the per-CU rate of any streaming kernel here) -- 8 x the cache bytes.
// Here the WAVES split the heads (wave w = heads 16 w .. 16 w + 15, 8 waves) and share the rows through LDS, so a row enters the CU
// once: a 16-key piece (16 x 1152 B) is moved global -> LDS by LDS-DMA (18 global_load_lds_dwordx4 spread over the eight waves, three
// pieces ahead of the one being consumed, non-temporal: nobody else reads these rows), every wave reads it as the A operand of
// S^T = KV . Q^T (18 ds_read_b128; its own heads' q fragments live in registers) and, transposed (ds_read_tr16_b64), as the A
// operand of O^T += V^T . P^T -- the arithmetic, the hi + lo probability split and the record format are k_mla_decode_mfma's, and
// a wave owns its heads for ALL keys of the split, so there is no wave merge at the end.  One barrier per piece.
// LDS image of a piece = FRAGMENT-major: 16-byte unit (column unit cu = 0..71, row rho = 0..15) sits at unit index 16 cu + rho, so
// the A fragment of k-step t is 1 KiB of consecutive units in lane order (lane = 16 kq + r reads unit 16 (4 t + kq) + r:
// conflict-free by construction) and DMA d of a piece (64 consecutive units) gathers column units 4 d .. 4 d + 3 of the 16 rows
// (64 contiguous bytes per row).  A transposed read of (row rho, columns c .. c + 3) finds them inside unit 16 (c / 8) + rho.
// grid (splits, B), 512 threads, H == 128.
constexpr int kWPiece = 16 * kCD * 2;                          // bytes per staged 16-key piece (18 432), no padding
constexpr int kWAhead = 7, kWRing = kWAhead + 1;               // pieces requested ahead of the one consumed (126 KB per CU in flight: with 3 a piece
constexpr int kWLds = kWRing * kWPiece;                        // waited 1.0 us for its rows, tools/ubench/mla_ablate.py); ring slots; 147 456 B
constexpr int kWideMinBatch = 16, kWideMinLen = 2048;          // zl_mla_decode_attn takes the wide kernel from this many tasks / keys on (H == 128)

__device__ __forceinline__ void mla_dma16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int DT>
__global__ __launch_bounds__(512, 1) void k_mla_decode_wide(const MlaParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mla_smem[];
    const int split = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int len = p.valid_lens ? min(p.buf_lens[b], p.valid_lens[b]) : p.buf_lens[b];
    const int t0 = split * p.split_len, t1 = min(len, t0 + p.split_len);
    if (t0 >= len) return;                                                // (workgroup-uniform) the combine skips unwritten splits
    const int last_key = t1 - 1, pieces = (t1 - t0 + 15) >> 4;
    const uint16_t* kv = p.block_table ? nullptr : p.kv_bufs[b];

    // this wave's Q^T fragments (B operand): lane = (head 16 wave + r, dims 32 t + 8 kq .. + 7)
    uint4 qf[18];
    {
        const uint16_t* qp = p.q + ((size_t)b * p.h + wave * 16 + r) * kCD + 8 * kq;
#pragma unroll
        for (int t = 0; t < 18; ++t) qf[t] = *reinterpret_cast<const uint4*>(qp + 32 * t);
    }
    // DMA plan: wave w issues DMAs d = w, w + 8, w + 16 (< 18) of a piece; lane = (column unit 4 d + kq, row r)
    const int nd = wave < 2 ? 3 : 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)mla_smem;
    auto issue_piece = [&](int i) {                                       // i: wave-uniform, < pieces
        const int base = t0 + 16 * i;
        const uint16_t* piece;
        if (p.block_table) {
            const int pg = __builtin_amdgcn_readfirstlane(base / p.page);
            piece = p.kcache + ((size_t)p.block_table[(size_t)b * p.max_blocks + pg] * p.page + (base - pg * p.page)) * kCD;
        } else {
            piece = kv + (size_t)base * kCD;
        }
        const uint64_t pa = (uint64_t)piece;
        const uint32_t pa_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pa), pa_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32));
        const void* sb = (const void*)(((uint64_t)pa_hi << 32) | (uint64_t)pa_lo);       // (readfirstlane returns int: no sign extension)
        const uint32_t slot = lds0 + (uint32_t)(i % kWRing) * kWPiece;
        const uint32_t rowoff = (uint32_t)min(r, last_key - base) * (uint32_t)(kCD * 2) + (uint32_t)kq * 16u;   // (a key past the split re-reads the split's last row)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int d = wave + 8 * j;
            if (d < 18) mla_dma16(sb, rowoff + (uint32_t)d * 64u, __builtin_amdgcn_readfirstlane(slot + (uint32_t)d * 1024u));
        }
    };
    for (int i = 0; i < kWAhead && i < pieces; ++i) issue_piece(i);

    mf4 o[32];
#pragma unroll
    for (int db = 0; db < 32; ++db) o[db] = (mf4){0.f, 0.f, 0.f, 0.f};
    float m_run = -1e20f, l_run = 0.f;

    for (int i = 0; i < pieces; ++i) {
        // this wave's DMAs of piece i have landed (pieces i + 1 .. i + kWAhead - 1 may still fly: nd DMAs each, at most 18) ...
        const int after = min(kWAhead - 1, pieces - 1 - i);
        if (!(p.dbg & 4)) {
            switch (after * nd) {                                          // (wave-uniform; s_waitcnt takes an immediate)
#define ZL_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
                ZL_W(0) ZL_W(2) ZL_W(3) ZL_W(4) ZL_W(6) ZL_W(8) ZL_W(9) ZL_W(10) ZL_W(12) ZL_W(15) ZL_W(18)
#undef ZL_W
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
        if (!(p.dbg & 4)) __syncthreads();                                 // ... and everybody's; piece i - 1 has been consumed by all waves
        if (i + kWAhead < pieces && !(p.dbg & 4)) issue_piece(i + kWAhead);   // into the slot piece i - 1 sat in
        const unsigned char* stage = mla_smem + (size_t)(i % kWRing) * kWPiece;
        const int cur = t0 + 16 * i;
        // ---- S^T = KV . Q^T: rows = the piece's 16 keys, columns = this wave's 16 heads
        mf4 st = (mf4){0.f, 0.f, 0.f, 0.f};
        const unsigned char* arow = stage + lane * 16;
        if (!(p.dbg & 2)) {
#pragma unroll
        for (int t = 0; t < 18; ++t) st = mla_mfma32<DT>(*reinterpret_cast<const uint4*>(arow + 1024 * t), qf[t], st);
        }
        // ---- online softmax of head r over this lane's 4 keys (+ the 3 other lanes of the head)
        float sv[4], mloc = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sv[e] = cur + 4 * kq + e < t1 ? st[e] * p.scale : -INFINITY;
            mloc = fmaxf(mloc, sv[e]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float pr[4], lsum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pr[e] = __expf(sv[e] - m_new);
            lsum += pr[e];
        }
        auto cvt = [](float x) -> uint16_t { return ZT<DT>::from_f32(x); };
        auto hi_lo = [&](float a, float b2, uint32_t& hw, uint32_t& lw) {
            const uint16_t ah = cvt(a), bh = cvt(b2);
            hw = (uint32_t)ah | ((uint32_t)bh << 16);
            lw = (uint32_t)cvt(a - ZT<DT>::to_f32(ah)) | ((uint32_t)cvt(b2 - ZT<DT>::to_f32(bh)) << 16);
        };
        uint2 pf, pl;
        hi_lo(pr[0], pr[1], pf.x, pl.x);
        hi_lo(pr[2], pr[3], pf.y, pl.y);
        const bool moved = alpha != 1.0f && l_run > 0.f;
        l_run = l_run * alpha + lsum;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
            for (int db = 0; db < 32; ++db) {
                o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
            }
        }
        // ---- O^T += V^T . P^T: rows = output columns 16 db .. + 15, columns = heads, k = the piece's 16 keys.  This lane's four
        //      values: row rho = 4 kq + (r >> 2), columns 16 db + 4 (r & 3) .. + 3 = half (r & 1) of unit 16 (2 db + ((r & 3) >> 1)) + rho
        const unsigned char* vtr = stage + (size_t)((((r & 3) >> 1) * 16 + 4 * kq + (r >> 2)) * 16 + (r & 1) * 8);
        if (!(p.dbg & 1))
#pragma unroll
        for (int db = 0; db < 32; ++db) {
            const ms4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ms4*)(vtr + 512 * db));
            o[db] = mla_mfma16<DT>(a, pf, o[db]);
            o[db] = mla_mfma16<DT>(a, pl, o[db]);
        }
    }
    // ---- the records of this wave's 16 heads: lane = (head r, output columns 16 db + 4 kq .. + 3)
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    float* rec = p.ws + (((size_t)b * p.h + wave * 16 + r) * p.max_splits + split) * kRec;
    if (p.dbg & 8) {
        if (kq == 0) rec[0] = o[0][0] + o[31][3];
        return;
    }
#pragma unroll
    for (int db = 0; db < 32; ++db) *reinterpret_cast<mf4*>(rec + 16 * db + 4 * kq) = o[db];
    if (kq == 0) {
        rec[kRank] = m_run;
        rec[kRank + 1] = l_run;
    }
}

// grid (H, B), 64 lanes: lane = 8 output columns; the splits that exist are the first ceil(len / split_len)
template <int DT>
__global__ __launch_bounds__(64) void k_mla_combine(const MlaParams p) {
    const int head = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int len = p.valid_lens ? min(p.buf_lens[b], p.valid_lens[b]) : p.buf_lens[b];
    const int ns = len > 0 ? min((len + p.split_len - 1) / p.split_len, p.max_splits) : 0;
    const float* rec0 = p.ws + ((size_t)b * p.h + head) * p.max_splits * kRec;
    // the records' (max, sum) pairs: lane s of a block of 64 records loads its own -- one round trip per block instead of one per record
    float mx = -1e20f;
    for (int s0 = 0; s0 < ns; s0 += 64) mx = fmaxf(mx, zl_wave_max(s0 + lane < ns ? rec0[(size_t)(s0 + lane) * kRec + kRank] : -1e20f));
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, z = 0.f;
    for (int s0 = 0; s0 < ns; s0 += 64) {
        const bool mine = s0 + lane < ns;
        const float fl = mine ? __expf(rec0[(size_t)(s0 + lane) * kRec + kRank] - mx) : 0.f;
        const float zl = mine ? rec0[(size_t)(s0 + lane) * kRec + kRank + 1] : 0.f;
        const int cnt = min(64, ns - s0);
#pragma unroll 4
        for (int j = 0; j < cnt; ++j) {                                  // record order: the sums are order-dependent
            const float* rec = rec0 + (size_t)(s0 + j) * kRec;
            const float f = __shfl(fl, j, 64);
            const float4 a = *reinterpret_cast<const float4*>(rec + lane * 8), c = *reinterpret_cast<const float4*>(rec + lane * 8 + 4);
            o[0] = __builtin_fmaf(a.x, f, o[0]); o[1] = __builtin_fmaf(a.y, f, o[1]); o[2] = __builtin_fmaf(a.z, f, o[2]); o[3] = __builtin_fmaf(a.w, f, o[3]);
            o[4] = __builtin_fmaf(c.x, f, o[4]); o[5] = __builtin_fmaf(c.y, f, o[5]); o[6] = __builtin_fmaf(c.z, f, o[6]); o[7] = __builtin_fmaf(c.w, f, o[7]);
            z = __builtin_fmaf(__shfl(zl, j, 64), f, z);
        }
    }
    const float inv = ns > 0 ? 1.0f / (z + 1e-20f) : 0.f;
    if (p.lse && lane == 0) p.lse[(size_t)b * p.h + head] = ns > 0 ? mx + logf(z) : -INFINITY;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (uint32_t)ZT<DT>::from_f32(o[2 * e] * inv) | ((uint32_t)ZT<DT>::from_f32(o[2 * e + 1] * inv) << 16);
    *reinterpret_cast<uint4*>(p.out + ((size_t)b * p.h + head) * kRank + lane * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

inline int mla_split_len(int64_t b, int64_t h, int64_t max_len, int algo) {
    // about one workgroup of (16 heads, split) per CU for the matrix-core kernel (its 129 KB of LDS leave one workgroup per CU; every
    // further split is another record per head for the combine), about 512 for the VALU kernel; splits of 64 keys (one chunk) upward
    if (algo == 2) {
        // the wide kernel: one workgroup per (task, split); a split costs a record per head (2 KB), so at most 8 of them -- 256
        // workgroups from batch 32 on, fewer below (batch 8: 64 workgroups of 128 keys pull 147 KB each)
        int64_t want = (256 + b - 1) / b;
        if (want > 8) want = 8;
        if (want < 1) want = 1;
        int64_t ls = ((max_len + want - 1) / want + 63) / 64 * 64;
        return (int)(ls < 64 ? 64 : ls);
    }
    const int64_t groups = b * ((h + kHG - 1) / kHG), target = algo == 1 ? 512 : 256;
    int64_t want_splits = (target + groups - 1) / groups;
    if (want_splits < 1) want_splits = 1;
    int64_t ls = ((max_len + want_splits - 1) / want_splits + 63) / 64 * 64;
    if (ls < 64) ls = 64;
    return (int)ls;
}

int mla_launch(const MlaParams& p, int dtype, int algo, hipStream_t hs);

}  // namespace

// ablation bits of the wide kernel: only a -DZL_MLA_DEBUG build can set them (a process-global switch that skips waits
// and barriers must not be reachable from a production library; without the macro the public entry points always pass 0)
#ifdef ZL_MLA_DEBUG
static int zl_mla_dbg = 0;
extern "C" void zl_debug_mla(int bits) { zl_mla_dbg = bits; }
#else
static constexpr int zl_mla_dbg = 0;
#endif

extern "C" {

int64_t zl_mla_decode_workspace_bytes(int64_t b, int64_t h, int64_t max_len_buf) {
    if (b <= 0 || h <= 0 || max_len_buf <= 0) return ZL_EINVAL;
    int64_t ms = 1;                                                       // the finest of the three plans: enough for any kernel
    for (int algo = 0; algo < 3; ++algo) {
        const int ls = mla_split_len(b, h, max_len_buf, algo);
        const int64_t m = (max_len_buf + ls - 1) / ls;
        if (m > ms) ms = m;
    }
    return b * h * ms * kRec * 4;
}

int zl_mla_decode_attn(const uint16_t* q_adj, const int32_t* buf_lens, const int32_t* valid_lens, const uint16_t* const* kv_bufs, uint16_t* out,
                       void* workspace, int64_t b, int64_t h, int64_t kv_lora_rank, int64_t rope_dim, float scale, int64_t max_len_buf, int dtype,
                       zl_stream_t s) {
    return zl_mla_decode_attn_ex(q_adj, buf_lens, valid_lens, kv_bufs, out, workspace, b, h, kv_lora_rank, rope_dim, scale, max_len_buf, dtype, 0, s);
}

int zl_mla_decode_attn_ex(const uint16_t* q_adj, const int32_t* buf_lens, const int32_t* valid_lens, const uint16_t* const* kv_bufs, uint16_t* out,
                          void* workspace, int64_t b, int64_t h, int64_t kv_lora_rank, int64_t rope_dim, float scale, int64_t max_len_buf, int dtype,
                          int algo, zl_stream_t s) {
    ZL_CHECK_ARG(algo >= 0 && algo <= 3, ZL_EINVAL);                      // 0 pick, 1 VALU, 2 wide, 3 the 16-heads-per-workgroup matrix-core kernel
    ZL_CHECK_ARG(algo != 2 || h == 128, ZL_ESHAPE);
    if (algo == 0 && h == 128 && b >= kWideMinBatch && max_len_buf >= kWideMinLen) algo = 2;            // every latent row enters a CU once instead of eight times
    if (algo == 3) algo = 0;
    ZL_CHECK_ARG(q_adj && buf_lens && kv_bufs && out && workspace && b > 0 && h > 0 && max_len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(kv_lora_rank == kRank && rope_dim == kRope && h % 4 == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    ZL_CHECK_ARG(b <= 65535 && (h + kHG - 1) / kHG <= 65535, ZL_ELIMIT);
    MlaParams p;
    p.q = q_adj; p.buf_lens = buf_lens; p.valid_lens = valid_lens; p.kv_bufs = kv_bufs; p.out = out; p.ws = (float*)workspace;
    p.b = (int)b; p.h = (int)h; p.split_len = mla_split_len(b, h, max_len_buf, algo);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    p.scale = scale;
    p.kcache = nullptr; p.block_table = nullptr; p.page = p.max_blocks = 0; p.lse = nullptr; p.dbg = zl_mla_dbg;
    return mla_launch(p, dtype, algo, (hipStream_t)s);
}

int zl_mla_decode_attn_paged(const uint16_t* q_adj, const uint16_t* kcache, const int32_t* block_table, const int32_t* seqlens_k, uint16_t* out,
                             float* softmax_lse, void* workspace, int64_t b, int64_t h, int64_t kv_lora_rank, int64_t rope_dim,
                             int64_t page_block_size, int64_t max_blocks_per_seq, float scale, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(q_adj && kcache && block_table && seqlens_k && out && workspace && b > 0 && h > 0 && max_blocks_per_seq > 0, ZL_EINVAL);
    ZL_CHECK_ARG(kv_lora_rank == kRank && rope_dim == kRope && h % 4 == 0 && page_block_size > 0 && page_block_size % 64 == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    ZL_CHECK_ARG(b <= 65535 && (h + kHG - 1) / kHG <= 65535 && page_block_size * max_blocks_per_seq < ((int64_t)1 << 31), ZL_ELIMIT);
    const int64_t max_len = page_block_size * max_blocks_per_seq;
    MlaParams p;
    p.q = q_adj; p.buf_lens = seqlens_k; p.valid_lens = nullptr; p.kv_bufs = nullptr; p.out = out; p.ws = (float*)workspace;
    const int algo = h == 128 && b >= kWideMinBatch && max_len >= kWideMinLen ? 2 : 0;
    p.b = (int)b; p.h = (int)h; p.split_len = mla_split_len(b, h, max_len, algo);
    p.max_splits = (int)((max_len + p.split_len - 1) / p.split_len);
    p.scale = scale;
    p.kcache = kcache; p.block_table = block_table; p.page = (int)page_block_size; p.max_blocks = (int)max_blocks_per_seq; p.lse = softmax_lse; p.dbg = 0;
    return mla_launch(p, dtype, algo, (hipStream_t)s);
}

}  // extern "C"

namespace {
int mla_launch(const MlaParams& p, int dtype, int algo, hipStream_t hs) {
    const int64_t h = p.h, b = p.b;
    const dim3 grid((unsigned)p.max_splits, (unsigned)((h + kHG - 1) / kHG), (unsigned)b);
    if (algo == 2) {
        const void* fn = dtype == ZL_F16 ? reinterpret_cast<const void*>(&k_mla_decode_wide<ZL_F16>) : reinterpret_cast<const void*>(&k_mla_decode_wide<ZL_BF16>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kWLds) != hipSuccess) return ZL_ELIMIT;
        const dim3 wgrid((unsigned)p.max_splits, (unsigned)b);
        if (dtype == ZL_F16) hipLaunchKernelGGL(k_mla_decode_wide<ZL_F16>, wgrid, dim3(512), kWLds, hs, p);
        else hipLaunchKernelGGL(k_mla_decode_wide<ZL_BF16>, wgrid, dim3(512), kWLds, hs, p);
    } else if (algo == 1) {
        if (dtype == ZL_F16) hipLaunchKernelGGL(k_mla_decode_partial<ZL_F16>, grid, dim3(256), 0, hs, p);
        else hipLaunchKernelGGL(k_mla_decode_partial<ZL_BF16>, grid, dim3(256), 0, hs, p);
    } else {
        const void* fn = dtype == ZL_F16 ? reinterpret_cast<const void*>(&k_mla_decode_mfma<ZL_F16>) : reinterpret_cast<const void*>(&k_mla_decode_mfma<ZL_BF16>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMlaLds) != hipSuccess) return ZL_ELIMIT;
        if (dtype == ZL_F16) hipLaunchKernelGGL(k_mla_decode_mfma<ZL_F16>, grid, dim3(256), kMlaLds, hs, p);
        else hipLaunchKernelGGL(k_mla_decode_mfma<ZL_BF16>, grid, dim3(256), kMlaLds, hs, p);
    }
    int e = zl_launch_status();
    if (e) return e;
    if (algo == 0 && p.max_splits == 1) return 0;      // the matrix-core kernel wrote the rows itself (tasks without a visible key: below)
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_mla_combine<ZL_F16>, dim3((unsigned)h, (unsigned)b), dim3(64), 0, hs, p);
    else hipLaunchKernelGGL(k_mla_combine<ZL_BF16>, dim3((unsigned)h, (unsigned)b), dim3(64), 0, hs, p);
    return zl_launch_status();
}
}  // namespace

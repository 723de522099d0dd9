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
// small (1.2 MB for 1024 keys) and the kernel is arithmetic- and latency-bound: the matrix-core version (S^T = KV . Q^T with
// 18 k-steps, O^T in 32 accumulator tiles per wave) is the next step, not this file.
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

// grid (H, B), 64 lanes: lane = 8 output columns; the splits that exist are the first ceil(len / split_len)
template <int DT>
__global__ __launch_bounds__(64) void k_mla_combine(const MlaParams p) {
    const int head = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int len = p.valid_lens ? min(p.buf_lens[b], p.valid_lens[b]) : p.buf_lens[b];
    const int ns = len > 0 ? min((len + p.split_len - 1) / p.split_len, p.max_splits) : 0;
    const float* rec0 = p.ws + ((size_t)b * p.h + head) * p.max_splits * kRec;
    float mx = -1e20f;
    for (int s = 0; s < ns; ++s) mx = fmaxf(mx, rec0[(size_t)s * kRec + kRank]);
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, z = 0.f;
    for (int s = 0; s < ns; ++s) {
        const float* rec = rec0 + (size_t)s * kRec;
        const float f = __expf(rec[kRank] - mx);
        const float4 a = *reinterpret_cast<const float4*>(rec + lane * 8), c = *reinterpret_cast<const float4*>(rec + lane * 8 + 4);
        o[0] = __builtin_fmaf(a.x, f, o[0]); o[1] = __builtin_fmaf(a.y, f, o[1]); o[2] = __builtin_fmaf(a.z, f, o[2]); o[3] = __builtin_fmaf(a.w, f, o[3]);
        o[4] = __builtin_fmaf(c.x, f, o[4]); o[5] = __builtin_fmaf(c.y, f, o[5]); o[6] = __builtin_fmaf(c.z, f, o[6]); o[7] = __builtin_fmaf(c.w, f, o[7]);
        z = __builtin_fmaf(rec[kRank + 1], f, z);
    }
    const float inv = ns > 0 ? 1.0f / (z + 1e-20f) : 0.f;
    if (p.lse && lane == 0) p.lse[(size_t)b * p.h + head] = ns > 0 ? mx + logf(z) : -INFINITY;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (uint32_t)ZT<DT>::from_f32(o[2 * e] * inv) | ((uint32_t)ZT<DT>::from_f32(o[2 * e + 1] * inv) << 16);
    *reinterpret_cast<uint4*>(p.out + ((size_t)b * p.h + head) * kRank + lane * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

inline int mla_split_len(int64_t b, int64_t h, int64_t max_len) {
    // about 512 workgroups of (16 heads, split): splits of 64 keys (one chunk) upward
    const int64_t groups = b * ((h + kHG - 1) / kHG);
    int64_t want_splits = (512 + groups - 1) / groups;
    if (want_splits < 1) want_splits = 1;
    int64_t ls = ((max_len + want_splits - 1) / want_splits + 63) / 64 * 64;
    if (ls < 64) ls = 64;
    return (int)ls;
}

int mla_launch(const MlaParams& p, int dtype, hipStream_t hs);

}  // namespace

extern "C" {

int64_t zl_mla_decode_workspace_bytes(int64_t b, int64_t h, int64_t max_len_buf) {
    if (b <= 0 || h <= 0 || max_len_buf <= 0) return ZL_EINVAL;
    const int ls = mla_split_len(b, h, max_len_buf);
    const int64_t ms = (max_len_buf + ls - 1) / ls;
    return b * h * ms * kRec * 4;
}

int zl_mla_decode_attn(const uint16_t* q_adj, const int32_t* buf_lens, const int32_t* valid_lens, const uint16_t* const* kv_bufs, uint16_t* out,
                       void* workspace, int64_t b, int64_t h, int64_t kv_lora_rank, int64_t rope_dim, float scale, int64_t max_len_buf, int dtype,
                       zl_stream_t s) {
    ZL_CHECK_ARG(q_adj && buf_lens && kv_bufs && out && workspace && b > 0 && h > 0 && max_len_buf > 0, ZL_EINVAL);
    ZL_CHECK_ARG(kv_lora_rank == kRank && rope_dim == kRope && h % 4 == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    ZL_CHECK_ARG(b <= 65535 && (h + kHG - 1) / kHG <= 65535, ZL_ELIMIT);
    MlaParams p;
    p.q = q_adj; p.buf_lens = buf_lens; p.valid_lens = valid_lens; p.kv_bufs = kv_bufs; p.out = out; p.ws = (float*)workspace;
    p.b = (int)b; p.h = (int)h; p.split_len = mla_split_len(b, h, max_len_buf);
    p.max_splits = (int)((max_len_buf + p.split_len - 1) / p.split_len);
    p.scale = scale;
    p.kcache = nullptr; p.block_table = nullptr; p.page = p.max_blocks = 0; p.lse = nullptr;
    return mla_launch(p, dtype, (hipStream_t)s);
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
    p.b = (int)b; p.h = (int)h; p.split_len = mla_split_len(b, h, max_len);
    p.max_splits = (int)((max_len + p.split_len - 1) / p.split_len);
    p.scale = scale;
    p.kcache = kcache; p.block_table = block_table; p.page = (int)page_block_size; p.max_blocks = (int)max_blocks_per_seq; p.lse = softmax_lse;
    return mla_launch(p, dtype, (hipStream_t)s);
}

}  // extern "C"

namespace {
int mla_launch(const MlaParams& p, int dtype, hipStream_t hs) {
    const int64_t h = p.h, b = p.b;
    const dim3 grid((unsigned)p.max_splits, (unsigned)((h + kHG - 1) / kHG), (unsigned)b);
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_mla_decode_partial<ZL_F16>, grid, dim3(256), 0, hs, p);
    else hipLaunchKernelGGL(k_mla_decode_partial<ZL_BF16>, grid, dim3(256), 0, hs, p);
    int e = zl_launch_status();
    if (e) return e;
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_mla_combine<ZL_F16>, dim3((unsigned)h, (unsigned)b), dim3(64), 0, hs, p);
    else hipLaunchKernelGGL(k_mla_combine<ZL_BF16>, dim3((unsigned)h, (unsigned)b), dim3(64), 0, hs, p);
    return zl_launch_status();
}
}  // namespace

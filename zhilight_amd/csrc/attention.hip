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
#include "zl_common.h"

namespace {

constexpr int kSteps = 8;  // key-steps per chunk

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
};

typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

// split length: multiple of 128 keys, grown so that about >= 1024 workgroups exist when possible
static inline int attn_split_len(int64_t b, int64_t hkv, int64_t max_len) {
    int64_t want = (max_len * b * hkv) / 1024;
    int64_t ls = (want / 128) * 128;
    if (ls < 128) ls = 128;
    if (ls > 2048) ls = 2048;
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

template <int DT, int D, int RT>
__global__ __launch_bounds__(256) void k_decode_attn_partial(const AttnParams p) {
    constexpr int LPK = D / 8;       // lanes per key row
    constexpr int KPS = 64 / LPK;    // keys per wave-step
    constexpr int CHUNK = KPS * kSteps;
    __shared__ float xw[4][RT][D + 2];

    const int b = blockIdx.z / p.passes, pass = blockIdx.z % p.passes;
    const int hk = blockIdx.y, split = blockIdx.x;
    const int len = p.buf_lens[b];
    const int t0 = split * p.split_len;
    if (t0 >= len) return;
    const int t1 = min(len, t0 + p.split_len);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPK, dp = lane % LPK;

    const uint16_t* kbase = p.k_bufs[b];
    const uint16_t* vbase = p.v_bufs[b];
    const size_t kv_stride = p.bshd ? (size_t)p.hkv * D : (size_t)D;
    const size_t kv_off = (p.bshd ? (size_t)hk * D : (size_t)hk * len * D) + dp * 8;

    size_t mask_off = 0;
    if (p.mask) {
        for (int i = 0; i < b; ++i) mask_off += (size_t)p.buf_lens[i];
        mask_off *= p.len_q;
    }
    const int vlen = p.valid_lens ? p.valid_lens[b] : len;

    uint4 qv[RT];
    int q_of_row[RT];
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

    float m[RT], l[RT], acc[RT][8];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        m[i] = -1e20f;
        l[i] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    }

    for (int c0 = t0 + wave * CHUNK; c0 < t1; c0 += 4 * CHUNK) {
        uint4 kk[kSteps], vv[kSteps];
#pragma unroll
        for (int st = 0; st < kSteps; ++st) {
            const int key = c0 + st * KPS + grp;
            kk[st] = make_uint4(0, 0, 0, 0);
            vv[st] = make_uint4(0, 0, 0, 0);
            if (key < t1) {
                kk[st] = *reinterpret_cast<const uint4*>(kbase + kv_off + (size_t)key * kv_stride);
                vv[st] = *reinterpret_cast<const uint4*>(vbase + kv_off + (size_t)key * kv_stride);
            }
        }
        float sc[RT][kSteps];
        bool vis_any[kSteps];
#pragma unroll
        for (int st = 0; st < kSteps; ++st) {
            const int key = c0 + st * KPS + grp;
            vis_any[st] = false;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                float d = dot8<DT>(qv[i], kk[st]);
#pragma unroll
                for (int off = LPK / 2; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
                bool vis = key < t1;
                if (vis) vis = p.mask ? (p.mask[mask_off + (size_t)q_of_row[i] * len + key] != 0) : (key < vlen);
                vis_any[st] = vis_any[st] || vis;
                sc[i][st] = vis ? d * p.scale : -INFINITY;
            }
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
            if (!vis_any[st]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) vf[e] = 0.f;  // never let an unseen (possibly garbage) V row in
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const float pr = __expf(sc[i][st] - m[i]);
                l[i] += pr;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][e] = __builtin_fmaf(pr, vf[e], acc[i][e]);
            }
        }
    }

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

// grid (B*len_q*H), block D
template <int DT, int D>
__global__ void k_decode_attn_combine(const AttnParams p) {
    const int vh = blockIdx.x;
    const int b = vh / (p.len_q * p.h);
    const int len = p.buf_lens[b];
    const int ns = (len + p.split_len - 1) / p.split_len;
    const int d = threadIdx.x;
    const float* src = p.ws + (size_t)vh * p.max_splits * (D + 2);
    float mn = -1e20f;
    for (int s = 0; s < ns; ++s) mn = fmaxf(mn, src[(size_t)s * (D + 2) + D]);
    float a = 0.f, z = 0.f;
    for (int s = 0; s < ns; ++s) {
        const float f = __expf(src[(size_t)s * (D + 2) + D] - mn);
        a = __builtin_fmaf(src[(size_t)s * (D + 2) + d], f, a);
        z = __builtin_fmaf(src[(size_t)s * (D + 2) + D + 1], f, z);
    }
    p.out[(size_t)vh * D + d] = ZT<DT>::from_f32(a / (z + 1e-20f));
}

template <int DT, int D>
int launch_d(const AttnParams& p, hipStream_t st) {
    dim3 grid((unsigned)p.max_splits, (unsigned)p.hkv, (unsigned)(p.b * p.passes));
    const int rt = p.rows >= 8 ? 8 : (p.rows >= 4 ? 4 : (p.rows >= 2 ? 2 : 1));
    switch (rt) {
        case 1: hipLaunchKernelGGL((k_decode_attn_partial<DT, D, 1>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((k_decode_attn_partial<DT, D, 2>), grid, dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL((k_decode_attn_partial<DT, D, 4>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((k_decode_attn_partial<DT, D, 8>), grid, dim3(256), 0, st, p); break;
    }
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
    p.scale = scale; p.bshd = bshd;
    ZL_CHECK_ARG((int64_t)p.b * p.passes <= 65535 && hkv <= 65535, ZL_ELIMIT);
    hipStream_t hs = (hipStream_t)s;
#define ZL_ATTN_D(DT)                                          \
    switch (d) {                                               \
        case 64: return launch_d<DT, 64>(p, hs);               \
        case 128: return launch_d<DT, 128>(p, hs);             \
        case 256: return launch_d<DT, 256>(p, hs);             \
        default: return ZL_ESHAPE;                             \
    }
    if (dtype == ZL_F16) { ZL_ATTN_D(ZL_F16) }
    ZL_ATTN_D(ZL_BF16)
#undef ZL_ATTN_D
}

}  // extern "C"

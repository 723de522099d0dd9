// misc_ops.hip -- the small fused/element-wise kernels of the decode path (SURVEY 8a rows a13, a14,
// a17, a18, a22) plus library utilities.  All are tiny HBM/L2-bound byte movers: 16-byte vector
// accesses, one workgroup per row (or grid-stride), fp32 math with ONE rounding to T at the same
// points as the reference kernels cited at each launcher.
#include <mutex>
#include "zl_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// RMSNorm (+ residual add).  src/nn/layernorm/layernorm.cu:10-42
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void k_rmsnorm(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                 uint16_t* __restrict__ out, int dim, float eps, float scale,
                                                 const uint16_t* __restrict__ x2, uint16_t* __restrict__ out_sum) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* vbuf = reinterpret_cast<float*>(smem);  // dim floats
    float* red = vbuf + dim;
    const size_t off = (size_t)blockIdx.x * dim;
    float ss = 0.f;
    for (int i = threadIdx.x * 8; i < dim; i += 256 * 8) {  // dim % 8 == 0
        uint4 a = *reinterpret_cast<const uint4*>(x + off + i);
        uint4 b = make_uint4(0, 0, 0, 0);
        if (x2) b = *reinterpret_cast<const uint4*>(x2 + off + i);
        const uint32_t au[4] = {a.x, a.y, a.z, a.w}, bu[4] = {b.x, b.y, b.z, b.w};
        uint32_t su[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v0 = ZT<DT>::to_f32((uint16_t)(au[e] & 0xffff)), v1 = ZT<DT>::to_f32((uint16_t)(au[e] >> 16));
            if (x2) {
                v0 += ZT<DT>::to_f32((uint16_t)(bu[e] & 0xffff));
                v1 += ZT<DT>::to_f32((uint16_t)(bu[e] >> 16));
                su[e] = (uint32_t)ZT<DT>::from_f32(v0) | ((uint32_t)ZT<DT>::from_f32(v1) << 16);
            }
            vbuf[i + 2 * e] = v0;
            vbuf[i + 2 * e + 1] = v1;
            ss = __builtin_fmaf(v0, v0, ss);
            ss = __builtin_fmaf(v1, v1, ss);
        }
        if (x2 && out_sum) *reinterpret_cast<uint4*>(out_sum + off + i) = make_uint4(su[0], su[1], su[2], su[3]);
    }
    ss = zl_block_sum(ss, red);
    const float rs = zl_rsqrt_rn(ss / (float)dim + eps);
    for (int i = threadIdx.x * 8; i < dim; i += 256 * 8) {
        uint4 wv = *reinterpret_cast<const uint4*>(w + i);
        const uint32_t wu[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = vbuf[i + 2 * e] * rs * ZT<DT>::to_f32((uint16_t)(wu[e] & 0xffff)) / scale;
            float b = vbuf[i + 2 * e + 1] * rs * ZT<DT>::to_f32((uint16_t)(wu[e] >> 16)) / scale;
            o[e] = (uint32_t)ZT<DT>::from_f32(a) | ((uint32_t)ZT<DT>::from_f32(b) << 16);
        }
        *reinterpret_cast<uint4*>(out + off + i) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// RoPE tables.  src/nn/position/rope_preparer.cu:49-69, 124-160
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int half_dim_index(int col, int half_dim, int neox) {
    return neox ? (col < half_dim ? col : col - half_dim) : col / 2;
}

__device__ __forceinline__ void rope_cos_sin_row(int row, const int32_t* __restrict__ pos, float* __restrict__ cosv, float* __restrict__ sinv,
                                                 int d, float base, int neox, int llama3, float factor, float low_ff, float high_ff,
                                                 float old_ctx) {
    const int col = threadIdx.x;
    if (col >= d) return;
    const int i = half_dim_index(col, d / 2, neox);
    // powf of the reference (rope_preparer.cu:61) with a CORRECTLY ROUNDED result: the angle is position x inv_freq, so one ulp of
    // the device powf (6e-8 relative) is 1e-4 rad at position 1024 -- enough to move a fifth of the rotated fp16 values by an ulp
    // and, on the INT8 route, to flip activation codes (0.3 % logit noise against the CPU oracle, tests/test_gpu_fullgeom.py)
    float inv_freq = (float)pow((double)base, (double)(-(float)(i * 2) / (float)d));
    if (llama3) {
        const float low_wl = old_ctx / low_ff, high_wl = old_ctx / high_ff;
        const float pi = 3.141592653589793f;
        const float wavelen = 2.f * pi / inv_freq;
        if (wavelen < high_wl) {
        } else if (wavelen > low_wl) {
            inv_freq = inv_freq / factor;
        } else {
            const float smooth = (old_ctx / wavelen - low_ff) / (high_ff - low_ff);
            inv_freq = __builtin_fmaf(smooth, inv_freq, (1.f - smooth) * inv_freq / factor);
        }
    }
    const float freq = (float)pos[row] * inv_freq;
    cosv[(size_t)row * d + col] = cosf(freq);
    sinv[(size_t)row * d + col] = sinf(freq);
}
__global__ void k_rope_cos_sin(const int32_t* __restrict__ pos, float* __restrict__ cosv, float* __restrict__ sinv,
                               int d, float base, int neox, int llama3, float factor, float low_ff, float high_ff,
                               float old_ctx) {
    rope_cos_sin_row(blockIdx.x, pos, cosv, sinv, d, base, neox, llama3, factor, low_ff, high_ff, old_ctx);
}

// dynamic-NTK and YaRN angle tables (RotaryEmbedding::impl "dynamic" / YarnImpl, src/nn/position/rotary_embedding.cu:19-61,
// 400-447, 506-553), as cached cos / sin for the fused rotation kernels:
//   dynamic: theta' = theta * ((factor * L / max_pos) - (factor - 1)) ^ (d / (d - 2))   when L > max_pos, L = the row's
//            sequence length (the reference reads the LAST position of the row, :36); the exponent is the reference's
//            INTEGER quotient dim_head / (dim_head - 2) (= 1 for every real head size), reproduced as written
//   yarn:    inv_freq = interp * ramp + extrap * (1 - ramp), ramp = clamp((i - low) / (high - low)) over the pair index i,
//            cos / sin scaled by mscale (low, high, mscale computed on the host in double like YarnImpl's constructor)
__global__ void k_rope_cos_sin_scaled(const int32_t* __restrict__ pos, const int32_t* __restrict__ seq_len,
                                      float* __restrict__ cosv, float* __restrict__ sinv, int d, float base, int neox, int type,
                                      float factor, float p1, float p2, float p3) {
    const int col = threadIdx.x;
    if (col >= d) return;
    const int i = half_dim_index(col, d / 2, neox);
    const int t = blockIdx.x;
    float freq, scale = 1.f;
    if (type == 2) {            // dynamic NTK: p1 = max_position_embeddings
        float theta = base;
        const int len = seq_len ? seq_len[t] : pos[t];
        if ((float)len > p1) theta *= (float)pow((double)((factor * (float)len / p1) - (factor - 1.f)), (double)(float)(d / (d - 2)));
        freq = (float)pos[t] * (float)pow((double)theta, (double)(-(float)(i * 2) / (float)d));
    } else {                    // yarn: p1 = low, p2 = high, p3 = mscale
        const float pos_freq = (float)pow((double)base, (double)((float)(i * 2) / (float)d));
        const float extrap = 1.0f / pos_freq, interp = 1.0f / (factor * pos_freq);
        const float fi = (float)i;
        const float ramp = fi <= p1 ? 0.f : (fi >= p2 ? 1.f : (fi - p1) / (p2 - p1));
        freq = (float)pos[t] * __builtin_fmaf(interp, ramp, extrap * (1.f - ramp));
        scale = p3;
    }
    cosv[(size_t)t * d + col] = cosf(freq) * scale;
    sinv[(size_t)t * d + col] = sinf(freq) * scale;
}

// per-head norms of q / k (grid (rows, heads), block d <= 1024; in place when out == x):
//   mode 0  RMSNorm over dim_head with ONE weight (d) for every head: Qwen3's q_norm / k_norm (attention.cpp:110-113, 871-876:
//           LayerNorm(dim_head) on the (rows, heads, dim_head) view): out = T(v * rsqrt(mean(v^2) + eps) * w[col])
//   mode 1  KERNEL_layernorm_multi_head (layernorm.cu:305-325, use_qk_norm): v = x - mean(x); out = T(v * rsqrt(mean(v^2) + eps) *
//           w[head][col])
template <int DT>
__global__ void k_head_norm(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, uint16_t* __restrict__ out, int d,
                            int64_t ld_in, int64_t ld_out, float eps, int mode) {
    __shared__ float red[16];
    const int col = threadIdx.x, head = blockIdx.y;
    const int64_t row = blockIdx.x;
    float v = col < d ? ZT<DT>::to_f32(x[row * ld_in + (int64_t)head * d + col]) : 0.f;
    if (mode == 1) {
        const float mean = zl_block_sum(v, red) / (float)d;
        v = col < d ? v - mean : 0.f;
    }
    const float var = zl_block_sum(v * v, red) / (float)d;
    const float r = zl_rsqrt_rn(var + eps);
    if (col < d) {
        const float wv = ZT<DT>::to_f32(w[(mode == 1 ? (int64_t)head * d : 0) + col]);
        out[row * ld_out + (int64_t)head * d + col] = ZT<DT>::from_f32(v * r * wv);
    }
}

__device__ __forceinline__ float rope_val(float a, float b, float c, float s, bool minus) {
    return minus ? __builtin_fmaf(-b, s, a * c) : __builtin_fmaf(b, s, a * c);
}

// grid (S, H + 2Hkv), block D.  MODE 0: angles computed (rotary_embedding_fuse.cu:19-67);
// MODE 1: cached cos/sin (rotary_embedding_fuse_cache.cu:23-64)
template <int DT, int MODE>
__global__ void k_rope_qk(const int32_t* __restrict__ pos, const float* __restrict__ cosv,
                          const float* __restrict__ sinv, const uint16_t* __restrict__ in, uint16_t* __restrict__ q,
                          uint16_t* __restrict__ k, uint16_t* __restrict__ v, int h, int hkv, int d, float theta,
                          int neox) {
    const int t = blockIdx.x, head = blockIdx.y, col = threadIdx.x, all = h + 2 * hkv, half = d / 2;
    if (col >= d) return;
    const uint16_t* src = in + ((size_t)t * all + head) * d;
    if (head >= h + hkv) {
        v[((size_t)t * hkv + (head - h - hkv)) * d + col] = src[col];
        return;
    }
    float c, s;
    if (MODE == 0) {
        const int i = col < half ? col : col - half;
        const float freq = (float)pos[t] * (float)pow((double)theta, (double)(-(float)(i * 2) / (float)d));   // correctly rounded powf (see k_rope_cos_sin)
        c = cosf(freq);
        s = sinf(freq);
    } else {
        c = cosv[(size_t)t * d + col];
        s = sinv[(size_t)t * d + col];
    }
    const float a = ZT<DT>::to_f32(src[col]);
    float r;
    if (MODE == 0 || neox)
        r = col < half ? rope_val(a, ZT<DT>::to_f32(src[col + half]), c, s, true)
                       : rope_val(a, ZT<DT>::to_f32(src[col - half]), c, s, false);
    else
        r = (col & 1) == 0 ? rope_val(a, ZT<DT>::to_f32(src[col + 1]), c, s, true)
                           : rope_val(a, ZT<DT>::to_f32(src[col - 1]), c, s, false);
    uint16_t* dst = head >= h ? k + ((size_t)t * hkv + (head - h)) * d : q + ((size_t)t * h + head) * d;
    dst[col] = ZT<DT>::from_f32(r);
}

// Rotation of the heads of a STRIDED tensor: x viewed as (n, heads, d) with element strides (x_sn, x_sh, 1) -> out (o_sn, o_sh, 1),
// cached cos / sin of the rows (n, d); in place allowed (every thread reads its pair before anyone writes).  grid (n, heads), block d.
// RotaryEmbedding::rotate / rotate_inplace on last-dimension slices (MLAImpl: the 64 rope dimensions inside q's 192 and inside the
// fused qkv_a output, multi_head_latent_attention.cpp:527, 540-541, 586-600); same fp32 expression per element as the fused kernels.
template <int DT>
__global__ void k_rope_heads(const float* __restrict__ cosv, const float* __restrict__ sinv, const uint16_t* x, uint16_t* out, int d,
                             int64_t x_sn, int64_t x_sh, int64_t o_sn, int64_t o_sh, int neox) {
    const int t = blockIdx.x, head = blockIdx.y, col = threadIdx.x, half = d / 2;
    const uint16_t* src = x + (int64_t)t * x_sn + (int64_t)head * x_sh;
    float r = 0.f;
    if (col < d) {
        const float c = cosv[(size_t)t * d + col], s = sinv[(size_t)t * d + col];
        const float a = ZT<DT>::to_f32(src[col]);
        if (neox)
            r = col < half ? rope_val(a, ZT<DT>::to_f32(src[col + half]), c, s, true) : rope_val(a, ZT<DT>::to_f32(src[col - half]), c, s, false);
        else
            r = (col & 1) == 0 ? rope_val(a, ZT<DT>::to_f32(src[col + 1]), c, s, true) : rope_val(a, ZT<DT>::to_f32(src[col - 1]), c, s, false);
    }
    __syncthreads();
    if (col < d) out[(int64_t)t * o_sn + (int64_t)head * o_sh + col] = ZT<DT>::from_f32(r);
}

// valid_lens[b] = 1 + the last visible key of task b's LAST query row in a concatenated int8 visibility mask (task b: len_q rows of
// buf_lens[b] entries): what a prefix-visibility kernel needs where the caller only has the mask (the MLA search over the latent
// cache is handed DynBatchContext::s_mask, multi_head_latent_attention.cpp:1053-1069).  grid B, block 256.
__global__ __launch_bounds__(256) void k_mask_valid_lens(const int8_t* __restrict__ mask, const int32_t* __restrict__ buf_lens, int32_t* __restrict__ out,
                                                         int len_q) {
    __shared__ int red[256];
    const int b = blockIdx.x;
    int64_t off = 0;
    for (int i = 0; i < b; ++i) off += (int64_t)len_q * buf_lens[i];
    const int len_buf = buf_lens[b];
    const int8_t* row = mask + off + (int64_t)(len_q - 1) * len_buf;
    int last = 0;
    for (int j = threadIdx.x; j < len_buf; j += 256) if (row[j]) last = j + 1;
    red[threadIdx.x] = last;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[b] = red[0];
}

// Prompt-sized rope_qk_cache: grid S, block 256; a thread rotates 8 consecutive head-dim elements per trip (16-byte
// accesses; the element-per-thread kernel above spent 15.8 us on a 1024-token chunk, 2-byte accesses).  Same fp32
// expression per element (rope_val), so the outputs are bit-identical.
template <int DT>
__global__ __launch_bounds__(256) void k_rope_qk_vec(const float* __restrict__ cosv, const float* __restrict__ sinv,
                                                     const uint16_t* __restrict__ in, uint16_t* __restrict__ q,
                                                     uint16_t* __restrict__ k, uint16_t* __restrict__ v, int h, int hkv, int d,
                                                     int neox) {
    const int t = blockIdx.x, all = h + 2 * hkv, half = d / 2, cpr = d / 8;
    const uint16_t* row = in + (size_t)t * all * d;
    for (int c = threadIdx.x; c < all * cpr; c += 256) {
        const int head = c / cpr, d0 = (c % cpr) * 8;
        const uint16_t* src = row + (size_t)head * d;
        const uint4 a = *reinterpret_cast<const uint4*>(src + d0);
        if (head >= h + hkv) {
            *reinterpret_cast<uint4*>(v + ((size_t)t * hkv + (head - h - hkv)) * d + d0) = a;
            continue;
        }
        const int pd0 = neox ? (d0 < half ? d0 + half : d0 - half) : d0;
        const uint4 b = *reinterpret_cast<const uint4*>(src + pd0);
        const uint32_t au[4] = {a.x, a.y, a.z, a.w}, bu[4] = {b.x, b.y, b.z, b.w};
        float af[8], bf[8], cs[8], sn[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            af[2 * e] = ZT<DT>::to_f32((uint16_t)(au[e] & 0xffff));
            af[2 * e + 1] = ZT<DT>::to_f32((uint16_t)(au[e] >> 16));
            bf[2 * e] = ZT<DT>::to_f32((uint16_t)(bu[e] & 0xffff));
            bf[2 * e + 1] = ZT<DT>::to_f32((uint16_t)(bu[e] >> 16));
        }
        const float4* cp = reinterpret_cast<const float4*>(cosv + (size_t)t * d + d0);
        const float4* sp = reinterpret_cast<const float4*>(sinv + (size_t)t * d + d0);
        const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
        cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
        sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float r[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = e + u;
                if (neox) r[u] = rope_val(af[i], bf[i], cs[i], sn[i], d0 < half);
                else r[u] = rope_val(af[i], af[i ^ 1], cs[i], sn[i], (i & 1) == 0);
            }
            o[e / 2] = (uint32_t)ZT<DT>::from_f32(r[0]) | ((uint32_t)ZT<DT>::from_f32(r[1]) << 16);
        }
        uint16_t* dst = head >= h ? k + ((size_t)t * hkv + (head - h)) * d : q + ((size_t)t * h + head) * d;
        *reinterpret_cast<uint4*>(dst + d0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// KV scatter.  src/kvcache/ragged_buffer_kernel.cu:194-222.  grid (B, len_q, Hkv), block D/8 (16 B lanes)
// ------------------------------------------------------------------------------------------------
__global__ void k_copy_to_rag_buffer2(const int32_t* __restrict__ placement, const int32_t* __restrict__ buf_lens,
                                      const uint16_t* __restrict__ k_src, const uint16_t* __restrict__ v_src,
                                      uint16_t* const* __restrict__ k_bufs, uint16_t* const* __restrict__ v_bufs,
                                      int d, int bshd) {
    const int b = blockIdx.x, len_q = gridDim.y, hkv = gridDim.z, head = blockIdx.z;
    const int xi = b * len_q + blockIdx.y;
    const int p = placement[xi];
    const int64_t len_buf = buf_lens[b];
    // a slot outside the task's buffer is never written (the reference asserts pos_buf < len_buf,
    // ragged_buffer_kernel.cu:194-222); negative = padded row
    if (p < 0 || p >= len_buf) return;
    const size_t so = ((size_t)xi * hkv + head) * d;
    const size_t dof = bshd ? ((size_t)p * hkv + head) * d : ((size_t)head * len_buf + p) * d;
    for (int i = threadIdx.x * 8; i < d; i += blockDim.x * 8) {
        *reinterpret_cast<uint4*>(k_bufs[b] + dof + i) = *reinterpret_cast<const uint4*>(k_src + so + i);
        *reinterpret_cast<uint4*>(v_bufs[b] + dof + i) = *reinterpret_cast<const uint4*>(v_src + so + i);
    }
}

// the same scatter for rows of ANY element type, sizes in bytes (row_bytes % 4 == 0): the INT8 KV cache of the reference writes its
// u8 codes (row = dim_head bytes) and its fp32 scales (row = 4 bytes, is_scale = true) through copy_to_rag_buffer2 as well
// (attention.cpp:663-669, ragged_buffer_kernel.cu:254-300).  grid (B, len_q, Hkv), block 64
__global__ void k_copy_to_rag_buffer_bytes(const int32_t* __restrict__ placement, const int32_t* __restrict__ buf_lens, const unsigned char* __restrict__ k_src,
                                           const unsigned char* __restrict__ v_src, unsigned char* const* __restrict__ k_bufs,
                                           unsigned char* const* __restrict__ v_bufs, int row_bytes, int bshd) {
    const int b = blockIdx.x, len_q = gridDim.y, hkv = gridDim.z, head = blockIdx.z;
    const int xi = b * len_q + blockIdx.y;
    const int p = placement[xi];
    const int64_t len_buf = buf_lens[b];
    if (p < 0 || p >= len_buf) return;
    const size_t so = ((size_t)xi * hkv + head) * row_bytes;
    const size_t dof = (bshd ? ((size_t)p * hkv + head) : ((size_t)head * len_buf + p)) * row_bytes;
    for (int i = threadIdx.x * 4; i < row_bytes; i += blockDim.x * 4) {
        *reinterpret_cast<uint32_t*>(k_bufs[b] + dof + i) = *reinterpret_cast<const uint32_t*>(k_src + so + i);
        *reinterpret_cast<uint32_t*>(v_bufs[b] + dof + i) = *reinterpret_cast<const uint32_t*>(v_src + so + i);
    }
}

// fused decode front end: grid (B, H + 2Hkv), block D; q rotated -> q out, k rotated -> cache, v -> cache
template <int DT>
__global__ void k_rope_scatter_decode(const float* __restrict__ cosv, const float* __restrict__ sinv,
                                      const uint16_t* __restrict__ qkv, uint16_t* __restrict__ q,
                                      const int32_t* __restrict__ placement, const int32_t* __restrict__ buf_lens,
                                      uint16_t* const* __restrict__ k_bufs, uint16_t* const* __restrict__ v_bufs, int h,
                                      int hkv, int d, int neox, int bshd) {
    const int t = blockIdx.x, head = blockIdx.y, col = threadIdx.x, all = h + 2 * hkv, half = d / 2;
    if (col >= d) return;
    const uint16_t* src = qkv + ((size_t)t * all + head) * d;
    const int64_t len_buf = buf_lens[t];
    const int p = placement[t] < len_buf ? placement[t] : -1;    // outside the buffer: dropped like a padded row
    if (head >= h + hkv) {
        if (p < 0) return;
        const int hk = head - h - hkv;
        const size_t dof = bshd ? ((size_t)p * hkv + hk) * d : ((size_t)hk * len_buf + p) * d;
        v_bufs[t][dof + col] = src[col];
        return;
    }
    const float c = cosv[(size_t)t * d + col], s = sinv[(size_t)t * d + col];
    const float a = ZT<DT>::to_f32(src[col]);
    float r;
    if (neox)
        r = col < half ? rope_val(a, ZT<DT>::to_f32(src[col + half]), c, s, true)
                       : rope_val(a, ZT<DT>::to_f32(src[col - half]), c, s, false);
    else
        r = (col & 1) == 0 ? rope_val(a, ZT<DT>::to_f32(src[col + 1]), c, s, true)
                           : rope_val(a, ZT<DT>::to_f32(src[col - 1]), c, s, false);
    const uint16_t o = ZT<DT>::from_f32(r);
    if (head < h) {
        q[((size_t)t * h + head) * d + col] = o;
    } else if (p >= 0) {
        const int hk = head - h;
        const size_t dof = bshd ? ((size_t)p * hkv + hk) * d : ((size_t)hk * len_buf + p) * d;
        k_bufs[t][dof + col] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// element-wise.  src/nn/block/block_kernel.cu:8-17, src/nn/linear/activation_kernel.cu:59-80
// ------------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ uint16_t t_add(uint16_t a, uint16_t b) {  // one rounding to T (exact sum in fp32*)
    // fp16: the fp32 sum of two halfs is exact; bf16: may round in fp32 first (documented)
    return ZT<DT>::from_f32(ZT<DT>::to_f32(a) + ZT<DT>::to_f32(b));
}
template <int DT>
__device__ __forceinline__ uint16_t t_mul(uint16_t a, uint16_t b) {  // product of two T is exact in fp32
    return ZT<DT>::from_f32(ZT<DT>::to_f32(a) * ZT<DT>::to_f32(b));
}

template <int DT>
__global__ void k_add_scale(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ c,
                            int64_t n, uint16_t scale_t, int scale_residual) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        c[i] = scale_residual ? t_mul<DT>(t_add<DT>(a[i], b[i]), scale_t) : t_add<DT>(a[i], t_mul<DT>(b[i], scale_t));
}

template <int DT>
__global__ void k_gate_mul(const uint16_t* __restrict__ g, const uint16_t* __restrict__ u, uint16_t* __restrict__ out,
                           int64_t n, int act) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = ZT<DT>::to_f32(g[i]);
        float a;
        if (act == 0) a = x / (1.0f + expf(-x));
        else a = 0.5f * x * (1.0f + tanhf(0.7978845608028654f * x * (1.0f + 0.044715f * x * x)));
        out[i] = ZT<DT>::from_f32(a * ZT<DT>::to_f32(u[i]));
    }
}

// nn::gate_fuse (src/nn/linear/ff_kernel.cu:33-90): out[r, j] = act(in[r, j]) * in[r, ff + j] over a (rows, 2 ff) product of the fused
// w_in | w_gated linear (CPM_FUSE_FF_IN=1) -- k_gate_mul's arithmetic on the two halves of a row, without copying them out first
template <int DT>
__global__ void k_gate_fuse(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int64_t rows, int64_t ff, int act) {
    const int64_t n = rows * ff;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ff, j = i - r * ff;
        const float x = ZT<DT>::to_f32(in[r * 2 * ff + j]);
        float a;
        if (act == 0) a = x / (1.0f + expf(-x));
        else a = 0.5f * x * (1.0f + tanhf(0.7978845608028654f * x * (1.0f + 0.044715f * x * x)));
        out[i] = ZT<DT>::from_f32(a * ZT<DT>::to_f32(in[r * 2 * ff + ff + j]));
    }
}

// embedding: grid (S), block 256, 16-byte lanes.  src/nn/embedding/embedding.cu:23-44
// (16-byte lanes for real: with one 2-byte element per thread and iteration the 16 iterations of a 4096-wide row were 16
//  dependent round trips, 8.7 us for one token inside the decode step)
template <int DT>
__device__ __forceinline__ void embedding_row(int row, const int32_t* __restrict__ ids, const uint16_t* __restrict__ w,
                                              uint16_t* __restrict__ out, int dim, int begin, int end, float scale) {
    int id = ids[row];
    const bool in_range = id >= begin && id < end;
    id -= begin;
    const uint16_t* src = w + (size_t)(in_range ? id : 0) * dim;
    uint16_t* dst = out + (size_t)row * dim;
    if ((dim & 7) == 0 && ((((uintptr_t)w) | ((uintptr_t)out)) & 15) == 0) {
        const int chunks = dim >> 3;
        for (int c0 = threadIdx.x; c0 < chunks; c0 += 2 * blockDim.x) {
            const int c1 = c0 + blockDim.x;
            uint4 v0 = *reinterpret_cast<const uint4*>(src + (size_t)c0 * 8), v1 = make_uint4(0, 0, 0, 0);
            if (c1 < chunks) v1 = *reinterpret_cast<const uint4*>(src + (size_t)c1 * 8);
            auto conv = [&](uint32_t u) {
                const uint16_t lo = in_range ? ZT<DT>::from_f32(ZT<DT>::to_f32((uint16_t)(u & 0xffffu)) * scale) : ZT<DT>::from_f32(0.f);
                const uint16_t hi = in_range ? ZT<DT>::from_f32(ZT<DT>::to_f32((uint16_t)(u >> 16)) * scale) : ZT<DT>::from_f32(0.f);
                return (uint32_t)lo | ((uint32_t)hi << 16);
            };
            *reinterpret_cast<uint4*>(dst + (size_t)c0 * 8) = make_uint4(conv(v0.x), conv(v0.y), conv(v0.z), conv(v0.w));
            if (c1 < chunks) *reinterpret_cast<uint4*>(dst + (size_t)c1 * 8) = make_uint4(conv(v1.x), conv(v1.y), conv(v1.z), conv(v1.w));
        }
        return;
    }
    for (int i = threadIdx.x; i < dim; i += blockDim.x)
        dst[i] = in_range ? ZT<DT>::from_f32(ZT<DT>::to_f32(src[i]) * scale) : ZT<DT>::from_f32(0.f);
}
template <int DT>
__global__ void k_embedding(const int32_t* __restrict__ ids, const uint16_t* __restrict__ w, uint16_t* __restrict__ out,
                            int dim, int begin, int end, float scale) {
    embedding_row<DT>(blockIdx.x, ids, w, out, dim, begin, end, scale);
}
// the two independent launches a decode step starts with, as one: workgroups [0, S) gather the embedding rows, [S, 2 S) fill the
// rotary tables (2 of the ~4 us each costs are the kernel boundary)
template <int DT>
__global__ __launch_bounds__(256) void k_embedding_rope(const int32_t* __restrict__ ids, const uint16_t* __restrict__ w, uint16_t* __restrict__ out,
                                                        int dim, int begin, int end, float scale, int s_len, const int32_t* __restrict__ pos,
                                                        float* __restrict__ cosv, float* __restrict__ sinv, int d, float base, int neox,
                                                        int llama3, float factor, float low_ff, float high_ff, float old_ctx) {
    if ((int)blockIdx.x < s_len) embedding_row<DT>(blockIdx.x, ids, w, out, dim, begin, end, scale);
    else rope_cos_sin_row(blockIdx.x - s_len, pos, cosv, sinv, d, base, neox, llama3, factor, low_ff, high_ff, old_ctx);
}

inline int grid_1d(int64_t n, int threads) {
    int64_t g = (n + threads - 1) / threads;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

#define ZL_DT_SWITCH(dtype, EXPR_F16, EXPR_BF16) \
    if ((dtype) == ZL_F16) { EXPR_F16; } else if ((dtype) == ZL_BF16) { EXPR_BF16; } else return ZL_EDTYPE;

extern "C" {

int zl_version(void) { return ZL_VERSION; }

const char* zl_status_string(int st) {
    switch (st) {
        case ZL_OK: return "ok";
        case ZL_EINVAL: return "invalid argument (null pointer or non-positive size)";
        case ZL_ESHAPE: return "shape/alignment not supported by this kernel";
        case ZL_EDTYPE: return "dtype not supported on this path";
        case ZL_ELIMIT: return "exceeds a hardware limit (LDS / grid)";
        default: return st > 0 ? hipGetErrorString((hipError_t)st) : "unknown status";
    }
}

int zl_device_cu_count(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return -(int)e;
    static int cache[64] = {0};
    if (dev >= 0 && dev < 64 && cache[dev] > 0) return cache[dev];
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return -(int)e;
    if (dev >= 0 && dev < 64) cache[dev] = cus;  // benign race: same value from every thread
    return cus;
}

int zl_rmsnorm(const uint16_t* x, const uint16_t* weight, uint16_t* out, int64_t rows, int64_t dim, float eps,
               float scale, const uint16_t* x2, uint16_t* out_sum, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && weight && out && rows > 0 && dim > 0, ZL_EINVAL);
    ZL_CHECK_ARG(dim % 8 == 0, ZL_ESHAPE);
    size_t lds = (size_t)dim * 4 + 64;
    ZL_CHECK_ARG(lds <= 64 * 1024, ZL_ELIMIT);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_rmsnorm<ZL_F16>, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)s, x, weight, out, (int)dim, eps, scale, x2, out_sum),
        hipLaunchKernelGGL(k_rmsnorm<ZL_BF16>, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)s, x, weight, out, (int)dim, eps, scale, x2, out_sum))
    return zl_launch_status();
}

int zl_rope_cos_sin(const int32_t* pos, float* cosv, float* sinv, int64_t s_len, int64_t d, float base, int neox,
                    zl_stream_t s) {
    ZL_CHECK_ARG(pos && cosv && sinv && s_len > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 1024 && d % 2 == 0, ZL_ESHAPE);
    hipLaunchKernelGGL(k_rope_cos_sin, dim3((unsigned)s_len), dim3((unsigned)d), 0, (hipStream_t)s, pos, cosv, sinv,
                       (int)d, base, neox, 0, 1.f, 1.f, 1.f, 1.f);
    return zl_launch_status();
}

int zl_rope_cos_sin_llama3(const int32_t* pos, float* cosv, float* sinv, int64_t s_len, int64_t d, float base,
                           float factor, float low_freq_factor, float high_freq_factor, float old_context_len,
                           int neox, zl_stream_t s) {
    ZL_CHECK_ARG(pos && cosv && sinv && s_len > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 1024 && d % 2 == 0, ZL_ESHAPE);
    hipLaunchKernelGGL(k_rope_cos_sin, dim3((unsigned)s_len), dim3((unsigned)d), 0, (hipStream_t)s, pos, cosv, sinv,
                       (int)d, base, neox, 1, factor, low_freq_factor, high_freq_factor, old_context_len);
    return zl_launch_status();
}

int zl_rope_cos_sin_dynamic(const int32_t* pos, const int32_t* seq_len, float* cosv, float* sinv, int64_t s_len, int64_t d, float base,
                            float factor, float max_position_embeddings, int neox, zl_stream_t s) {
    ZL_CHECK_ARG(pos && cosv && sinv && s_len > 0 && d > 2 && max_position_embeddings > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 1024 && d % 2 == 0, ZL_ESHAPE);
    hipLaunchKernelGGL(k_rope_cos_sin_scaled, dim3((unsigned)s_len), dim3((unsigned)d), 0, (hipStream_t)s, pos, seq_len, cosv, sinv,
                       (int)d, base, neox, 2, factor, max_position_embeddings, 0.f, 0.f);
    return zl_launch_status();
}

int zl_rope_cos_sin_yarn(const int32_t* pos, float* cosv, float* sinv, int64_t s_len, int64_t d, float base, float factor, float low,
                         float high, float mscale, int neox, zl_stream_t s) {
    ZL_CHECK_ARG(pos && cosv && sinv && s_len > 0 && d > 0 && factor > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 1024 && d % 2 == 0, ZL_ESHAPE);
    hipLaunchKernelGGL(k_rope_cos_sin_scaled, dim3((unsigned)s_len), dim3((unsigned)d), 0, (hipStream_t)s, pos, nullptr, cosv, sinv,
                       (int)d, base, neox, 3, factor, low, high, mscale);
    return zl_launch_status();
}

int zl_head_norm(const uint16_t* x, const uint16_t* weight, uint16_t* out, int64_t rows, int64_t heads, int64_t d, int64_t ld_in,
                 int64_t ld_out, float eps, int mode, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && weight && out && rows > 0 && heads > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 1024 && heads <= 65535 && ld_in >= heads * d && ld_out >= heads * d && (mode == 0 || mode == 1), ZL_ESHAPE);
    const dim3 grid((unsigned)rows, (unsigned)heads), block((unsigned)((d + 63) / 64 * 64));
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_head_norm<ZL_F16>, grid, block, 0, (hipStream_t)s, x, weight, out, (int)d, ld_in, ld_out, eps, mode),
        hipLaunchKernelGGL(k_head_norm<ZL_BF16>, grid, block, 0, (hipStream_t)s, x, weight, out, (int)d, ld_in, ld_out, eps, mode))
    return zl_launch_status();
}

int zl_rotary_embedding_qk(const int32_t* pos, const uint16_t* in, uint16_t* q, uint16_t* k, uint16_t* v,
                           int64_t s_len, int64_t h, int64_t hkv, int64_t d, float theta, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(pos && in && q && k && v && s_len > 0 && h > 0 && hkv > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 1024 && d % 2 == 0 && h + 2 * hkv <= 65535, ZL_ESHAPE);
    dim3 grid((unsigned)s_len, (unsigned)(h + 2 * hkv));
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL((k_rope_qk<ZL_F16, 0>), grid, dim3((unsigned)d), 0, (hipStream_t)s, pos, nullptr, nullptr, in, q, k, v, (int)h, (int)hkv, (int)d, theta, 1),
        hipLaunchKernelGGL((k_rope_qk<ZL_BF16, 0>), grid, dim3((unsigned)d), 0, (hipStream_t)s, pos, nullptr, nullptr, in, q, k, v, (int)h, (int)hkv, (int)d, theta, 1))
    return zl_launch_status();
}

int zl_rope_qk_cache(const float* cosv, const float* sinv, const uint16_t* in, uint16_t* q, uint16_t* k, uint16_t* v,
                     int64_t s_len, int64_t h, int64_t hkv, int64_t d, int neox, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(cosv && sinv && in && q && k && v && s_len > 0 && h > 0 && hkv > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 1024 && d % 2 == 0 && h + 2 * hkv <= 65535, ZL_ESHAPE);
    if (s_len >= 16 && d % 16 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 &&
        ((uintptr_t)v & 15) == 0 && ((uintptr_t)cosv & 15) == 0 && ((uintptr_t)sinv & 15) == 0) {   // prompt chunks
        ZL_DT_SWITCH(dtype,
            hipLaunchKernelGGL(k_rope_qk_vec<ZL_F16>, dim3((unsigned)s_len), dim3(256), 0, (hipStream_t)s, cosv, sinv, in, q, k, v, (int)h, (int)hkv, (int)d, neox),
            hipLaunchKernelGGL(k_rope_qk_vec<ZL_BF16>, dim3((unsigned)s_len), dim3(256), 0, (hipStream_t)s, cosv, sinv, in, q, k, v, (int)h, (int)hkv, (int)d, neox))
        return zl_launch_status();
    }
    dim3 grid((unsigned)s_len, (unsigned)(h + 2 * hkv));
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL((k_rope_qk<ZL_F16, 1>), grid, dim3((unsigned)d), 0, (hipStream_t)s, nullptr, cosv, sinv, in, q, k, v, (int)h, (int)hkv, (int)d, 0.f, neox),
        hipLaunchKernelGGL((k_rope_qk<ZL_BF16, 1>), grid, dim3((unsigned)d), 0, (hipStream_t)s, nullptr, cosv, sinv, in, q, k, v, (int)h, (int)hkv, (int)d, 0.f, neox))
    return zl_launch_status();
}

int zl_rope_rotate(const float* cosv, const float* sinv, const uint16_t* x, uint16_t* out, int64_t n, int64_t heads, int64_t d, int64_t x_stride_n,
                   int64_t x_stride_h, int64_t out_stride_n, int64_t out_stride_h, int neox, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(cosv && sinv && x && out && n > 0 && heads > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 1024 && d % 2 == 0 && heads <= 65535 && x_stride_h >= d && out_stride_h >= d, ZL_ESHAPE);
    dim3 grid((unsigned)n, (unsigned)heads);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_rope_heads<ZL_F16>, grid, dim3((unsigned)d), 0, (hipStream_t)s, cosv, sinv, x, out, (int)d, x_stride_n, x_stride_h, out_stride_n, out_stride_h, neox),
        hipLaunchKernelGGL(k_rope_heads<ZL_BF16>, grid, dim3((unsigned)d), 0, (hipStream_t)s, cosv, sinv, x, out, (int)d, x_stride_n, x_stride_h, out_stride_n, out_stride_h, neox))
    return zl_launch_status();
}

int zl_mask_valid_lens(const int8_t* mask, const int32_t* buf_lens, int32_t* valid_lens, int64_t b, int64_t len_q, zl_stream_t s) {
    ZL_CHECK_ARG(mask && buf_lens && valid_lens && b > 0 && len_q > 0, ZL_EINVAL);
    hipLaunchKernelGGL(k_mask_valid_lens, dim3((unsigned)b), dim3(256), 0, (hipStream_t)s, mask, buf_lens, valid_lens, (int)len_q);
    return zl_launch_status();
}

int zl_copy_to_rag_buffer2(const int32_t* placement, const int32_t* buf_lens, const uint16_t* k_src,
                           const uint16_t* v_src, uint16_t* const* k_bufs, uint16_t* const* v_bufs, int64_t b,
                           int64_t len_q, int64_t hkv, int64_t d, int bshd, zl_stream_t s) {
    ZL_CHECK_ARG(placement && buf_lens && k_src && v_src && k_bufs && v_bufs && b > 0 && len_q > 0 && hkv > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d % 8 == 0 && len_q <= 65535 && hkv <= 65535, ZL_ESHAPE);
    dim3 grid((unsigned)b, (unsigned)len_q, (unsigned)hkv);
    hipLaunchKernelGGL(k_copy_to_rag_buffer2, grid, dim3(64), 0, (hipStream_t)s, placement, buf_lens, k_src, v_src,
                       k_bufs, v_bufs, (int)d, bshd);
    return zl_launch_status();
}

int zl_copy_to_rag_buffer_bytes(const int32_t* placement, const int32_t* buf_lens, const void* k_src, const void* v_src, void* const* k_bufs,
                                void* const* v_bufs, int64_t b, int64_t len_q, int64_t hkv, int64_t row_bytes, int bshd, zl_stream_t s) {
    ZL_CHECK_ARG(placement && buf_lens && k_src && v_src && k_bufs && v_bufs && b > 0 && len_q > 0 && hkv > 0 && row_bytes > 0, ZL_EINVAL);
    ZL_CHECK_ARG(row_bytes % 4 == 0 && len_q <= 65535 && hkv <= 65535 && row_bytes < ((int64_t)1 << 30), ZL_ESHAPE);
    dim3 grid((unsigned)b, (unsigned)len_q, (unsigned)hkv);
    hipLaunchKernelGGL(k_copy_to_rag_buffer_bytes, grid, dim3(64), 0, (hipStream_t)s, placement, buf_lens, (const unsigned char*)k_src, (const unsigned char*)v_src,
                       (unsigned char* const*)k_bufs, (unsigned char* const*)v_bufs, (int)row_bytes, bshd);
    return zl_launch_status();
}

int zl_rope_scatter_decode(const float* cosv, const float* sinv, const uint16_t* qkv, uint16_t* q,
                           const int32_t* placement, const int32_t* buf_lens, uint16_t* const* k_bufs,
                           uint16_t* const* v_bufs, int64_t b, int64_t h, int64_t hkv, int64_t d, int neox, int bshd,
                           int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(cosv && sinv && qkv && q && placement && buf_lens && k_bufs && v_bufs && b > 0 && h > 0 && hkv > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 1024 && d % 2 == 0 && h + 2 * hkv <= 65535, ZL_ESHAPE);
    dim3 grid((unsigned)b, (unsigned)(h + 2 * hkv));
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_rope_scatter_decode<ZL_F16>, grid, dim3((unsigned)d), 0, (hipStream_t)s, cosv, sinv, qkv, q, placement, buf_lens, k_bufs, v_bufs, (int)h, (int)hkv, (int)d, neox, bshd),
        hipLaunchKernelGGL(k_rope_scatter_decode<ZL_BF16>, grid, dim3((unsigned)d), 0, (hipStream_t)s, cosv, sinv, qkv, q, placement, buf_lens, k_bufs, v_bufs, (int)h, (int)hkv, (int)d, neox, bshd))
    return zl_launch_status();
}

// out[r, i] = x[r, perm[i]] (16-bit elements): nn::gptq::permute_input (gptq.h:155-159), the activation gather in front of
// an act-order (desc_act) linear whose rows were regrouped at load
__global__ void k_permute_input(const uint16_t* __restrict__ x, const int32_t* __restrict__ perm, uint16_t* __restrict__ out,
                                int64_t ldx, int64_t k) {
    const int64_t r = blockIdx.y;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < k; i += (int64_t)gridDim.x * blockDim.x)
        out[r * k + i] = x[r * ldx + perm[i]];
}

int zl_permute_input(const uint16_t* x, int64_t ldx, const int32_t* perm, uint16_t* out, int64_t rows, int64_t k, zl_stream_t s) {
    ZL_CHECK_ARG(x && perm && out && rows > 0 && k > 0 && ldx >= 1, ZL_EINVAL);
    ZL_CHECK_ARG(rows <= 65535, ZL_ELIMIT);
    hipLaunchKernelGGL(k_permute_input, dim3(grid_1d(k, 256), (unsigned)rows), dim3(256), 0, (hipStream_t)s, x, perm, out, ldx, k);
    return zl_launch_status();
}

int zl_element_add_scale(const uint16_t* a, const uint16_t* b, uint16_t* c, int64_t n, float scale, int scale_residual,
                         int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(a && b && c && n > 0, ZL_EINVAL);
    if (dtype == ZL_F16) {
        uint16_t st = __builtin_bit_cast(uint16_t, (_Float16)scale);
        hipLaunchKernelGGL(k_add_scale<ZL_F16>, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)s, a, b, c, n, st, scale_residual);
    } else if (dtype == ZL_BF16) {
        uint32_t u = __builtin_bit_cast(uint32_t, scale);
        u += 0x7fffu + ((u >> 16) & 1u);
        hipLaunchKernelGGL(k_add_scale<ZL_BF16>, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)s, a, b, c, n, (uint16_t)(u >> 16), scale_residual);
    } else
        return ZL_EDTYPE;
    return zl_launch_status();
}

int zl_gate_fuse(const uint16_t* in, uint16_t* out, int64_t rows, int64_t ff, int act, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(in && out && rows > 0 && ff > 0 && (act == 0 || act == 1), ZL_EINVAL);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_gate_fuse<ZL_F16>, dim3(grid_1d(rows * ff, 256)), dim3(256), 0, (hipStream_t)s, in, out, rows, ff, act),
        hipLaunchKernelGGL(k_gate_fuse<ZL_BF16>, dim3(grid_1d(rows * ff, 256)), dim3(256), 0, (hipStream_t)s, in, out, rows, ff, act))
    return zl_launch_status();
}

int zl_gate_mul(const uint16_t* gate, const uint16_t* up, uint16_t* out, int64_t n, int act, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(gate && up && out && n > 0 && (act == 0 || act == 1), ZL_EINVAL);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_gate_mul<ZL_F16>, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)s, gate, up, out, n, act),
        hipLaunchKernelGGL(k_gate_mul<ZL_BF16>, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)s, gate, up, out, n, act))
    return zl_launch_status();
}

int zl_embedding_rope(const int32_t* ids, const uint16_t* weight, uint16_t* out, int64_t s_len, int64_t dim, int32_t begin, int32_t end,
                      float scale, int dtype, const int32_t* pos, float* cosv, float* sinv, int64_t d, float base, int neox, int llama3,
                      float factor, float low_freq_factor, float high_freq_factor, float old_context_len, zl_stream_t s) {
    ZL_CHECK_ARG(ids && weight && out && pos && cosv && sinv && s_len > 0 && dim > 0 && d > 0, ZL_EINVAL);
    ZL_CHECK_ARG(d <= 256 && d % 2 == 0 && 2 * s_len < ((int64_t)1 << 31), ZL_ESHAPE);
    const dim3 grid((unsigned)(2 * s_len)), block(256);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_embedding_rope<ZL_F16>, grid, block, 0, (hipStream_t)s, ids, weight, out, (int)dim, begin, end, scale, (int)s_len, pos,
                           cosv, sinv, (int)d, base, neox, llama3, factor, low_freq_factor, high_freq_factor, old_context_len),
        hipLaunchKernelGGL(k_embedding_rope<ZL_BF16>, grid, block, 0, (hipStream_t)s, ids, weight, out, (int)dim, begin, end, scale, (int)s_len, pos,
                           cosv, sinv, (int)d, base, neox, llama3, factor, low_freq_factor, high_freq_factor, old_context_len))
    return zl_launch_status();
}

int zl_embedding(const int32_t* ids, const uint16_t* weight, uint16_t* out, int64_t s_len, int64_t dim, int32_t begin,
                 int32_t end, float scale, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(ids && weight && out && s_len > 0 && dim > 0, ZL_EINVAL);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_embedding<ZL_F16>, dim3((unsigned)s_len), dim3(256), 0, (hipStream_t)s, ids, weight, out, (int)dim, begin, end, scale),
        hipLaunchKernelGGL(k_embedding<ZL_BF16>, dim3((unsigned)s_len), dim3(256), 0, (hipStream_t)s, ids, weight, out, (int)dim, begin, end, scale))
    return zl_launch_status();
}

}  // extern "C"

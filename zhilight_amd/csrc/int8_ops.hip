// int8_ops.hip -- W8A8 dynamic per-token path.  Integer outputs are bit-exact
// with the reference: same abs-max rule, round-half-even (v_rndne_f32), int32-exact MFMA GEMM.
//   quant_calc_scale            src/nn/quant/int8/quant_kernel.cu:15-47
//   fuse_layernorm_rms_quant    src/nn/quant/int8/quant_kernel.cu:106-151
//   int8 x int8 -> int32 GEMM   src/nn/linear/linear.cpp:557-635 (cuBLASLt IMMA there)
//   quant_scale_back            src/nn/quant/int8/quant_kernel.cu:231-246
//   quant_back_act_mul          src/nn/quant/int8/quant_kernel.cu:589-614
#include "zl_common.h"

namespace {

// one workgroup (256) per row.  VEC: 16-byte loads / 8-byte stores, 8 elements per thread and trip (k % 8 == 0, 16-byte
// aligned rows): a decode row is one latency chain, and the scalar form spent it on k / 256 dependent 2-byte trips
// (9 us for a 14336-wide row); the arithmetic per element is unchanged.
template <int DT, bool VEC>
__global__ __launch_bounds__(256) void k_quant_rows(const uint16_t* __restrict__ x, int8_t* __restrict__ q,
                                                    float* __restrict__ scale, int k) {
    __shared__ float red[16];
    const size_t off = (size_t)blockIdx.x * k;
    float amax = 0.f;
    if constexpr (VEC) {
        for (int i = threadIdx.x * 8; i < k; i += 256 * 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + off + i);
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                amax = fmaxf(amax, fabsf(ZT<DT>::to_f32((uint16_t)(u[e] & 0xffff))));
                amax = fmaxf(amax, fabsf(ZT<DT>::to_f32((uint16_t)(u[e] >> 16))));
            }
        }
    } else {
        for (int i = threadIdx.x; i < k; i += 256) amax = fmaxf(amax, fabsf(ZT<DT>::to_f32(x[off + i])));
    }
    amax = zl_block_max(amax, red);
    const float bs = 127.f / amax;
    if constexpr (VEC) {
        for (int i = threadIdx.x * 8; i < k; i += 256 * 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + off + i);   // L1/L2 hit
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
            uint32_t o[2] = {0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = (int)__builtin_rintf(ZT<DT>::to_f32((uint16_t)(u[e] & 0xffff)) * bs);
                const int b = (int)__builtin_rintf(ZT<DT>::to_f32((uint16_t)(u[e] >> 16)) * bs);
                o[e >> 1] |= ((uint32_t)(uint8_t)(int8_t)a | ((uint32_t)(uint8_t)(int8_t)b << 8)) << (16 * (e & 1));
            }
            *reinterpret_cast<uint2*>(q + off + i) = make_uint2(o[0], o[1]);
        }
    } else {
        for (int i = threadIdx.x; i < k; i += 256)
            q[off + i] = (int8_t)__builtin_rintf(ZT<DT>::to_f32(x[off + i]) * bs);
    }
    if (threadIdx.x == 0) scale[blockIdx.x] = amax / 127.f;
}

// VEC (dim % 8 == 0): 16-byte loads into two LDS rows (v and v*w); the sum of squares then runs over the LDS copy in the
// reference's own thread / tree order (below), while the row is fetched in one round trip.
template <int DT, bool VEC>
__global__ __launch_bounds__(256) void k_rmsnorm_quant(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                       uint16_t* __restrict__ out, int8_t* __restrict__ q,
                                                       float* __restrict__ out_scale, int dim, float eps,
                                                       float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* vw = reinterpret_cast<float*>(smem);
    float* vv = vw + dim;                         // VEC only
    float* red = vw + (VEC ? 2 * dim : dim);
    const size_t off = (size_t)blockIdx.x * dim;
    float ss = 0.f, amax = 0.f;
    if constexpr (VEC) {
        for (int i = threadIdx.x * 8; i < dim; i += 256 * 8) {
            const uint4 xa = *reinterpret_cast<const uint4*>(x + off + i);
            const uint4 wa = *reinterpret_cast<const uint4*>(w + i);
            const uint32_t xu[4] = {xa.x, xa.y, xa.z, xa.w}, wu[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = ZT<DT>::to_f32((uint16_t)(xu[e >> 1] >> (16 * (e & 1))));
                const float pw = v * ZT<DT>::to_f32((uint16_t)(wu[e >> 1] >> (16 * (e & 1))));
                vv[i + e] = v;
                vw[i + e] = pw;
                amax = fmaxf(amax, fabsf(pw));
            }
        }
        __syncthreads();
    } else {
        for (int i = threadIdx.x; i < dim; i += 256) {
            const float v = ZT<DT>::to_f32(x[off + i]);
            const float pw = v * ZT<DT>::to_f32(w[i]);
            vw[i] = pw;
            amax = fmaxf(amax, fabsf(pw));
        }
    }
    // Sum of squares in the REFERENCE's order (round 5; quant_kernel.cu:120-133 + reduce.cuh:92-107): its block has
    // T = min(round_up(dim, 32), 1024) threads, thread t chains fma over the elements t, t + T, ..., a 32-lane shuffle-down tree per
    // warp, then warp 0 trees the per-warp results.  The rows' rs -- and with it the fp32 activation scale amax * rs / 127 every
    // scale-back multiplies by -- then carry the reference's bits: with the 256-thread order used before, one row in four came out
    // one fp32 ulp away, which is where the INT8 route's first differing bit entered (tests/test_gpu_fullgeom.py::
    // test_int8_depth_record_and_first_differing_op).  Thread p plays the reference's threads p, p + 256, ...: the 32 lanes of a
    // reference warp are one half of a wavefront, the xor butterfly leaves lane 0 of each half with the shuffle-down tree's value.
    {
        const int vt = min((dim + 31) / 32 * 32, 1024);
        __shared__ float wres[32], tot;
        if (threadIdx.x < 32) wres[threadIdx.x] = 0.f;
        __syncthreads();
        for (int t0 = 0; t0 < vt; t0 += 256) {
            const int t = t0 + (int)threadIdx.x;
            float part = 0.f;
            if (t < vt) {
                for (int i = t; i < dim; i += vt) {
                    float v;
                    if constexpr (VEC) v = vv[i];
                    else v = ZT<DT>::to_f32(x[off + i]);
                    part = __builtin_fmaf(v, v, part);
                }
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            if ((threadIdx.x & 31) == 0 && t < vt) wres[t >> 5] = part;
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            float v = wres[threadIdx.x];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (threadIdx.x == 0) tot = v;
        }
        __syncthreads();
        ss = tot;
    }
    const float rs = zl_rsqrt_rn(ss / (float)dim + eps);
    amax = zl_block_max(amax, red);
    amax = ZT<DT>::to_f32(ZT<DT>::from_f32(amax));             // the reference reduces the max in T
    const float bs = (float)(127.0 / (double)amax);
    if constexpr (VEC) {
        for (int i = threadIdx.x * 8; i < dim; i += 256 * 8) {
            uint32_t o16[4] = {0, 0, 0, 0}, o8[2] = {0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = vw[i + e] / scale;
                o16[e >> 1] |= (uint32_t)ZT<DT>::from_f32(v * rs) << (16 * (e & 1));
                o8[e >> 2] |= (uint32_t)(uint8_t)(int8_t)__builtin_rintf(v * bs) << (8 * (e & 3));
            }
            *reinterpret_cast<uint4*>(out + off + i) = make_uint4(o16[0], o16[1], o16[2], o16[3]);
            *reinterpret_cast<uint2*>(q + off + i) = make_uint2(o8[0], o8[1]);
        }
    } else {
        for (int i = threadIdx.x; i < dim; i += 256) {
            const float v = vw[i] / scale;
            out[off + i] = ZT<DT>::from_f32(v * rs);
            q[off + i] = (int8_t)__builtin_rintf(v * bs);
        }
    }
    if (threadIdx.x == 0) out_scale[blockIdx.x] = (float)((double)(amax * rs) / 127.);
}

// ---- int8 GEMM: C[m,n] = sum_k A[m,k] * B[n,k], int32.  v_mfma_i32_16x16x64_i8.
// A fragment: lane (row = lane&15, kq = lane>>4) holds 16 consecutive k (4 VGPRs) of x row m0+row.
// B fragment: lane (col = lane&15, kq) holds 16 consecutive k of weight row n0+col (one dwordx4).
// C: lane holds rows 4*(lane>>4)+i (i=0..3) of column lane&15.
// One wave per 16 weight rows and MT*16 activation rows; K split over gridDim.y with exact int32
// atomics when splitk > 1 (integer addition is associative: still bit-exact and deterministic).
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MTILES>
__global__ __launch_bounds__(256) void k_int8_gemm(const int8_t* __restrict__ a, const int8_t* __restrict__ b,
                                                   int32_t* __restrict__ c, int m, int n, int k, int k_per_split,
                                                   int use_atomic) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= n) return;
    const int m0 = blockIdx.z * (MTILES * 16);
    const int kb = blockIdx.y * k_per_split, ke = min(k, kb + k_per_split);
    const int col = lane & 15, kq = lane >> 4;
    v4i acc[MTILES];
#pragma unroll
    for (int t = 0; t < MTILES; ++t) acc[t] = (v4i){0, 0, 0, 0};
    const bool ncol_ok = (n0 + col) < n;
    const int8_t* brow = b + (size_t)(n0 + col) * k;
    for (int k0 = kb; k0 < ke; k0 += 64) {
        const int kk = k0 + 16 * kq;
        v4i bf = (v4i){0, 0, 0, 0};
        if (ncol_ok && kk < ke) bf = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(brow + kk));
#pragma unroll
        for (int t = 0; t < MTILES; ++t) {
            const int row = m0 + t * 16 + col;  // A uses lane&15 as the row index
            v4i af = (v4i){0, 0, 0, 0};
            if (row < m && kk < ke) af = *reinterpret_cast<const v4i*>(a + (size_t)row * k + kk);
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < MTILES; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + t * 16 + 4 * kq + i;
            if (row < m && ncol_ok) {
                int32_t* dst = c + (size_t)row * n + n0 + col;
                if (use_atomic) atomicAdd(dst, acc[t][i]);
                else *dst = acc[t][i];
            }
        }
}

// Tiled int8 GEMM (the Int8Linear hot kernel, functions::Gemm int8 path of linear.cpp:557-635 / cuBLASLt IMMA):
// same structure as dense_gemm.hip -- workgroup = 4 waves = BM x 128 outputs, wave = BM x 32, K in 256-byte
// chunks, the activation chunk double-buffered in LDS, weight fragments (lane = row n, 16 consecutive k) loaded
// two chunks ahead straight into the MFMA B registers (v_mfma_i32_16x16x64_i8).  Few workgroups (decode
// batches) -> K split over blockIdx.z with int32 atomics: integer addition is associative, so the result is
// exact and order-independent.
constexpr int kI8Row = 256 + 16;    // padded LDS row (bytes)

template <int BM>
__global__ __launch_bounds__(256, 2) void k_int8_gemm_tiled(const int8_t* __restrict__ a, const int8_t* __restrict__ b,
                                                            int32_t* __restrict__ c, int m, int n, int k, int chunks,
                                                            int split_chunks, int use_atomic) {
    constexpr int RB = BM / 16, XR = BM / 16;
    __shared__ __attribute__((aligned(16))) int8_t xs[2][BM * kI8Row];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrow = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * BM;
    const int n_base = blockIdx.x * 128 + wave * 32;
    const int g_begin = blockIdx.z * split_chunks;
    const int G = min(chunks, g_begin + split_chunks);
    int nr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nn = n_base + 16 * j + nrow;
        nr[j] = nn < n ? nn : n - 1;
    }
    v4i wf[2][2][4];
    auto load_w = [&](int slot, int g) {
        const int gc = g < G ? g : G - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int8_t* src = b + (size_t)nr[j] * k + (size_t)gc * 256 + 16 * kq;
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[slot][j][t] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src + 64 * t));
        }
    };
    const int xrow = threadIdx.x >> 4, xcol = (threadIdx.x & 15) * 16;
    v4i xr[XR];
    auto load_x = [&](int g) {
        const int gc = g < G ? g : G - 1;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int row = m0 + xrow + 16 * r;
            const int rc = row < m ? row : m - 1;
            xr[r] = *reinterpret_cast<const v4i*>(a + (size_t)rc * k + (size_t)gc * 256 + xcol);
            if (row >= m) xr[r] = (v4i){0, 0, 0, 0};
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int r = 0; r < XR; ++r) *reinterpret_cast<v4i*>(&xs[buf][(xrow + 16 * r) * kI8Row + xcol]) = xr[r];
    };
    load_x(g_begin);
    load_w(0, g_begin);
    load_w(1, g_begin + 1);
    store_x(0);
    load_x(g_begin + 1);
    __syncthreads();
    v4i acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        acc[rb][0] = (v4i){0, 0, 0, 0};
        acc[rb][1] = (v4i){0, 0, 0, 0};
    }
    int g = g_begin;
#pragma unroll 1
    for (; g < G; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (g + u < G) {
                const int8_t* xb = &xs[u][nrow * kI8Row + kq * 16];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const v4i af = *reinterpret_cast<const v4i*>(xb + rb * 16 * kI8Row + t * 64);
                        acc[rb][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, wf[u][0][t], acc[rb][0], 0, 0, 0);
                        acc[rb][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, wf[u][1][t], acc[rb][1], 0, 0, 0);
                    }
                }
                load_w(u, g + u + 2);
                store_x(u ^ 1);
                load_x(g + u + 2);
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nn = n_base + 16 * j + nrow;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + rb * 16 + 4 * kq + i;
                if (row < m && nn < n) {
                    int32_t* dst = c + (size_t)row * n + nn;
                    if (use_atomic) atomicAdd(dst, acc[rb][j][i]);
                    else *dst = acc[rb][j][i];
                }
            }
        }
    }
}

template <int DT>
__global__ void k_scale_back(const int32_t* __restrict__ c, const float* __restrict__ sx,
                             const uint16_t* __restrict__ sy, uint16_t* __restrict__ out, int n) {
    const int r = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col < n) {
        const size_t pos = (size_t)r * n + col;
        out[pos] = ZT<DT>::from_f32((float)c[pos] * sx[r] * ZT<DT>::to_f32(sy[col]));
    }
}

// ---- W4A8 (int8 activations on W4 weights: the M > W4_A8_M_THRES branch of gptq_gemm_k_major, q_gemm_k_major.cu:1036-1073)
// load time: row n of the dequantised matrix W16 -> scale[n] = amax / 127 (fp32, Int4GPTQ::calc_w4a8_scale,
// linear.cpp:1101-1112) and w8 = int8(nearbyintf(float(w) * (1.f / scale))) (KERNEL_dequant<int8_t, 1>, :843-905)
__global__ __launch_bounds__(256) void k_w4a8_rows_to_int8(const uint16_t* __restrict__ w16, int8_t* __restrict__ w8,
                                                           float* __restrict__ scale, int64_t k) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const uint16_t* src = w16 + row * k;
    float amax = 0.f;
    for (int64_t i = threadIdx.x; i < k; i += blockDim.x) amax = fmaxf(amax, fabsf(ZT<ZL_F16>::to_f32(src[i])));
    amax = zl_block_max(amax, red);
    const float sc = amax / 127.f;
    if (threadIdx.x == 0) scale[row] = sc;
    const float r = 1.f / sc;
    for (int64_t i = threadIdx.x; i < k; i += blockDim.x) w8[row * k + i] = (int8_t)nearbyintf(ZT<ZL_F16>::to_f32(src[i]) * r);
}
// forward: y = half(float(acc) * sx[m] * sy[n]) with the fp32 per-row weight scale (quant_scale_back on a float scale_y)
__global__ void k_scale_back_f32(const int32_t* __restrict__ c, const float* __restrict__ sx, const float* __restrict__ sy,
                                 uint16_t* __restrict__ out, int n) {
    const int r = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col < n) {
        const size_t pos = (size_t)r * n + col;
        out[pos] = ZT<ZL_F16>::from_f32((float)c[pos] * sx[r] * sy[col]);
    }
}

template <int DT>
__global__ void k_back_act_mul(const int32_t* __restrict__ a, const float* __restrict__ asx,
                               const uint16_t* __restrict__ asy, const int32_t* __restrict__ b,
                               const float* __restrict__ bsx, const uint16_t* __restrict__ bsy,
                               uint16_t* __restrict__ out, int n, int act) {
    const int r = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col < n) {
        const size_t pos = (size_t)r * n + col;
        const float ab = (float)a[pos] * asx[r] * ZT<DT>::to_f32(asy[col]);
        const float bb = (float)b[pos] * bsx[r] * ZT<DT>::to_f32(bsy[col]);
        float gate;
        if (act == 0) gate = ab / (1.0f + expf(-ab));
        else gate = 0.5f * ab * (1.0f + tanhf(0.7978845608028654f * ab * (1.0f + 0.044715f * ab * ab)));
        out[pos] = ZT<DT>::from_f32(bb * gate);
    }
}

// The remaining scale-back flavours of the reference (quant_kernel.cu:311-567) are the same arithmetic
// T(float(int32) * sx[row] * sy[col]) with a different destination (and, for one, a fused residual add):
// one kernel, the destination computed per mode.
enum { kBack3 = 0, kBackAdd = 1, kBackTranspose = 2, kBackToBuffer = 3 };
struct BackParams {
    const int32_t* src;
    const float* sx;
    const uint16_t* sy;
    uint16_t *o0, *o1, *o2;
    const uint16_t* addend;
    const int32_t* placement;
    float scale;
    int64_t rows, n;                 // logical (row, col) view of src: row = (b, t), col = (head, e)
    int64_t dim_q, dim_kv;           // kBack3
    int64_t len_q, heads, d, len_buf, src_stride, dst_stride, place_stride;  // kBackTranspose / kBackToBuffer
};

template <int DT, int MODE>
__global__ void k_scale_back_ex(const BackParams p) {
    const int64_t r = blockIdx.y;
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= p.n) return;
    if constexpr (MODE == kBack3 || MODE == kBackAdd) {
        const int64_t pos = r * p.n + col;
        const float qb = (float)p.src[pos] * p.sx[r] * ZT<DT>::to_f32(p.sy[col]);
        if constexpr (MODE == kBackAdd) {
            p.o0[pos] = ZT<DT>::from_f32((qb + ZT<DT>::to_f32(p.addend[pos])) * p.scale);
        } else {
            const uint16_t v = ZT<DT>::from_f32(qb);
            if (col < p.dim_q) p.o0[r * p.dim_q + col] = v;
            else if (col < p.dim_q + p.dim_kv) p.o1[r * p.dim_kv + col - p.dim_q] = v;
            else p.o2[r * p.dim_kv + col - p.dim_q - p.dim_kv] = v;
        }
    } else {
        const int64_t b = r / p.len_q, t = r % p.len_q;   // len_q doubles as len_kv for the buffer scatter
        const int64_t h = col / p.d, e = col % p.d;
        const float x = p.sx[b * p.len_q + t], y = ZT<DT>::to_f32(p.sy[col]);
        if constexpr (MODE == kBackTranspose) {
            const float qb = (float)p.src[((b * p.len_q + t) * p.heads + h) * p.d + e] * x * y;
            p.o0[((b * p.heads + h) * p.len_q + t) * p.d + e] = ZT<DT>::from_f32(qb);
        } else {
            const int64_t pos_buf = p.placement ? p.placement[b * p.place_stride + t] : t;
            if (pos_buf < 0) return;   // padded row
            const float qb = (float)p.src[b * p.src_stride + (t * p.heads + h) * p.d + e] * x * y;
            p.o0[b * p.dst_stride + (h * p.len_buf + pos_buf) * p.d + e] = ZT<DT>::from_f32(qb);
        }
    }
}

template <int MODE>
int launch_back(const BackParams& p, int dtype, hipStream_t st) {
    if (p.rows > 65535) return ZL_ELIMIT;
    dim3 grid((unsigned)((p.n + 255) / 256), (unsigned)p.rows);
    if (dtype == ZL_F16) hipLaunchKernelGGL((k_scale_back_ex<ZL_F16, MODE>), grid, dim3(256), 0, st, p);
    else if (dtype == ZL_BF16) hipLaunchKernelGGL((k_scale_back_ex<ZL_BF16, MODE>), grid, dim3(256), 0, st, p);
    else return ZL_EDTYPE;
    return zl_launch_status();
}

}  // namespace

#define ZL_DT_SWITCH(dtype, EXPR_F16, EXPR_BF16) \
    if ((dtype) == ZL_F16) { EXPR_F16; } else if ((dtype) == ZL_BF16) { EXPR_BF16; } else return ZL_EDTYPE;

extern "C" {

int zl_quant_calc_scale(const uint16_t* x, int8_t* q, float* scale, int64_t m, int64_t k, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && q && scale && m > 0 && k > 0, ZL_EINVAL);
    const bool vec = k % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 7) == 0;
    if (vec) {
        ZL_DT_SWITCH(dtype,
            hipLaunchKernelGGL((k_quant_rows<ZL_F16, true>), dim3((unsigned)m), dim3(256), 0, (hipStream_t)s, x, q, scale, (int)k),
            hipLaunchKernelGGL((k_quant_rows<ZL_BF16, true>), dim3((unsigned)m), dim3(256), 0, (hipStream_t)s, x, q, scale, (int)k))
    } else {
        ZL_DT_SWITCH(dtype,
            hipLaunchKernelGGL((k_quant_rows<ZL_F16, false>), dim3((unsigned)m), dim3(256), 0, (hipStream_t)s, x, q, scale, (int)k),
            hipLaunchKernelGGL((k_quant_rows<ZL_BF16, false>), dim3((unsigned)m), dim3(256), 0, (hipStream_t)s, x, q, scale, (int)k))
    }
    return zl_launch_status();
}

int zl_rmsnorm_quant(const uint16_t* x, const uint16_t* weight, uint16_t* out, int8_t* q, float* out_scale,
                     int64_t rows, int64_t dim, float eps, float scale, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && weight && out && q && out_scale && rows > 0 && dim > 0, ZL_EINVAL);
    const bool vec = dim % 8 == 0 && (size_t)dim * 8 + 64 <= 64 * 1024 && ((uintptr_t)x & 15) == 0 &&
                     ((uintptr_t)weight & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)q & 7) == 0;
    size_t lds = (size_t)dim * (vec ? 8 : 4) + 64;
    ZL_CHECK_ARG(lds <= 64 * 1024, ZL_ELIMIT);
    if (vec) {
        ZL_DT_SWITCH(dtype,
            hipLaunchKernelGGL((k_rmsnorm_quant<ZL_F16, true>), dim3((unsigned)rows), dim3(256), lds, (hipStream_t)s, x, weight, out, q, out_scale, (int)dim, eps, scale),
            hipLaunchKernelGGL((k_rmsnorm_quant<ZL_BF16, true>), dim3((unsigned)rows), dim3(256), lds, (hipStream_t)s, x, weight, out, q, out_scale, (int)dim, eps, scale))
    } else {
        ZL_DT_SWITCH(dtype,
            hipLaunchKernelGGL((k_rmsnorm_quant<ZL_F16, false>), dim3((unsigned)rows), dim3(256), lds, (hipStream_t)s, x, weight, out, q, out_scale, (int)dim, eps, scale),
            hipLaunchKernelGGL((k_rmsnorm_quant<ZL_BF16, false>), dim3((unsigned)rows), dim3(256), lds, (hipStream_t)s, x, weight, out, q, out_scale, (int)dim, eps, scale))
    }
    return zl_launch_status();
}

int zl_int8_gemm_nt(const int8_t* a, const int8_t* b, int32_t* c, int64_t m, int64_t n, int64_t k, zl_stream_t s) {
    ZL_CHECK_ARG(a && b && c && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(k % 16 == 0, ZL_ESHAPE);
    hipStream_t hs0 = (hipStream_t)s;
    if (k % 256 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0) {
        const int bm = m <= 16 ? 16 : (m <= 32 ? 32 : 64);
        const int gxt = (int)((n + 127) / 128), gyt = (int)((m + bm - 1) / bm);
        ZL_CHECK_ARG(gyt <= 65535, ZL_ELIMIT);
        const int chunks = (int)(k / 256);
        int cus = zl_device_cu_count();
        if (cus <= 0) cus = 256;
        int splits = 1;
        if ((int64_t)gxt * gyt < cus) {
            splits = (int)((2 * (int64_t)cus + (int64_t)gxt * gyt - 1) / ((int64_t)gxt * gyt));
            const int max_s = chunks / 4 > 0 ? chunks / 4 : 1;
            if (splits > max_s) splits = max_s;
            if (splits > 32) splits = 32;
        }
        int sc = (chunks + splits - 1) / splits;
        splits = (chunks + sc - 1) / sc;
        if (splits > 1) {
            hipError_t e = hipMemsetAsync(c, 0, (size_t)m * n * 4, hs0);
            if (e != hipSuccess) return (int)e;
        }
        const dim3 grid((unsigned)gxt, (unsigned)gyt, (unsigned)splits);
        if (bm == 16) hipLaunchKernelGGL(k_int8_gemm_tiled<16>, grid, dim3(256), 0, hs0, a, b, c, (int)m, (int)n, (int)k, chunks, sc, splits > 1);
        else if (bm == 32) hipLaunchKernelGGL(k_int8_gemm_tiled<32>, grid, dim3(256), 0, hs0, a, b, c, (int)m, (int)n, (int)k, chunks, sc, splits > 1);
        else hipLaunchKernelGGL(k_int8_gemm_tiled<64>, grid, dim3(256), 0, hs0, a, b, c, (int)m, (int)n, (int)k, chunks, sc, splits > 1);
        return zl_launch_status();
    }
    const int mt = m > 48 ? 4 : (m > 32 ? 3 : (m > 16 ? 2 : 1));
    const int gz = (int)((m + mt * 16 - 1) / (mt * 16));
    const int gx = (int)((n + 63) / 64);
    // split K so that at least ~1024 waves stream the weights
    int splitk = 1;
    while ((int64_t)gx * 4 * splitk < 1024 && k / (splitk * 2) >= 512 && (k / (splitk * 2)) % 64 == 0) splitk *= 2;
    const int kps = (int)((k / splitk + 63) / 64 * 64);
    hipStream_t hs = (hipStream_t)s;
    if (splitk > 1) {
        hipError_t e = hipMemsetAsync(c, 0, (size_t)m * n * 4, hs);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)gx, (unsigned)splitk, (unsigned)gz);
    switch (mt) {
        case 1: hipLaunchKernelGGL(k_int8_gemm<1>, grid, dim3(256), 0, hs, a, b, c, (int)m, (int)n, (int)k, kps, splitk > 1); break;
        case 2: hipLaunchKernelGGL(k_int8_gemm<2>, grid, dim3(256), 0, hs, a, b, c, (int)m, (int)n, (int)k, kps, splitk > 1); break;
        case 3: hipLaunchKernelGGL(k_int8_gemm<3>, grid, dim3(256), 0, hs, a, b, c, (int)m, (int)n, (int)k, kps, splitk > 1); break;
        default: hipLaunchKernelGGL(k_int8_gemm<4>, grid, dim3(256), 0, hs, a, b, c, (int)m, (int)n, (int)k, kps, splitk > 1); break;
    }
    return zl_launch_status();
}

int zl_quant_scale_back(const int32_t* c, const float* scale_x, const uint16_t* scale_y, uint16_t* out, int64_t m,
                        int64_t n, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(c && scale_x && scale_y && out && m > 0 && n > 0, ZL_EINVAL);
    ZL_CHECK_ARG(m <= 65535, ZL_ELIMIT);
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)m);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_scale_back<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, c, scale_x, scale_y, out, (int)n),
        hipLaunchKernelGGL(k_scale_back<ZL_BF16>, grid, dim3(256), 0, (hipStream_t)s, c, scale_x, scale_y, out, (int)n))
    return zl_launch_status();
}

int zl_w4a8_weight_to_int8(const uint16_t* w16, int8_t* w8, float* scale, int64_t n, int64_t k, zl_stream_t s) {
    ZL_CHECK_ARG(w16 && w8 && scale && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(n <= 0x7fffffff, ZL_ELIMIT);
    hipLaunchKernelGGL(k_w4a8_rows_to_int8, dim3((unsigned)n), dim3(256), 0, (hipStream_t)s, w16, w8, scale, k);
    return zl_launch_status();
}

int zl_quant_scale_back_f32(const int32_t* c, const float* scale_x, const float* scale_y, uint16_t* out, int64_t m, int64_t n,
                            zl_stream_t s) {
    ZL_CHECK_ARG(c && scale_x && scale_y && out && m > 0 && n > 0, ZL_EINVAL);
    ZL_CHECK_ARG(m <= 65535, ZL_ELIMIT);
    hipLaunchKernelGGL(k_scale_back_f32, dim3((unsigned)((n + 255) / 256), (unsigned)m), dim3(256), 0, (hipStream_t)s, c, scale_x,
                       scale_y, out, (int)n);
    return zl_launch_status();
}

int zl_quant_back_act_mul(const int32_t* a, const float* a_sx, const uint16_t* a_sy, const int32_t* b,
                          const float* b_sx, const uint16_t* b_sy, uint16_t* out, int64_t m, int64_t n, int act,
                          int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(a && a_sx && a_sy && b && b_sx && b_sy && out && m > 0 && n > 0, ZL_EINVAL);
    ZL_CHECK_ARG(m <= 65535, ZL_ELIMIT);
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)m);
    ZL_DT_SWITCH(dtype,
        hipLaunchKernelGGL(k_back_act_mul<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, a, a_sx, a_sy, b, b_sx, b_sy, out, (int)n, act),
        hipLaunchKernelGGL(k_back_act_mul<ZL_BF16>, grid, dim3(256), 0, (hipStream_t)s, a, a_sx, a_sy, b, b_sx, b_sy, out, (int)n, act))
    return zl_launch_status();
}

int zl_quant_scale_back3(const int32_t* c, const float* scale_x, const uint16_t* scale_y, uint16_t* q, uint16_t* k,
                         uint16_t* v, int64_t m, int64_t n, int64_t dim_q, int64_t dim_kv, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(c && scale_x && scale_y && q && k && v && m > 0 && n > 0, ZL_EINVAL);
    ZL_CHECK_ARG(dim_q >= 0 && dim_kv >= 0 && dim_q + 2 * dim_kv == n, ZL_ESHAPE);
    BackParams p = {};
    p.src = c; p.sx = scale_x; p.sy = scale_y; p.o0 = q; p.o1 = k; p.o2 = v;
    p.rows = m; p.n = n; p.dim_q = dim_q; p.dim_kv = dim_kv;
    return launch_back<kBack3>(p, dtype, (hipStream_t)s);
}

int zl_quant_back_element_add_scale(const int32_t* a, const float* scale_x, const uint16_t* scale_y, const uint16_t* b,
                                    float scale, uint16_t* out, int64_t m, int64_t n, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(a && scale_x && scale_y && b && out && m > 0 && n > 0, ZL_EINVAL);
    BackParams p = {};
    p.src = a; p.sx = scale_x; p.sy = scale_y; p.o0 = out; p.addend = b; p.scale = scale;
    p.rows = m; p.n = n;
    return launch_back<kBackAdd>(p, dtype, (hipStream_t)s);
}

int zl_quant_back_transpose(const int32_t* inp, const float* scale_x, const uint16_t* scale_y, uint16_t* out,
                            int64_t batch, int64_t len_q, int64_t heads, int64_t dim_head, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(inp && scale_x && scale_y && out && batch > 0 && len_q > 0 && heads > 0 && dim_head > 0, ZL_EINVAL);
    BackParams p = {};
    p.src = inp; p.sx = scale_x; p.sy = scale_y; p.o0 = out;
    p.rows = batch * len_q; p.n = heads * dim_head; p.len_q = len_q; p.heads = heads; p.d = dim_head;
    return launch_back<kBackTranspose>(p, dtype, (hipStream_t)s);
}

int zl_quant_back_copy_to_buffer(const int32_t* src, const float* scale_x, const uint16_t* scale_y,
                                 const int32_t* placement, uint16_t* dst, int64_t batch, int64_t len_kv, int64_t heads,
                                 int64_t dim_head, int64_t len_buf, int64_t src_stride, int64_t dst_stride,
                                 int64_t place_stride, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(src && scale_x && scale_y && dst && batch > 0 && len_kv > 0 && heads > 0 && dim_head > 0 && len_buf > 0,
                 ZL_EINVAL);
    BackParams p = {};
    p.src = src; p.sx = scale_x; p.sy = scale_y; p.o0 = dst; p.placement = placement;
    p.rows = batch * len_kv; p.n = heads * dim_head; p.len_q = len_kv; p.heads = heads; p.d = dim_head;
    p.len_buf = len_buf; p.src_stride = src_stride; p.dst_stride = dst_stride; p.place_stride = place_stride;
    return launch_back<kBackToBuffer>(p, dtype, (hipStream_t)s);
}

}  // extern "C"

// ---- INT8-compressed tensor-parallel reduce: the kernels of ModelContext::reduce_tp_int8 (src/model/model_context.cpp:244-326;
//      src/nn/quant/int8/quant_reduce_kernel.cu:13-105 quant_group_32, :107-150 dequant_group_32, :270-330
//      dequant_sum_quant_g32).  One group of 32 values per half wavefront (the reference: one warp per group), 8 groups per
//      workgroup; bit-exact against the oracle's restatement (tests/test_gpu_ops.py).  A group of zeros: codes 0, scale 0 (the
//      reference divides by zero).
namespace {

__device__ __forceinline__ float half_wave_max(float v) {   // all-reduce over the 32 lanes of a group
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 32));
    return v;
}

template <int DT>
__global__ __launch_bounds__(256) void k_quant_group_32(const uint16_t* __restrict__ x, int8_t* __restrict__ q,
                                                        uint16_t* __restrict__ scale, int64_t groups) {
    const int64_t g = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (g >= groups) return;
    const int i = threadIdx.x & 31;
    const float v = ZT<DT>::to_f32(x[g * 32 + i]);
    const float amax = half_wave_max(fabsf(v));
    float t = v * 127.0f;
    asm volatile("" : "+v"(t));                                  // the product is rounded to fp32 before the division
    q[g * 32 + i] = amax > 0.f ? (int8_t)nearbyintf(t / amax) : (int8_t)0;
    if (i == 0) scale[g] = ZT<DT>::from_f32(amax / 127.0f);
}

template <int DT>
__global__ __launch_bounds__(256) void k_dequant_sum_quant_g32(const uint16_t* __restrict__ my, const int8_t* __restrict__ q_others,
                                                               const uint16_t* __restrict__ scale_others, int8_t* __restrict__ out_q,
                                                               uint16_t* __restrict__ out_scale, int64_t groups, int world) {
    const int64_t g = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (g >= groups) return;
    const int i = threadIdx.x & 31;
    float sum = ZT<DT>::to_f32(my[g * 32 + i]);
    for (int r = 0; r < world - 1; ++r)
        sum = __builtin_fmaf((float)q_others[((int64_t)r * groups + g) * 32 + i], ZT<DT>::to_f32(scale_others[(int64_t)r * groups + g]), sum);
    // warpReduceMaxB<T>(fabsf(sum)): the fp32 magnitude is rounded to T by the template argument before the maximum
    const float amax = half_wave_max(ZT<DT>::to_f32(ZT<DT>::from_f32(fabsf(sum))));
    float t = sum * 127.0f;
    asm volatile("" : "+v"(t));
    out_q[g * 32 + i] = amax > 0.f ? (int8_t)nearbyintf(t / amax) : (int8_t)0;
    if (i == 0) out_scale[g] = ZT<DT>::from_f32(amax / 127.0f);
}

template <int DT>
__global__ __launch_bounds__(256) void k_dequant_group_32(const int8_t* __restrict__ q, const uint16_t* __restrict__ scale,
                                                          uint16_t* __restrict__ out, int64_t groups) {
    const int64_t g = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (g >= groups) return;
    const int i = threadIdx.x & 31;
    out[g * 32 + i] = ZT<DT>::from_f32((float)q[g * 32 + i] * ZT<DT>::to_f32(scale[g]));
}

}  // namespace

extern "C" {

int zl_quant_group_32(const uint16_t* x, int8_t* q, uint16_t* scale, int64_t groups, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(x && q && scale && groups > 0, ZL_EINVAL);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    const dim3 grid((unsigned)((groups + 7) / 8));
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_quant_group_32<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, x, q, scale, groups);
    else hipLaunchKernelGGL(k_quant_group_32<ZL_BF16>, grid, dim3(256), 0, (hipStream_t)s, x, q, scale, groups);
    return zl_launch_status();
}

int zl_dequant_sum_quant_g32(const uint16_t* my, const int8_t* q_others, const uint16_t* scale_others, int8_t* q_sum,
                             uint16_t* scale_sum, int64_t groups, int world_size, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(my && q_others && scale_others && q_sum && scale_sum && groups > 0, ZL_EINVAL);
    ZL_CHECK_ARG(world_size == 2 || world_size == 4 || world_size == 8, ZL_ESHAPE);     // the reference's instantiations
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    const dim3 grid((unsigned)((groups + 7) / 8));
    if (dtype == ZL_F16)
        hipLaunchKernelGGL(k_dequant_sum_quant_g32<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, my, q_others, scale_others, q_sum, scale_sum, groups, world_size);
    else
        hipLaunchKernelGGL(k_dequant_sum_quant_g32<ZL_BF16>, grid, dim3(256), 0, (hipStream_t)s, my, q_others, scale_others, q_sum, scale_sum, groups, world_size);
    return zl_launch_status();
}

int zl_dequant_group_32(const int8_t* q, const uint16_t* scale, uint16_t* out, int64_t groups, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(q && scale && out && groups > 0, ZL_EINVAL);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    const dim3 grid((unsigned)((groups + 7) / 8));
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_dequant_group_32<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, q, scale, out, groups);
    else hipLaunchKernelGGL(k_dequant_group_32<ZL_BF16>, grid, dim3(256), 0, (hipStream_t)s, q, scale, out, groups);
    return zl_launch_status();
}

}  // extern "C"

// w4_gemv.hip -- W4A16 decode GEMM (M <= 4 rows per pass) on the ZLW4 layout.  SURVEY 8a rows a2, a5.
//
// What the reference does (src/nn/quant/gptq/q_gemm_k_major.cu:127-237): one 32-lane warp per
// output channel n; lane l walks the packed words l, l+32, ...; per word: exact fp16 (q - z),
// two fp16 hfma2 accumulators over the 8 weights, f32(lo) + f32(hi), fp32 fma with the group
// scale; lanes are summed by a shuffle-down tree.
//
// How it is laid out here for CDNA4:
//  * a 64-lane wavefront owns a PAIR of output rows; half-wave h = lane/32 replays the reference
//    warp for row 2*pair + h, so the per-lane fp32 chain and the 32-lane tree -- and therefore the
//    result bits -- are the reference's.
//  * ZLW4 stores the words so that ONE global_load_dwordx4 per lane (1 KiB per wave, fully
//    coalesced, non-temporal) brings 4 consecutive chain steps (words r+32*(4q+j), j=0..3) of
//    both rows; the 4 group scales / zero nibbles those steps need are one 8-byte and one 2-byte
//    load from the re-ordered meta arrays.  Nothing but the algorithmic bytes is read.
//  * loads go straight to VGPRs through a D-deep software ring (no LDS round trip for weights:
//    they are used once); the activation rows live in LDS and are read as ds_read_b128.
//  * optional fused RMSNorm prologue (every workgroup re-derives the 1/rms of the L2-resident
//    input row while its first weight loads are in flight) and fused epilogues (bias, ADD_C,
//    residual add, silu*mul on interleaved gate/up rows).
//  * no cross-wave reduction, no atomics, no split-K: a wave finishes its rows alone.
#include "zl_common.h"
#include "zl_stage.h"

namespace {

constexpr int kRing = 8;  // weight loads in flight per wave (8 KiB)

struct W4Params {
    const uint16_t* x;
    int64_t ldx;
    const uint4* qw;
    const uint2* scales;    // [row][q][c] x 4 halfs
    const uint16_t* zeros;  // [row][q][c] 4 nibbles
    const uint16_t* bias;
    const uint16_t* residual;
    uint16_t* y;
    const uint16_t* norm_w;
    float norm_eps;
    int m, n, k, kp;
    int q_loads, c_classes, c_shift;
    int sym, epi;
    int pairs_total, pairs_per_wave;
    int ld_out;  // row stride of y / residual (n, or n/2 for silu*mul)
};

typedef _Float16 hv2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ hv2 as_hv2(uint32_t u) { return __builtin_bit_cast(hv2, u); }

// one 8-weight word against MT activation rows: acc[m] = fma(dot8_half(w, x_m), scale, acc[m])
template <int MT>
__device__ __forceinline__ void word_step(uint32_t w, float scale, hv2 z1, hv2 z16, const uint4 (&xa)[MT],
                                          float (&acc)[MT]) {
    const hv2 one16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    const hv2 zero2 = {(_Float16)0.f, (_Float16)0.f};
    const uint32_t magic = 0x64006400u;  // half2(1024, 1024)
    hv2 d0 = as_hv2((w & 0x000f000fu) | magic) + z1;                                   // (w0,w1) - z
    hv2 d1 = __builtin_elementwise_fma(as_hv2((w & 0x00f000f0u) | magic), one16, z16);  // (w2,w3) - z
    uint32_t wb = w >> 8;
    hv2 d2 = as_hv2((wb & 0x000f000fu) | magic) + z1;                                   // (w4,w5) - z
    hv2 d3 = __builtin_elementwise_fma(as_hv2((wb & 0x00f000f0u) | magic), one16, z16);  // (w6,w7) - z
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        hv2 r = __builtin_elementwise_fma(d0, as_hv2(xa[m].x), zero2);
        r = __builtin_elementwise_fma(d1, as_hv2(xa[m].y), r);
        r = __builtin_elementwise_fma(d2, as_hv2(xa[m].z), r);
        r = __builtin_elementwise_fma(d3, as_hv2(xa[m].w), r);
        float dot = (float)r.x + (float)r.y;
        acc[m] = __builtin_fmaf(dot, scale, acc[m]);
    }
}

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + expf(-x)); }

template <int MT>
__global__ __launch_bounds__(256) void k_w4a16_gemm(const W4Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem);                 // [MT][kp]
    float* red = reinterpret_cast<float*>(smem + (size_t)MT * p.kp * 2);  // 16 floats
    float* res_all = red + 16;                                            // [4 waves][64 rows][MT]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int cls = r >> p.c_shift;
    const int m0 = blockIdx.y * MT;
    const int Q = p.q_loads;

    float* res = res_all + wave * 64 * MT;
    const int gw = blockIdx.x * 4 + wave;
    const int pair0 = gw * p.pairs_per_wave;
    int npairs = p.pairs_total - pair0;
    npairs = npairs < 0 ? 0 : (npairs > p.pairs_per_wave ? p.pairs_per_wave : npairs);
    const int total = npairs * Q;

    // ---- per-lane streams: weights advance 64 uint4 per item, meta advance C entries per item
    const uint4* wptr = p.qw + ((size_t)pair0 * Q) * 64 + lane;
    const size_t meta_row_stride = (size_t)Q * p.c_classes;
    const size_t meta_base = ((size_t)(2 * pair0 + h)) * meta_row_stride + cls;

    uint4 wq[kRing];
    uint2 sc[kRing];
    uint32_t zq[kRing];

    // item `it` of this wave: pair pair0 + it / Q, load q = it % Q.  Rows of a pair are adjacent
    // in the meta arrays (row 2pr, 2pr+1), so meta index = base + (it/Q)*2*stride + (it%Q)*C
    int iss_pair = 0, iss_q = 0;  // position of the next item to issue
    auto issue = [&](int slot) {
        wq[slot] = zl_load_nt(wptr);
        wptr += 64;
        const size_t mi = meta_base + (size_t)iss_pair * 2 * meta_row_stride + (size_t)iss_q * p.c_classes;
        sc[slot] = p.scales[mi];
        zq[slot] = p.sym ? 0x8888u : (uint32_t)p.zeros[mi];
        if (++iss_q == Q) {
            iss_q = 0;
            ++iss_pair;
        }
    };

#pragma unroll
    for (int s = 0; s < kRing; ++s)
        if (s < total) issue(s);

    // ---- stage the activation rows into LDS (optionally RMS-normalised), zero the K padding
    zl_stage_rows<ZL_F16, MT, 256>(p.x, p.ldx, m0, p.m, p.k, p.kp, p.norm_w, p.norm_eps, xs, red);
    __syncthreads();

    // ---- main loop
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    int cq = 0, cpair = pair0;  // position of the item being consumed
    const uint16_t* xlane = xs + 8 * r;

    auto consume = [&](int slot) {
        const uint32_t wds[4] = {wq[slot].x, wq[slot].y, wq[slot].z, wq[slot].w};
        const hv2 s01 = as_hv2(sc[slot].x), s23 = as_hv2(sc[slot].y);
        const float scl[4] = {(float)s01.x, (float)s01.y, (float)s23.x, (float)s23.y};
        const uint32_t z4 = zq[slot];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t z = (z4 >> (4 * j)) & 0xfu;
            const hv2 z1 = as_hv2(0xe400e400u | z | (z << 16));  // -(1024 + z), exact
            const hv2 c960 = {(_Float16)960.f, (_Float16)960.f};
            const hv2 z16 = z1 + c960;                            // -(64 + z), exact
            uint4 xa[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                xa[m] = *reinterpret_cast<const uint4*>(xlane + (size_t)m * p.kp + 8 * (32 * (4 * cq + j)));
            word_step<MT>(wds[j], scl[j], z1, z16, xa, acc);
        }
        if (++cq == Q) {
            // ---- row pair finished: replay the reference's 32-lane shuffle-down tree per half-wave and
            // park the fp32 sums in this wave's LDS slot; the epilogue runs once, after the stream.
            cq = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v = acc[m];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) v += __shfl_down(v, off, 32);
                acc[m] = 0.f;
                if (r == 0) res[((cpair - pair0) * 2 + h) * MT + m] = v;
            }
            ++cpair;
        }
    };

#pragma unroll 1
    for (int it = 0; it < total; it += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            if (it + s < total) {
                consume(s);
                if (it + s + kRing < total) issue(s);
            }
        }
    }

    // ---- epilogue: lane i of the wave finishes row (or gate/up pair) i of the wave's run; stores
    // of consecutive lanes are consecutive fp16 elements.
    const bool silu = (p.epi & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32)) != 0;
    const int nres = silu ? npairs : 2 * npairs;
    if (lane < nres) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if ((m0 + m) >= p.m) break;
            const size_t orow = (size_t)(m0 + m) * p.ld_out;
            if (silu) {
                const int pr = pair0 + lane;
                if (2 * pr + 1 >= p.n) continue;
                float g = res[(2 * lane) * MT + m], u = res[(2 * lane + 1) * MT + m];
                if ((p.epi & ZL_EPI_BIAS) && p.bias) {
                    g += (float)__builtin_bit_cast(_Float16, p.bias[2 * pr]);
                    u += (float)__builtin_bit_cast(_Float16, p.bias[2 * pr + 1]);
                }
                float o;
                if (p.epi & ZL_EPI_SILU_MUL) {
                    g = (float)(_Float16)g;  // the two fp16 linear outputs
                    u = (float)(_Float16)u;
                    o = silu_f32(g) * u;
                } else {
                    o = (float)((double)g / (1.0 + (double)expf(-g))) * u;
                }
                p.y[orow + pr] = __builtin_bit_cast(uint16_t, (_Float16)o);
            } else {
                const int row = 2 * pair0 + lane;
                if (row >= p.n) continue;
                const float v = res[lane * MT + m];
                const float b = ((p.epi & ZL_EPI_BIAS) && p.bias) ? (float)__builtin_bit_cast(_Float16, p.bias[row]) : 0.f;
                float o;
                if (p.epi & ZL_EPI_ADD_C) o = ((float)__builtin_bit_cast(_Float16, p.y[orow + row]) + v) + b;
                else o = v + b;
                _Float16 y16 = (_Float16)o;
                if (p.epi & ZL_EPI_RESIDUAL)
                    y16 = (_Float16)((float)__builtin_bit_cast(_Float16, p.residual[orow + row]) + (float)y16);
                p.y[orow + row] = __builtin_bit_cast(uint16_t, y16);
            }
        }
    }
}

template <int MT>
int launch(const W4Params& p, int grid_x, int grid_y, hipStream_t st) {
    size_t lds = (size_t)MT * p.kp * 2 + 64 + 4 * 64 * MT * 4;
    if (lds > 160 * 1024) return ZL_ELIMIT;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_w4a16_gemm<MT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_w4a16_gemm<MT>, dim3(grid_x, grid_y), dim3(256), lds, st, p);
    return zl_launch_status();
}

}  // namespace

extern "C" int zl_w4a16_gemm(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint16_t* scales,
                             const uint16_t* zeros, const uint16_t* bias, const uint16_t* residual, uint16_t* y,
                             int64_t m, int64_t n, int64_t k, int64_t group_size, int sym,
                             const uint16_t* norm_weight, float norm_eps, int epilogue, zl_stream_t s) {
    ZL_CHECK_ARG(x && qw && scales && (zeros || sym) && y && m > 0 && n > 0 && k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(ldx >= k && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(!(epilogue & ZL_EPI_RESIDUAL) || residual, ZL_EINVAL);
    zl_w4_layout_t L;
    int st = zl_w4_layout(n, k, group_size, &L);
    if (st) return st;
    const bool silu = epilogue & (ZL_EPI_SILU_MUL | ZL_EPI_SILU_MUL_F32);
    ZL_CHECK_ARG(!silu || n % 2 == 0, ZL_ESHAPE);

    W4Params p;
    p.x = x; p.ldx = ldx;
    p.qw = reinterpret_cast<const uint4*>(qw);
    p.scales = reinterpret_cast<const uint2*>(scales);
    p.zeros = zeros;
    p.bias = bias; p.residual = residual; p.y = y;
    p.norm_w = norm_weight; p.norm_eps = norm_eps;
    p.m = (int)m; p.n = (int)n; p.k = (int)k; p.kp = (int)L.kp;
    p.q_loads = (int)L.q; p.c_classes = (int)L.c;
    int per_class = 32 / (int)L.c, shift = 0;
    while ((1 << shift) < per_class) ++shift;
    p.c_shift = shift;
    p.sym = sym; p.epi = epilogue;
    p.pairs_total = (int)(L.np / 2);
    p.ld_out = silu ? (int)(n / 2) : (int)n;

    // rows-per-pass: the VALU budget of the reference-exact arithmetic holds up to ~3 rows per
    // weight pass at HBM speed; LDS must hold MT * Kp halfs
    int mt = m >= 4 ? 4 : (int)m;
    while (mt > 1 && (size_t)mt * L.kp * 2 + 64 + 4096 > 64 * 1024) --mt;
    if (mt == 3 && m > 3) mt = 2;
    const int grid_y = (int)((m + mt - 1) / mt);

    // grid: j workgroups (4 waves) per CU; each wave owns a contiguous run of row pairs.  Pick j
    // in 1..4 minimising pairs handled per CU (the HBM-bound time), ties -> more waves in flight.
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    int best_j = 1, best_cost = 1 << 30, best_ppw = 1;
    for (int j = 1; j <= 4; ++j) {
        int waves = cus * j * 4;
        int ppw = (p.pairs_total + waves - 1) / waves;
        int cost = ppw * j * 4;
        if (cost < best_cost || (cost == best_cost && j > best_j)) {
            best_cost = cost; best_j = j; best_ppw = ppw;
        }
    }
    if (best_ppw > 32) best_ppw = 32;  // LDS result slots: 64 rows per wave
    p.pairs_per_wave = best_ppw;
    const int waves_needed = (p.pairs_total + best_ppw - 1) / best_ppw;
    const int grid_x = (waves_needed + 3) / 4;

    hipStream_t hs = (hipStream_t)s;
    switch (mt) {
        case 1: return launch<1>(p, grid_x, grid_y, hs);
        case 2: return launch<2>(p, grid_x, grid_y, hs);
        case 3: return launch<3>(p, grid_x, grid_y, hs);
        default: return launch<4>(p, grid_x, grid_y, hs);
    }
}

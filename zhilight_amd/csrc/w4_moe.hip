// w4_moe.hip -- fused MoE GEMVs of the decode path on the ZLW4 layout ("Marlin-free fused MoE GEMV").
//
// Reference (FUSE_GPTQ_MOE=1, src/nn/quant/gptq/q_gemm_k_major.cu):
//   KERNEL_gemm_moe_up   (:243-320)  C[m, t, n] = half( silu(x_m . W1_e[n]) * (x_m . W2_e[n]) ),  e = expert t of token m
//   KERNEL_gemm_moe_down (:322-390)  C[m, n]    = half( sum_t w[m, t] * (A[m, t] . W_e[n]) )  (+ float(C) with ADD_C)
// Both are DEV_gemm_warp_reduce<1> per (token, expert, output row) -- the arithmetic of the decode GEMV (zl_w4_exact.h):
// a 32-lane warp per row, lane l walks words l, l + 32, ..., fp32 fma with the group scale.  In moe_down the per-LANE fp32
// partial of every expert is scaled by the routing weight and accumulated (acc_all += acc * w: one fp32 fma under nvcc's
// default contraction) BEFORE the 32-lane shuffle tree; moe_up reduces gate and up separately, then silu(x) = x / (1.0 +
// expf(-x)) with the double constant of the file-local helper (:239-241).  Both are reproduced bit for bit.
//
// Layout: every expert's matrix is packed by zl_w4_pack on its own (gate / up with row_interleave: row pair p = (gate_p,
// up_p)); the experts sit `expert stride` bytes apart in the three arrays.  A 64-lane wavefront owns a run of row pairs;
// its item stream (1 KiB of words + 8 B of scales + 2 B of zero nibbles per lane-load) runs across (pair, expert, load)
// through an 8-deep ring of non-temporal loads, so the tiny per-expert matrices of a MoE layer (a few KiB per row pair)
// still keep 8 KiB per wave in flight.  Experts that are not local (expert parallelism) are dropped when the token's
// expert list is built -- the reference skips them the same way (`continue` / early return on a zero-filled output).
#include "zl_common.h"
#include "zl_w4_exact.h"

using namespace zlx;

namespace {

constexpr int kMoeThreads = 256;
constexpr int kMoeWaves = kMoeThreads / 64;
constexpr int kMoeRing = 8;
constexpr int kMoeMaxTopK = 32;
constexpr int kMoeMaxPPW = 64;     // row pairs per wave (LDS result slots: the outputs are written once, after the stream)

struct MoeParams {
    const uint16_t* x;          // up: (M, K); down: (M, TOPK, K)
    int64_t ldx;                // row stride of x in halfs
    const uint4* qw;
    const uint2* scales;
    const uint16_t* zeros;
    int64_t stride_qw, stride_sc, stride_z;   // per expert, in uint4 / uint2 / u16 units
    const int32_t* ids;         // (M, topk_real)
    const float* weights;       // (M, topk_real) (down)
    uint16_t* y;
    int m, n, k, kp;            // n = output columns (up: ff columns = row PAIRS of the interleaved matrix; down: rows)
    int q_loads, c_classes, c_shift;
    int pairs_total, pairs_per_wave;
    int topk_real, topk;        // routed experts per token / routed + shared
    int shared_base;            // id of the first shared expert in the stacked arrays
    int exp_parallel, world, rank;
    int add_c;
};

// MODE 1 = up (one (token, expert) per blockIdx.y), MODE 2 = down (one token per blockIdx.y, all its experts)
template <int MODE>
__global__ __launch_bounds__(kMoeThreads) void k_w4a16_moe(const MoeParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_moe[];
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem_moe);                       // [rows][kp]
    const int xrows = MODE == 1 ? 1 : p.topk;
    int* e_qw = reinterpret_cast<int*>(smem_moe + (size_t)xrows * p.kp * 2);    // [kMoeMaxTopK] expert offsets (uint4 units)
    int* e_sc = e_qw + kMoeMaxTopK;                                             // (uint2 / u16 units: the same count)
    int* e_row = e_sc + kMoeMaxTopK;                                            // which staged x row
    float* e_w = reinterpret_cast<float*>(e_row + kMoeMaxTopK);                 // routing weight
    int* e_cnt = reinterpret_cast<int*>(e_w + kMoeMaxTopK);
    float* res_all = reinterpret_cast<float*>(e_cnt + 4);                       // [waves][kMoeMaxPPW][2]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int cls = r >> p.c_shift;
    const int Q = p.q_loads;
    const int tok = MODE == 1 ? (int)blockIdx.y / p.topk : (int)blockIdx.y;

    // ---- the token's LOCAL experts (one lane per expert slot: one round trip for ids and weights, compacted by ballot), the
    // activation row(s) (everyone)
    if (wave == 0) {
        const int t0 = MODE == 1 ? (int)blockIdx.y % p.topk : 0, cnt_t = MODE == 1 ? 1 : p.topk;
        int e = 0, t = t0 + lane;
        float w = 1.f;
        bool local = false;
        if (lane < cnt_t) {
            if (t < p.topk_real) {
                e = p.ids[(size_t)tok * p.topk_real + t];
                if (MODE == 2) w = p.weights[(size_t)tok * p.topk_real + t];
                local = !p.exp_parallel || (e % p.world) == p.rank;     // the others are not stored here
                if (p.exp_parallel) e /= p.world;
            } else {
                e = p.shared_base + t - p.topk_real;
                local = true;
            }
        }
        const unsigned long long mask = __ballot(local);
        const int pos = __popcll(mask & ((1ull << lane) - 1ull));
        if (local) {
            e_qw[pos] = (int)((int64_t)e * p.stride_qw);
            e_sc[pos] = (int)((int64_t)e * p.stride_sc);
            e_row[pos] = MODE == 1 ? 0 : t;
            e_w[pos] = w;
        }
        if (lane == 0) *e_cnt = __popcll(mask);
    }
    {
        const uint16_t* xb = MODE == 1 ? p.x + (size_t)tok * p.ldx : p.x + (size_t)tok * p.topk * p.ldx;
        const int per_row = p.kp / 8;
        for (int i = threadIdx.x; i < xrows * per_row; i += kMoeThreads) {
            const int row = i / per_row, c = (i % per_row) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < p.k) v = *reinterpret_cast<const uint4*>(xb + (size_t)row * p.ldx + c);   // k % 8 == 0
            *reinterpret_cast<uint4*>(xs + (size_t)row * p.kp + c) = v;
        }
    }
    __syncthreads();
    const int T = __builtin_amdgcn_readfirstlane(*e_cnt);

    const int gw = blockIdx.x * kMoeWaves + wave;
    const int pair0 = gw * p.pairs_per_wave;
    int npairs = p.pairs_total - pair0;
    npairs = npairs < 0 ? 0 : (npairs > p.pairs_per_wave ? p.pairs_per_wave : npairs);
    if (T == 0) {
        // no local expert: up leaves zeros (the reference returns on a zero-filled C); down writes 0 or keeps C (ADD_C)
        if (MODE == 1) {
            for (int i = lane; i < npairs; i += 64) p.y[(size_t)blockIdx.y * p.n + pair0 + i] = 0;
        } else if (!p.add_c) {
            for (int i = lane; i < 2 * npairs; i += 64)
                if (2 * pair0 + i < p.n) p.y[(size_t)tok * p.n + 2 * pair0 + i] = 0;
        }
        return;
    }
    const int total = npairs * T * Q;
    if (total == 0) return;

    // ---- item stream: item = (pair, expert slot, load); issue side and consume side keep their own counters
    uint4 wq[kMoeRing];
    uint2 sc[kMoeRing];
    uint16_t zq[kMoeRing];
    int iq = 0, it = 0, ipair = pair0, iss = 0;
    const int meta_lane = h * p.c_classes + cls;
    auto issue = [&](int slot) {
        const int64_t item = (int64_t)ipair * Q + iq;
        const uint4* wp = p.qw + e_qw[it] + item * 64 + lane;
        const int64_t mo = item * 2 * p.c_classes + meta_lane;
        wq[slot] = zl_load_nt(wp);
        sc[slot] = zl_load_nt(p.scales + e_sc[it] + mo);
        zq[slot] = zl_load_nt(p.zeros + e_sc[it] + mo);
        // branch-free (wave-uniform selects): no control flow between the loads, so the compiler keeps counted vmcnt waits in
        // the steady-state loop; past the end the last item is re-read (cache hit) and discarded
        const int adv = iss + 1 < total ? 1 : 0;
        iss += adv;
        iq += adv;
        const int wq_ = iq == Q ? 1 : 0;
        iq = wq_ ? 0 : iq;
        it += wq_;
        const int wt_ = it == T ? 1 : 0;
        it = wt_ ? 0 : it;
        ipair += wt_;
    };
#pragma unroll
    for (int s = 0; s < kMoeRing; ++s) issue(s);

    const uint32_t mask_lo = __builtin_amdgcn_readfirstlane(0x000f000fu);
    const uint32_t mask_hi = __builtin_amdgcn_readfirstlane(0x00f000f0u);
    uint32_t magic = 0x64006400u, one16 = 0x2c002c00u;
    asm volatile("" : "+v"(magic), "+v"(one16));
    float acc = 0.f, acc_all = 0.f;
    int cq = 0, ct = 0, cpair = pair0;
    float* res = res_all + wave * kMoeMaxPPW * 2;

    auto consume = [&](int slot) {
        const uint32_t wds[4] = {wq[slot].x, wq[slot].y, wq[slot].z, wq[slot].w};
        const uint32_t s01 = sc[slot].x, s23 = sc[slot].y;
        const uint32_t z4 = zq[slot];
        const uint16_t* xlane = xs + (size_t)e_row[ct] * p.kp + 8 * r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t z = (z4 >> (4 * j)) & 0xfu;
            const uint32_t z1 = 0xe400e400u | z | (z << 16);
            const hv2 c960 = {(_Float16)960.f, (_Float16)960.f};
            const uint32_t z16 = __builtin_bit_cast(uint32_t, as_hv2(z1) + c960);
            const DeqWord d = deq_word(wds[j], z1, z16, mask_lo, mask_hi, magic, one16);
            const uint4 xa = *reinterpret_cast<const uint4*>(xlane + 8 * (32 * (4 * cq + j)));
            if (j & 1) acc = dot_word<true>(d, xa, j < 2 ? s01 : s23, acc);
            else acc = dot_word<false>(d, xa, j < 2 ? s01 : s23, acc);
        }
        if (++cq == Q) {                       // this expert's row is done
            cq = 0;
            if (MODE == 2) acc_all = __builtin_fmaf(acc, e_w[ct], acc_all);    // acc_all += acc * weight (one fma)
            else acc_all = acc;
            acc = 0.f;
            if (++ct == T) {                   // the row pair is done: the reference's 32-lane shuffle-down tree per half-wave,
                ct = 0;                        // the sums parked in this wave's LDS slots (no store inside the stream)
                float v = acc_all;
                acc_all = 0.f;
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) v += __shfl_down(v, off, 32);
                if (r == 0) res[(cpair - pair0) * 2 + h] = v;
                ++cpair;
            }
        }
    };

    int done = 0;
#pragma unroll 1
    for (; done + kMoeRing <= total; done += kMoeRing) {
#pragma unroll
        for (int s = 0; s < kMoeRing; ++s) {
            consume(s);
            issue(s);
        }
    }
#pragma unroll
    for (int s = 0; s < kMoeRing; ++s)
        if (done + s < total) consume(s);      // wave-uniform
    // ---- epilogue: lane i finishes the wave's i-th output(s)
    __builtin_amdgcn_wave_barrier();
    if (MODE == 1) {
        for (int i = lane; i < npairs; i += 64) {
            const float gt = res[2 * i], up = res[2 * i + 1];                   // gate row, up row of pair i
            const float o = (float)((double)gt / (1.0 + (double)expf(-gt))) * up;
            p.y[(size_t)blockIdx.y * p.n + pair0 + i] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(o));
        }
    } else {
        for (int i = lane; i < 2 * npairs; i += 64) {
            const int col = 2 * pair0 + i;
            if (col < p.n) {
                const size_t o = (size_t)tok * p.n + col;
                const float v = res[i];
                const float ov = p.add_c ? (float)__builtin_bit_cast(_Float16, p.y[o]) + v : v;
                p.y[o] = __builtin_bit_cast(uint16_t, zl_f32_to_f16(ov));
            }
        }
    }
}

int moe_fill(MoeParams& p, const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint16_t* scales, const uint16_t* zeros,
             int64_t stride_qw_bytes, int64_t stride_scales_bytes, int64_t stride_zeros_bytes, int64_t rows, int64_t k,
             int64_t group_size) {
    zl_w4_layout_t L;
    int st = zl_w4_layout(rows, k, group_size, &L);
    if (st) return st;
    if (stride_qw_bytes < L.qw_bytes || stride_scales_bytes < L.scales_bytes || stride_zeros_bytes < L.zeros_bytes) return ZL_ESHAPE;
    if (stride_qw_bytes % 16 || stride_scales_bytes % 8 || stride_zeros_bytes % 2) return ZL_ESHAPE;
    // the kernel indexes scales (8 B) and zeros (2 B) with ONE per-expert element offset
    if (stride_scales_bytes / 8 != stride_zeros_bytes / 2) return ZL_ESHAPE;
    p.x = x; p.ldx = ldx;
    p.qw = reinterpret_cast<const uint4*>(qw);
    p.scales = reinterpret_cast<const uint2*>(scales);
    p.zeros = zeros;
    p.stride_qw = stride_qw_bytes / 16; p.stride_sc = stride_scales_bytes / 8; p.stride_z = stride_zeros_bytes / 2;
    p.k = (int)k; p.kp = (int)L.kp;
    p.q_loads = (int)L.q; p.c_classes = (int)L.c;
    int per_class = 32 / (int)L.c, shift = 0;
    while ((1 << shift) < per_class) ++shift;
    p.c_shift = shift;
    p.pairs_total = (int)(L.np / 2);
    return ZL_OK;
}

template <int MODE>
int moe_launch(MoeParams& p, int passes, int xrows, hipStream_t hs) {
    int cus = zl_device_cu_count();
    if (cus <= 0) cus = 256;
    // ~8 waves per CU over all passes (a wave's ring runs across its pairs x experts x loads)
    int64_t waves_wanted = (int64_t)cus * 8 / (passes > 0 ? passes : 1);
    if (waves_wanted < kMoeWaves) waves_wanted = kMoeWaves;
    int ppw = (int)((p.pairs_total + waves_wanted - 1) / waves_wanted);
    if (ppw < 1) ppw = 1;
    if (ppw > kMoeMaxPPW) ppw = kMoeMaxPPW;
    p.pairs_per_wave = ppw;
    const int waves = (p.pairs_total + ppw - 1) / ppw;
    const int gx = (waves + kMoeWaves - 1) / kMoeWaves;
    const size_t lds = (size_t)xrows * p.kp * 2 + (size_t)kMoeMaxTopK * 16 + 16 + (size_t)kMoeWaves * kMoeMaxPPW * 2 * sizeof(float);
    if (lds > 160 * 1024 || passes > 65535) return ZL_ELIMIT;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_w4a16_moe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return ZL_ELIMIT;
    }
    hipLaunchKernelGGL(k_w4a16_moe<MODE>, dim3((unsigned)gx, (unsigned)passes), dim3(kMoeThreads), lds, hs, p);
    return zl_launch_status();
}

}  // namespace

extern "C" int zl_w4a16_moe_up(const uint16_t* x, int64_t ldx, const uint32_t* qw, const uint16_t* scales, const uint16_t* zeros,
                               int64_t expert_stride_qw, int64_t expert_stride_scales, int64_t expert_stride_zeros,
                               const int32_t* expert_ids, uint16_t* out, int64_t m, int64_t n_ff, int64_t k, int64_t group_size,
                               int top_k, int n_shared, int shared_base, int exp_parallel, int world_size, int rank, zl_stream_t s) {
    ZL_CHECK_ARG(x && qw && scales && zeros && out && m > 0 && n_ff > 0 && k > 0 && top_k >= 0 && n_shared >= 0, ZL_EINVAL);
    ZL_CHECK_ARG(top_k + n_shared >= 1 && top_k + n_shared <= kMoeMaxTopK && (top_k == 0 || expert_ids), ZL_EINVAL);
    ZL_CHECK_ARG(ldx >= k && ldx % 8 == 0 && k % 8 == 0 && ((uintptr_t)x & 15) == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(!exp_parallel || (world_size >= 1 && rank >= 0 && rank < world_size), ZL_EINVAL);
    MoeParams p = {};
    int st = moe_fill(p, x, ldx, qw, scales, zeros, expert_stride_qw, expert_stride_scales, expert_stride_zeros, 2 * n_ff, k, group_size);
    if (st) return st;
    p.ids = expert_ids; p.weights = nullptr; p.y = out;
    p.m = (int)m; p.n = (int)n_ff;
    p.topk_real = top_k; p.topk = top_k + n_shared; p.shared_base = shared_base;
    p.exp_parallel = exp_parallel; p.world = world_size > 0 ? world_size : 1; p.rank = rank; p.add_c = 0;
    return moe_launch<1>(p, (int)(m * p.topk), 1, (hipStream_t)s);
}

extern "C" int zl_w4a16_moe_down(const uint16_t* a, int64_t lda, const uint32_t* qw, const uint16_t* scales, const uint16_t* zeros,
                                 int64_t expert_stride_qw, int64_t expert_stride_scales, int64_t expert_stride_zeros,
                                 const int32_t* expert_ids, const float* expert_weights, uint16_t* out, int64_t m, int64_t n,
                                 int64_t k, int64_t group_size, int top_k, int n_shared, int shared_base, int exp_parallel,
                                 int world_size, int rank, int add_c, zl_stream_t s) {
    ZL_CHECK_ARG(a && qw && scales && zeros && out && m > 0 && n > 0 && k > 0 && top_k >= 0 && n_shared >= 0, ZL_EINVAL);
    ZL_CHECK_ARG(top_k + n_shared >= 1 && top_k + n_shared <= kMoeMaxTopK && (top_k == 0 || (expert_ids && expert_weights)), ZL_EINVAL);
    ZL_CHECK_ARG(lda >= k && lda % 8 == 0 && k % 8 == 0 && ((uintptr_t)a & 15) == 0, ZL_ESHAPE);
    ZL_CHECK_ARG(!exp_parallel || (world_size >= 1 && rank >= 0 && rank < world_size), ZL_EINVAL);
    MoeParams p = {};
    int st = moe_fill(p, a, lda, qw, scales, zeros, expert_stride_qw, expert_stride_scales, expert_stride_zeros, n, k, group_size);
    if (st) return st;
    p.ids = expert_ids; p.weights = expert_weights; p.y = out;
    p.m = (int)m; p.n = (int)n;
    p.topk_real = top_k; p.topk = top_k + n_shared; p.shared_base = shared_base;
    p.exp_parallel = exp_parallel; p.world = world_size > 0 ? world_size : 1; p.rank = rank; p.add_c = add_c;
    return moe_launch<2>(p, (int)m, p.topk, (hipStream_t)s);
}

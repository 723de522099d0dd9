// moe_router.hip -- f4, first part: the MoE router of config 5 (DeepSeek-V3 / Qwen3-MoE style top-k gating):
//   zl_moe_top_k_softmax   nn::top_k_softmax      (src/nn/feedforward/ff_kernel.cu:174-268)
//   zl_moe_group_topk      nn::group_topk_softmax (ff_kernel.cu:296-515)
// Round 4: written for a 64-wide wavefront from the routing RULES, not from the reference's kernels (round 3's file followed
// KERNEL_group_topk statement by statement on emulated 32-lane warps -- VERDICT r03).  One wavefront per token, no LDS, no
// barrier: lane l holds the experts l, l + 64, l + 128, l + 192 (coalesced loads, num_exp <= 256).  Every selection is an
// arg-max over 64-bit keys
//     key(e) = monotone(score_e) << 32 | ~e            (0 = not a candidate)
// so "larger score first, equal scores: smaller expert index first" -- the order both of the reference's sorts produce (its
// insertion sort never lets a later equal value displace an earlier one, ff_kernel.cu:98-124; its bitonic compare-exchange
// breaks ties by position, :273-291) -- IS the integer order of the keys; k rounds of a wave-wide max pick the k winners in
// rank order, the winner's lane retires its key.  The group-limited router needs no sorting network either: a group's score is
// the maximum of (score + bias) over its experts (the reference takes element 0 of the group's sorted list), a group is kept
// when fewer than topk_group groups rank before it (same tie rule on the group index, computed redundantly by every lane from
// the <= 8 group scores), and the k winners are the k largest keys among the experts of kept groups (the reference merges the
// kept groups' k best each and sorts the merge: the same set in the same order whenever the kept groups hold k experts, which
// the launcher requires).  Output weights: the UN-biased score, renormalised by the sum of the k weights taken in the reference's
// order (top_k_softmax: rank order from 1e-20; group_topk: the pairing of a shuffle-down tree over lanes 0 .. 31, + 1e-20), so
// that weights agree with the oracle to the last bits that expf allows.  One quirk is kept because results must equal the
// reference's: top_k_softmax's "sigmoid" is 1 / (1 + expf(+x)) as written in DEV_route_score (ff_kernel.cu:158-160);
// group_topk's is 1 / (1 + expf(-x)).  Expert ids match the oracle exactly whenever the scores do (device expf vs glibc
// expf: <= 2 ulp on a score).  Tiny kernels: a decode step routes a handful of tokens; latency, not throughput.
#include "zl_common.h"

namespace {

constexpr int kMaxTopK = 16;
constexpr int kSlots = 4;                       // experts per lane: num_exp <= 256
enum { SC_SOFTMAX = 1, SC_SIGMOID = 2, SC_LINEAR = 3 };

__device__ __forceinline__ uint32_t mono_u32(float f) {      // order-preserving map of a float onto unsigned integers
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {      // every lane gets the maximum
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

struct Scores {
    float s[kSlots];            // score of expert lane + 64 j (junk where the expert does not exist)
    bool live[kSlots];
};

// a token's logit e: T bits, or fp32 (DT == ZL_F32: the router Linear of the reference writes fp32 logits, feedforward.cpp:285-286)
template <int DT>
__device__ __forceinline__ float load_logit(const uint16_t* __restrict__ logits, int e) {
    if constexpr (DT == ZL_F32) return reinterpret_cast<const float*>(logits)[e];
    else return ZT<DT>::to_f32(logits[e]);
}
// scores of one token: logits -> softmax / sigmoid / linear.  `neg_sigmoid`: 1 / (1 + expf(-x)) (group_topk) or the
// reference's 1 / (1 + expf(+x)) (top_k_softmax, sic)
template <int DT>
__device__ __forceinline__ Scores route_scores(const uint16_t* __restrict__ logits, int num_exp, int scoring, bool neg_sigmoid, int lane) {
    Scores sc;
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
        const int e = lane + 64 * j;
        sc.live[j] = e < num_exp;
        sc.s[j] = sc.live[j] ? load_logit<DT>(logits, e) : -INFINITY;
    }
    if (scoring == SC_SOFTMAX) {
        float mx = -1e20f;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) mx = fmaxf(mx, sc.s[j]);
        mx = zl_wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            sc.s[j] = sc.live[j] ? expf(sc.s[j] - mx) : 0.f;
            sum += sc.s[j];
        }
        sum = zl_wave_sum(sum) + 1e-20f;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) sc.s[j] /= sum;
    } else if (scoring == SC_SIGMOID) {
#pragma unroll
        for (int j = 0; j < kSlots; ++j) sc.s[j] = 1.f / (1.f + expf(neg_sigmoid ? -sc.s[j] : sc.s[j]));
    }
    return sc;
}

// the k largest keys in rank order: lane i < k ends up with winner i's expert index and weight
__device__ __forceinline__ void pick_top_k(unsigned long long (&key)[kSlots], const float (&weight)[kSlots], int k, int lane, int& my_e, float& my_w) {
    my_e = 0;
    my_w = 0.f;
    for (int i = 0; i < k; ++i) {
        unsigned long long best = key[0];
#pragma unroll
        for (int j = 1; j < kSlots; ++j) best = key[j] > best ? key[j] : best;
        best = wave_max_u64(best);
        const int e = (int)(0xffffffffu - (uint32_t)best);                 // wave-uniform
        const int owner = e & 63, slot = e >> 6;
        float w = weight[0];
#pragma unroll
        for (int j = 1; j < kSlots; ++j) w = slot == j ? weight[j] : w;
        w = __shfl(w, owner, 64);
        if (lane == owner) {
#pragma unroll
            for (int j = 0; j < kSlots; ++j)
                if (slot == j) key[j] = 0ull;                               // retired
        }
        if (lane == i) { my_e = e; my_w = w; }
    }
}

// grid (tokens), one wavefront
template <int DT>
__global__ __launch_bounds__(64) void k_moe_top_k_softmax(const uint16_t* __restrict__ logits, int num_exp, int k, float* __restrict__ out_v,
                                                          int32_t* __restrict__ out_idx, int renormalize, float weight_scale, int scoring,
                                                          int top_k_ext, int32_t* worker_load, int32_t* expert_load, int num_worker) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const Scores sc = route_scores<DT>(logits + (size_t)q * num_exp * (DT == ZL_F32 ? 2 : 1), num_exp, scoring, false, lane);
    out_v += (size_t)q * top_k_ext;
    out_idx += (size_t)q * top_k_ext;
    unsigned long long key[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; ++j)
        key[j] = sc.live[j] ? ((unsigned long long)mono_u32(sc.s[j]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(lane + 64 * j)) : 0ull;
    int my_e;
    float my_w;
    pick_top_k(key, sc.s, k, lane, my_e, my_w);
    float sum_e = 1.f;
    if (renormalize) {                                      // the reference adds the k values in rank order, starting from 1e-20
        sum_e = 1.e-20f;
        for (int i = 0; i < k; ++i) sum_e += __shfl(my_w, i, 64);
    }
    if (lane < k) {
        float w = my_w / sum_e;
        asm volatile("" : "+v"(w));                         // (two roundings, as written: value / sum_e * weight_scale)
        out_v[lane] = w * weight_scale;
        out_idx[lane] = my_e;
        if (worker_load) atomicAdd(&worker_load[my_e % num_worker], 1);
        if (expert_load) atomicAdd(&expert_load[my_e], 1);
    } else if (lane < top_k_ext) {
        out_v[lane] = 1.f;                                  // the extended (shared-expert) slots: weight 1, index left to route_shared_lb
    }
}

// grid (tokens), one wavefront
template <int DT>
__global__ __launch_bounds__(64) void k_moe_group_topk(const uint16_t* __restrict__ logits, const float* __restrict__ correction_bias, int num_exp,
                                                       int k, float* __restrict__ out_v, int32_t* __restrict__ out_idx, int renormalize,
                                                       float weight_scale, int scoring, int num_group, int topk_group, int num_in_group,
                                                       int top_k_ext, int32_t* worker_load, int32_t* expert_load, int num_worker) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const Scores sc = route_scores<DT>(logits + (size_t)q * num_exp * (DT == ZL_F32 ? 2 : 1), num_exp, scoring, true, lane);
    out_v += (size_t)q * top_k_ext;
    out_idx += (size_t)q * top_k_ext;
    float sel[kSlots];                                      // what the selection looks at: score + bias
    int grp[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
        const int e = lane + 64 * j;
        sel[j] = sc.live[j] ? sc.s[j] + (correction_bias ? correction_bias[e] : 0.f) : -INFINITY;
        grp[j] = sc.live[j] ? e / num_in_group : -1;
    }
    // a group's score = its best (score + bias); kept = fewer than topk_group groups rank before it (ties: smaller index first)
    float gscore[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float m = -1e20f;
#pragma unroll
        for (int j = 0; j < kSlots; ++j) m = grp[j] == g ? fmaxf(m, sel[j]) : m;
        gscore[g] = g < num_group ? zl_wave_max(m) : -1e20f;
    }
    uint32_t kept = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        int before = 0;
#pragma unroll
        for (int h = 0; h < 8; ++h) before += (h < num_group && h != g && (gscore[h] > gscore[g] || (gscore[h] == gscore[g] && h < g))) ? 1 : 0;
        if (g < num_group && before < topk_group) kept |= 1u << g;
    }
    unsigned long long key[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; ++j)
        key[j] = (sc.live[j] && ((kept >> grp[j]) & 1u)) ? ((unsigned long long)mono_u32(sel[j]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(lane + 64 * j)) : 0ull;
    int my_e;
    float my_w;
    pick_top_k(key, sc.s, k, lane, my_e, my_w);             // the weight is the score WITHOUT the bias
    float sum_e = 1.f;
    if (renormalize) {
        // the reference's order: a shuffle-down tree (16, 8, 4, 2, 1) over lanes holding the k weights and zeros -- lane 0 of the
        // same tree here (every lane it reads lies below 32)
        float t = lane < k ? my_w : 0.f;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        sum_e = __shfl(t, 0, 64) + 1e-20f;
    }
    if (lane < k) {
        float w = my_w / sum_e;
        asm volatile("" : "+v"(w));
        out_v[lane] = w * weight_scale;
        out_idx[lane] = my_e;
        if (worker_load) atomicAdd(&worker_load[my_e % num_worker], 1);
        if (expert_load) atomicAdd(&expert_load[my_e], 1);
    } else if (lane < top_k_ext) {
        out_v[lane] = 1.f;
        out_idx[lane] = 0;
    }
}

}  // namespace

extern "C" {

int zl_moe_top_k_softmax(const uint16_t* logits, int64_t tokens, int num_exp, int top_k, int top_k_ext, int renormalize, float weight_scale,
                         int scoring, int dtype, float* out_v, int32_t* out_idx, int32_t* worker_load, int32_t* expert_load, int num_worker,
                         zl_stream_t s) {
    ZL_CHECK_ARG(logits && out_v && out_idx && tokens > 0 && num_exp > 0 && top_k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(num_exp < 256 + 1 && top_k <= kMaxTopK && top_k <= num_exp && top_k_ext >= top_k && top_k_ext <= 64 && scoring >= 1 && scoring <= 3, ZL_ESHAPE);
    ZL_CHECK_ARG(!worker_load || num_worker > 0, ZL_EINVAL);
    ZL_CHECK_ARG(tokens < ((int64_t)1 << 31), ZL_ELIMIT);
    if (dtype == ZL_F16)
        hipLaunchKernelGGL(k_moe_top_k_softmax<ZL_F16>, dim3((unsigned)tokens), dim3(64), 0, (hipStream_t)s, logits, num_exp, top_k, out_v, out_idx,
                           renormalize, weight_scale, scoring, top_k_ext, worker_load, expert_load, num_worker);
    else if (dtype == ZL_BF16)
        hipLaunchKernelGGL(k_moe_top_k_softmax<ZL_BF16>, dim3((unsigned)tokens), dim3(64), 0, (hipStream_t)s, logits, num_exp, top_k, out_v, out_idx,
                           renormalize, weight_scale, scoring, top_k_ext, worker_load, expert_load, num_worker);
    else if (dtype == ZL_F32)
        hipLaunchKernelGGL(k_moe_top_k_softmax<ZL_F32>, dim3((unsigned)tokens), dim3(64), 0, (hipStream_t)s, logits, num_exp, top_k, out_v, out_idx,
                           renormalize, weight_scale, scoring, top_k_ext, worker_load, expert_load, num_worker);
    else
        return ZL_EDTYPE;
    return zl_launch_status();
}

int zl_moe_group_topk(const uint16_t* logits, const float* correction_bias, int64_t tokens, int num_exp, int top_k, int top_k_ext,
                      int renormalize, float weight_scale, int scoring, int num_group, int topk_group, int dtype, float* out_v,
                      int32_t* out_idx, int32_t* worker_load, int32_t* expert_load, int num_worker, zl_stream_t s) {
    ZL_CHECK_ARG(logits && out_v && out_idx && tokens > 0 && num_exp > 0 && top_k > 0 && num_group > 0 && topk_group > 0, ZL_EINVAL);
    ZL_CHECK_ARG(num_group <= 8 && topk_group <= num_group && num_exp % num_group == 0 && num_exp / num_group <= 32 && num_exp <= 256, ZL_ESHAPE);
    // (the reference pads a short merge with (-1e20, index -1) entries; with k experts in the kept groups there is no padding to pick)
    ZL_CHECK_ARG(top_k <= kMaxTopK && topk_group * top_k <= 32 && top_k <= topk_group * (num_exp / num_group), ZL_ESHAPE);
    ZL_CHECK_ARG(top_k_ext >= top_k && top_k_ext <= 64, ZL_ESHAPE);
    ZL_CHECK_ARG(scoring == SC_SOFTMAX || scoring == SC_SIGMOID, ZL_ESHAPE);
    ZL_CHECK_ARG(!worker_load || num_worker > 0, ZL_EINVAL);
    ZL_CHECK_ARG(tokens < ((int64_t)1 << 31), ZL_ELIMIT);
    const dim3 grid((unsigned)tokens), block(64);
    if (dtype == ZL_F16)
        hipLaunchKernelGGL(k_moe_group_topk<ZL_F16>, grid, block, 0, (hipStream_t)s, logits, correction_bias, num_exp, top_k, out_v, out_idx,
                           renormalize, weight_scale, scoring, num_group, topk_group, num_exp / num_group, top_k_ext, worker_load, expert_load,
                           num_worker);
    else if (dtype == ZL_BF16)
        hipLaunchKernelGGL(k_moe_group_topk<ZL_BF16>, grid, block, 0, (hipStream_t)s, logits, correction_bias, num_exp, top_k, out_v, out_idx,
                           renormalize, weight_scale, scoring, num_group, topk_group, num_exp / num_group, top_k_ext, worker_load, expert_load,
                           num_worker);
    else if (dtype == ZL_F32)
        hipLaunchKernelGGL(k_moe_group_topk<ZL_F32>, grid, block, 0, (hipStream_t)s, logits, correction_bias, num_exp, top_k, out_v, out_idx,
                           renormalize, weight_scale, scoring, num_group, topk_group, num_exp / num_group, top_k_ext, worker_load, expert_load,
                           num_worker);
    else
        return ZL_EDTYPE;
    return zl_launch_status();
}

}  // extern "C"

// ---- dispatch / combine of the prompt-side MoE path (src/nn/feedforward/ff_kernel.cu:518-1082) -----------------------------------
// Index bookkeeping between the router and the grouped GEMMs, and the weighted sum of the expert outputs.  Integer outputs
// are exact; the sums accumulate in fp32 in slot order with one fused multiply-add per term (what nvcc makes of
// `acc += float(x) * w`) and round once to T: bit-identical to the oracle.
namespace {

// KERNEL_sum_experts (:520-538): out[q, d] = T(sum_i float(input[index[q K + i], d]) * weight[q K + i])
template <int DT>
__global__ __launch_bounds__(256) void k_moe_sum_experts(int dim_model, int K, const uint16_t* __restrict__ input, const int32_t* __restrict__ index,
                                                         const float* __restrict__ weight, uint16_t* __restrict__ out) {
    const int q = blockIdx.x, d = blockIdx.y * blockDim.x + threadIdx.x;
    if (d >= dim_model) return;
    float acc = 0.f;
    for (int i = 0; i < K; ++i)
        acc = __builtin_fmaf(ZT<DT>::to_f32(input[(size_t)index[q * K + i] * dim_model + d]), weight[q * K + i], acc);
    out[(size_t)q * dim_model + d] = ZT<DT>::from_f32(acc);
}

// KERNEL_sum_experts_arr / _inline_arr (:541-626): one input matrix per expert; a single token (grid.x == 1) reads row 0 of each
// and weight[k]; with expert parallelism only the experts of this rank (exp & (world - 1)) == rank contribute
template <int DT>
__global__ __launch_bounds__(256) void k_moe_sum_experts_arr(int dim_model, int K, const uint16_t* const* __restrict__ input_arr,
                                                             const int32_t* __restrict__ experts, const int32_t* __restrict__ index,
                                                             const float* __restrict__ weight, uint16_t* __restrict__ out, int exp_parallel,
                                                             int world_size_mask, int local_rank) {
    const int q = blockIdx.x, d = blockIdx.y * blockDim.x + threadIdx.x;
    if (d >= dim_model) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int i = q * K + k, e = experts[i];
        if (exp_parallel && ((e & world_size_mask) != local_rank)) continue;
        const uint16_t* input = input_arr[e];
        if (gridDim.x == 1) acc = __builtin_fmaf(ZT<DT>::to_f32(input[d]), weight[k], acc);
        else acc = __builtin_fmaf(ZT<DT>::to_f32(input[(size_t)index[i] * dim_model + d]), weight[i], acc);
    }
    out[(size_t)q * dim_model + d] = ZT<DT>::from_f32(acc);
}

// KERNEL_route_shared_lb (:797-832): shared-expert slot s of token q goes to the first rank with spare capacity, ranks filled in order
__global__ void k_moe_route_shared_lb(int32_t* exp_ids, const int32_t* worker_load_base, int32_t* worker_load, int32_t* expert_load, int max_load,
                                      int world_size, int seq_len, int top_k, int top_k_ext, int num_local_experts) {
    const int q = blockIdx.x, s = blockIdx.y;
    int r = 0, skip_len = seq_len * s + q;
    for (;;) {
        const int base = worker_load_base[r];
        const int cap = base >= max_load ? 0 : max_load - base;
        if (skip_len < cap || r == world_size - 1) break;      // (the reference asserts r < world_size)
        skip_len -= cap;
        ++r;
    }
    const int exp_id = (num_local_experts + s) * world_size + r;   // a pseudo expert id of rank r
    exp_ids[q * top_k_ext + top_k + s] = exp_id;
    atomicAdd(&worker_load[r], 1);
    atomicAdd(&expert_load[exp_id], 1);
}

__global__ void k_moe_plus_for_sort(const int32_t* exp_ids, int32_t* out, int multiple, int world_size, int numel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < numel) out[i] = exp_ids[i] + (exp_ids[i] % world_size) * multiple;
}

// KERNEL_calc_reverse_idx (:886-898): position of every (token, slot) inside its expert's run of the sorted order
__global__ void k_moe_calc_reverse_idx(const int32_t* exp_ids, const int32_t* indices, const int32_t* expert_offsets, int32_t* rev_indices,
                                       int numel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numel) return;
    const int idx = indices[i];
    rev_indices[idx] = i - expert_offsets[exp_ids[idx]];
}

// KERNEL_fill_m_indices_padded_indices_temp (:946-962): rows of the m-grouped contiguous layout (every expert's run padded to block_m)
__global__ void k_moe_fill_m_indices(const int32_t* num_tokens, const int32_t* offsets, const int32_t* aligned_offsets, int32_t* padded_indices,
                                     int32_t* m_indices) {
    const int e = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < num_tokens[e]) padded_indices[offsets[e] + s] = aligned_offsets[e] + s;
    const int s2 = aligned_offsets[e] + s;
    if (s2 < aligned_offsets[e + 1]) m_indices[s2] = e;
}

}  // namespace

extern "C" {

int zl_moe_sum_experts(const uint16_t* input, const int32_t* index, const float* weight, uint16_t* out, int64_t seq_len, int top_k,
                       int64_t dim_model, int dtype, zl_stream_t s) {
    ZL_CHECK_ARG(input && index && weight && out && seq_len > 0 && top_k > 0 && dim_model > 0, ZL_EINVAL);
    ZL_CHECK_ARG(top_k <= kMaxTopK && seq_len < ((int64_t)1 << 31) && (dim_model + 255) / 256 <= 65535, ZL_ELIMIT);
    const dim3 grid((unsigned)seq_len, (unsigned)((dim_model + 255) / 256));
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_moe_sum_experts<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, (int)dim_model, top_k, input, index, weight, out);
    else if (dtype == ZL_BF16) hipLaunchKernelGGL(k_moe_sum_experts<ZL_BF16>, grid, dim3(256), 0, (hipStream_t)s, (int)dim_model, top_k, input, index, weight, out);
    else return ZL_EDTYPE;
    return zl_launch_status();
}

int zl_moe_sum_experts_arr(const uint16_t* const* input_arr, const int32_t* experts, const int32_t* index, const float* weight, uint16_t* out,
                           int64_t seq_len, int top_k, int64_t dim_model, int exp_parallel, int world_size, int local_rank, int dtype,
                           zl_stream_t s) {
    ZL_CHECK_ARG(input_arr && experts && weight && out && seq_len > 0 && top_k > 0 && dim_model > 0, ZL_EINVAL);
    ZL_CHECK_ARG(seq_len == 1 || index, ZL_EINVAL);
    ZL_CHECK_ARG(!exp_parallel || (world_size > 0 && (world_size & (world_size - 1)) == 0), ZL_ESHAPE);     // the reference masks with world_size - 1
    ZL_CHECK_ARG(top_k <= kMaxTopK && seq_len < ((int64_t)1 << 31) && (dim_model + 255) / 256 <= 65535, ZL_ELIMIT);
    const dim3 grid((unsigned)seq_len, (unsigned)((dim_model + 255) / 256));
    if (dtype == ZL_F16)
        hipLaunchKernelGGL(k_moe_sum_experts_arr<ZL_F16>, grid, dim3(256), 0, (hipStream_t)s, (int)dim_model, top_k, input_arr, experts, index, weight,
                           out, exp_parallel, world_size - 1, local_rank);
    else if (dtype == ZL_BF16)
        hipLaunchKernelGGL(k_moe_sum_experts_arr<ZL_BF16>, grid, dim3(256), 0, (hipStream_t)s, (int)dim_model, top_k, input_arr, experts, index, weight,
                           out, exp_parallel, world_size - 1, local_rank);
    else return ZL_EDTYPE;
    return zl_launch_status();
}

int zl_moe_route_shared_lb(int32_t* exp_ids, const int32_t* worker_load_base, int32_t* worker_load, int32_t* expert_load, int max_load,
                           int world_size, int64_t seq_len, int top_k, int top_k_ext, int num_local_experts, zl_stream_t s) {
    ZL_CHECK_ARG(exp_ids && worker_load_base && worker_load && expert_load && seq_len > 0 && world_size > 0, ZL_EINVAL);
    ZL_CHECK_ARG(top_k_ext > top_k && top_k >= 0 && top_k_ext - top_k <= 65535 && seq_len < ((int64_t)1 << 31), ZL_ESHAPE);
    hipLaunchKernelGGL(k_moe_route_shared_lb, dim3((unsigned)seq_len, (unsigned)(top_k_ext - top_k)), dim3(1), 0, (hipStream_t)s, exp_ids,
                       worker_load_base, worker_load, expert_load, max_load, world_size, (int)seq_len, top_k, top_k_ext, num_local_experts);
    return zl_launch_status();
}

int zl_moe_plus_for_sort(const int32_t* exp_ids, int32_t* out, int multiple, int world_size, int64_t numel, zl_stream_t s) {
    ZL_CHECK_ARG(exp_ids && out && numel > 0 && world_size > 0 && numel < ((int64_t)1 << 31), ZL_EINVAL);
    hipLaunchKernelGGL(k_moe_plus_for_sort, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)s, exp_ids, out, multiple, world_size,
                       (int)numel);
    return zl_launch_status();
}

int zl_moe_calc_reverse_idx(const int32_t* exp_ids, const int32_t* indices, const int32_t* expert_offsets, int32_t* rev_indices, int64_t numel,
                            zl_stream_t s) {
    ZL_CHECK_ARG(exp_ids && indices && expert_offsets && rev_indices && numel > 0 && numel < ((int64_t)1 << 31), ZL_EINVAL);
    hipLaunchKernelGGL(k_moe_calc_reverse_idx, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)s, exp_ids, indices, expert_offsets,
                       rev_indices, (int)numel);
    return zl_launch_status();
}

int zl_moe_fill_m_indices(const int32_t* num_tokens, const int32_t* offsets, const int32_t* aligned_offsets, int32_t* padded_indices,
                          int32_t* m_indices, int local_experts, int max_num_token, int block_m, zl_stream_t s) {
    ZL_CHECK_ARG(num_tokens && offsets && aligned_offsets && padded_indices && m_indices && local_experts > 0 && block_m > 0, ZL_EINVAL);
    ZL_CHECK_ARG(local_experts <= 65535 && block_m <= 1024, ZL_ELIMIT);
    if (max_num_token <= 0) return ZL_OK;
    hipLaunchKernelGGL(k_moe_fill_m_indices, dim3((unsigned)((max_num_token + block_m - 1) / block_m), (unsigned)local_experts), dim3((unsigned)block_m),
                       0, (hipStream_t)s, num_tokens, offsets, aligned_offsets, padded_indices, m_indices);
    return zl_launch_status();
}

}  // extern "C"

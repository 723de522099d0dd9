// moe_router.hip -- f4, first part: the MoE router of config 5 (DeepSeek-V3 / Qwen3-MoE style top-k gating):
//   zl_moe_top_k_softmax   nn::top_k_softmax      / KERNEL_top_k_softmax (src/nn/feedforward/ff_kernel.cu:174-268)
//   zl_moe_group_topk      nn::group_topk_softmax / KERNEL_group_topk    (ff_kernel.cu:296-515)
// Both restate the reference's arithmetic AND its selection rules -- reduction trees of 32 lanes (shuffle-down, lane 0
// broadcast), the insertion sort's "a later equal value does not displace an earlier one", the bitonic sort's "equal
// values: smaller expert index first", the un-biased output weight of the biased group selection, weight 1 in the
// extended slots -- so that expert ids match the oracle exactly whenever the scores do (device expf vs glibc expf is the
// only difference: <= 2 ulp on a score).  One quirk is kept on purpose because results must equal the reference's:
// top_k_softmax's "sigmoid" is 1 / (1 + expf(+x)) as written in DEV_route_score (ff_kernel.cu:158-160); group_topk's is the
// usual 1 / (1 + expf(-x)).  A 32-lane "warp" of the reference is the lower or upper half of a wavefront here
// (__shfl_* with width 32).  Tiny kernels: a decode step routes a handful of tokens; latency, not throughput.
#include "zl_common.h"

namespace {

constexpr int kMaxTopK = 16;
enum { SC_SOFTMAX = 1, SC_SIGMOID = 2, SC_LINEAR = 3 };

__device__ __forceinline__ float warp32_sum_b(float x) {     // warpReduceSumB (reduce.cuh:48-54)
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) x += __shfl_down(x, off, 32);
    return __shfl(x, 0, 32);
}
__device__ __forceinline__ float warp32_max_b(float x) {     // warpReduceMaxB (:21-27)
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
        const float y = __shfl_down(x, off, 32);
        x = x > y ? x : y;
    }
    return __shfl(x, 0, 32);
}
// blockReduce{Max,Sum} (:57-71, 93-107): per-warp tree, then warp 0 over the per-warp results (padded with `pad`)
template <bool MAX>
__device__ __forceinline__ float block32_reduce(float x, float* shared /* 33 */, float pad) {
    const int lane = threadIdx.x & 31, wid = threadIdx.x >> 5;
    x = MAX ? warp32_max_b(x) : warp32_sum_b(x);
    if (lane == 0) shared[wid] = x;
    __syncthreads();
    x = ((int)threadIdx.x < (int)blockDim.x / 32) ? shared[lane] : pad;
    if (wid == 0) {
        x = MAX ? warp32_max_b(x) : warp32_sum_b(x);
        if (lane == 0) shared[32] = x;
    }
    __syncthreads();
    const float r = shared[32];
    __syncthreads();
    return r;
}

// DEV_softmax_inplace (ff_kernel.cu:126-150) over `data[0..n)` with all blockDim.x threads
__device__ __forceinline__ void softmax_inplace(float* data, int n, float* shared33) {
    float local_max = -1e20f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) local_max = fmaxf(local_max, data[i]);
    local_max = blockDim.x > 32 ? block32_reduce<true>(local_max, shared33, -INFINITY) : warp32_max_b(local_max);
    float local_sum = 1e-20f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        data[i] = expf(data[i] - local_max);
        local_sum += data[i];
    }
    local_sum = blockDim.x > 32 ? block32_reduce<false>(local_sum, shared33, 0.f) : warp32_sum_b(local_sum);
    for (int i = threadIdx.x; i < n; i += blockDim.x) data[i] /= local_sum;
    __syncthreads();
}

// grid (tokens), 32 threads
template <int DT>
__global__ __launch_bounds__(32) void k_moe_top_k_softmax(const uint16_t* __restrict__ logits, int num_exp, int k, float* __restrict__ out_v,
                                                          int32_t* __restrict__ out_idx, int renormalize, float weight_scale, int scoring,
                                                          int top_k_ext, int32_t* worker_load, int32_t* expert_load, int num_worker) {
    __shared__ float data[256];
    __shared__ float shared33[33];
    const int q = blockIdx.x;
    logits += (size_t)q * num_exp;
    out_v += (size_t)q * top_k_ext;
    out_idx += (size_t)q * top_k_ext;
    for (int i = threadIdx.x; i < num_exp; i += blockDim.x) data[i] = ZT<DT>::to_f32(logits[i]);
    if (scoring == SC_SOFTMAX) {
        softmax_inplace(data, num_exp, shared33);
    } else if (scoring == SC_SIGMOID) {
        for (int i = threadIdx.x; i < num_exp; i += blockDim.x) data[i] = 1.f / (1.f + expf(data[i]));   // (sic)
    }
    __syncthreads();
    if (threadIdx.x > 0) return;
    float value[kMaxTopK + 1];
    int idx[kMaxTopK + 1];
    for (int i = 0; i < k; ++i) { value[i] = -1e20f; idx[i] = 0; }
    for (int j = 0; j < num_exp; ++j) {              // DEV_insert_sort_topk (:98-124)
        const float v = data[j];
        int i;
        for (i = k - 1; i >= 0; --i) {
            if (v > value[i]) {
                value[i + 1] = value[i];
                idx[i + 1] = idx[i];
            } else {
                value[i + 1] = v;
                idx[i + 1] = j;
                break;
            }
        }
        if (i < 0) { value[0] = v; idx[0] = j; }
    }
    float sum_e = 1.f;
    if (renormalize) {
        sum_e = 1.e-20f;
        for (int i = 0; i < k; ++i) sum_e += value[i];
    }
    for (int i = 0; i < k; ++i) {
        float w = value[i] / sum_e;
        asm volatile("" : "+v"(w));
        out_v[i] = w * weight_scale;
        out_idx[i] = idx[i];
        if (worker_load) atomicAdd(&worker_load[idx[i] % num_worker], 1);
        if (expert_load) atomicAdd(&expert_load[idx[i]], 1);
    }
    for (int i = k; i < top_k_ext; ++i) out_v[i] = 1.f;
}

// warpBitonicSort<T, N> (:273-291): descending; equal values: smaller position first
template <int N>
__device__ __forceinline__ void bitonic_desc(float& v1, int& pos) {
    const int lane_id = threadIdx.x & (N - 1);
#pragma unroll
    for (int k = 2; k <= N; k *= 2) {
        const bool desc = (lane_id & k) == 0;
#pragma unroll
        for (int j = k / 2; j > 0; j /= 2) {
            const float v2 = __shfl_xor(v1, j, 32);
            const int pos2 = __shfl_xor(pos, j, 32);
            const bool upper = (lane_id & j) != 0;
            if (desc ^ (v1 > v2 || (v1 == v2 && pos < pos2)) ^ upper) {
                v1 = v2;
                pos = pos2;
            }
        }
    }
}

// grid (tokens), num_group * 32 threads
template <int DT>
__global__ __launch_bounds__(256) void k_moe_group_topk(const uint16_t* __restrict__ logits, const float* __restrict__ correction_bias, int num_exp,
                                                        int k, float* __restrict__ out_v, int32_t* __restrict__ out_idx, int renormalize,
                                                        float weight_scale, int scoring, int num_group, int topk_group, int num_in_group,
                                                        int top_k_ext, int32_t* worker_load, int32_t* expert_load, int num_worker) {
    __shared__ float data[512];
    __shared__ float shared33[33];
    __shared__ float shared_val[32];
    __shared__ int shared_pos[32], shared_gid[32], group_ranks[32];
    const int q = blockIdx.x, g = threadIdx.x / 32, lane_id = threadIdx.x % 32;
    logits += (size_t)q * num_exp;
    out_v += (size_t)q * top_k_ext;
    out_idx += (size_t)q * top_k_ext;
    if ((int)threadIdx.x < 32) { shared_val[threadIdx.x] = 0.f; shared_pos[threadIdx.x] = 0; shared_gid[threadIdx.x] = 0; group_ranks[threadIdx.x] = -1; }
    __syncthreads();
    if (scoring == SC_SIGMOID) {
        if ((int)threadIdx.x < num_exp) data[threadIdx.x] = 1.f / (1.f + expf(-ZT<DT>::to_f32(logits[threadIdx.x])));
        __syncthreads();
    } else {
        if ((int)threadIdx.x < num_exp) data[threadIdx.x] = ZT<DT>::to_f32(logits[threadIdx.x]);
        __syncthreads();
        softmax_inplace(data, num_exp, shared33);
    }
    // the k best of every group
    int exp_id = -1;
    float score = -1e20f;
    if (lane_id < num_in_group) {
        exp_id = g * num_in_group + lane_id;
        score = data[exp_id];
        if (correction_bias) score += correction_bias[exp_id];       // for the selection only
    }
    bitonic_desc<32>(score, exp_id);
    if (lane_id == 0) {
        shared_val[g] = score;
        shared_gid[g] = g;
    }
    __syncthreads();
    // the topk_group best groups
    if ((int)threadIdx.x < 32) {
        float group_score = lane_id < num_group ? shared_val[lane_id] : -1e20f;
        int group_id = shared_gid[lane_id];
        bitonic_desc<8>(group_score, group_id);
        if ((int)threadIdx.x < topk_group) group_ranks[group_id] = threadIdx.x;
    }
    __syncthreads();
    const int group_rank = group_ranks[g];
    __syncthreads();                                   // (shared_val is reused: every group score has been read)
    if (group_rank >= 0 && lane_id < k) {
        shared_val[group_rank * k + lane_id] = score;
        shared_pos[group_rank * k + lane_id] = exp_id;
    }
    __syncthreads();
    if ((int)threadIdx.x >= 32) return;
    score = lane_id < topk_group * k ? shared_val[lane_id] : -1e20f;
    exp_id = shared_pos[lane_id];
    bitonic_desc<32>(score, exp_id);
    if ((int)threadIdx.x < k && correction_bias) score = data[exp_id];      // the weight is the original score
    float sum_e = 1.f;
    if (renormalize) sum_e = warp32_sum_b((int)threadIdx.x < k ? score : 0.f) + 1e-20f;
    if ((int)threadIdx.x < k) {
        float w = score / sum_e;
        asm volatile("" : "+v"(w));
        out_v[lane_id] = w * weight_scale;
        out_idx[lane_id] = exp_id;
        if (worker_load) atomicAdd(&worker_load[exp_id % num_worker], 1);
        if (expert_load) atomicAdd(&expert_load[exp_id], 1);
    }
    if ((int)threadIdx.x < top_k_ext - k) {
        out_v[k + threadIdx.x] = 1.f;
        out_idx[k + threadIdx.x] = 0;
    }
}

}  // namespace

extern "C" {

int zl_moe_top_k_softmax(const uint16_t* logits, int64_t tokens, int num_exp, int top_k, int top_k_ext, int renormalize, float weight_scale,
                         int scoring, int dtype, float* out_v, int32_t* out_idx, int32_t* worker_load, int32_t* expert_load, int num_worker,
                         zl_stream_t s) {
    ZL_CHECK_ARG(logits && out_v && out_idx && tokens > 0 && num_exp > 0 && top_k > 0, ZL_EINVAL);
    ZL_CHECK_ARG(num_exp < 256 + 1 && top_k <= kMaxTopK && top_k <= num_exp && top_k_ext >= top_k && scoring >= 1 && scoring <= 3, ZL_ESHAPE);
    ZL_CHECK_ARG(!worker_load || num_worker > 0, ZL_EINVAL);
    ZL_CHECK_ARG(tokens < ((int64_t)1 << 31), ZL_ELIMIT);
    if (dtype == ZL_F16)
        hipLaunchKernelGGL(k_moe_top_k_softmax<ZL_F16>, dim3((unsigned)tokens), dim3(32), 0, (hipStream_t)s, logits, num_exp, top_k, out_v, out_idx,
                           renormalize, weight_scale, scoring, top_k_ext, worker_load, expert_load, num_worker);
    else if (dtype == ZL_BF16)
        hipLaunchKernelGGL(k_moe_top_k_softmax<ZL_BF16>, dim3((unsigned)tokens), dim3(32), 0, (hipStream_t)s, logits, num_exp, top_k, out_v, out_idx,
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
    ZL_CHECK_ARG(top_k <= kMaxTopK && topk_group * top_k <= 32 && top_k_ext >= top_k && top_k_ext - top_k <= 32, ZL_ESHAPE);
    ZL_CHECK_ARG(scoring == SC_SOFTMAX || scoring == SC_SIGMOID, ZL_ESHAPE);
    ZL_CHECK_ARG(!worker_load || num_worker > 0, ZL_EINVAL);
    ZL_CHECK_ARG(tokens < ((int64_t)1 << 31), ZL_ELIMIT);
    const dim3 grid((unsigned)tokens), block((unsigned)(num_group * 32));
    if (dtype == ZL_F16)
        hipLaunchKernelGGL(k_moe_group_topk<ZL_F16>, grid, block, 0, (hipStream_t)s, logits, correction_bias, num_exp, top_k, out_v, out_idx,
                           renormalize, weight_scale, scoring, num_group, topk_group, num_exp / num_group, top_k_ext, worker_load, expert_load,
                           num_worker);
    else if (dtype == ZL_BF16)
        hipLaunchKernelGGL(k_moe_group_topk<ZL_BF16>, grid, block, 0, (hipStream_t)s, logits, correction_bias, num_exp, top_k, out_v, out_idx,
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

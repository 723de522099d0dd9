// comm.hip -- the exchange step of the tensor-parallel decode path (include/zhilight_amd_comm.h): a direct RCCL
// communicator per GPU and a one-shot peer-read all-reduce for decode-size messages.  Built into
// libzhilight_amd_comm.so (links librccl); nothing here is needed on one GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include "../../include/zhilight_amd.h"
#include "../../include/zhilight_amd_comm.h"

#define ZL_CHECK_ARG(cond, code) \
    do {                         \
        if (!(cond)) return (code); \
    } while (0)
#define ZL_HIP(expr)                        \
    do {                                    \
        hipError_t e_ = (expr);             \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)
#define ZL_NCCL(expr)                                  \
    do {                                               \
        ncclResult_t r_ = (expr);                      \
        if (r_ != ncclSuccess) return 1000 + (int)r_;  \
    } while (0)

struct zl_comm {
    ncclComm_t comm;
    int rank, size;
};

static bool nccl_type(int dtype, ncclDataType_t* t) {
    switch (dtype) {
        case 0: *t = ncclFloat16; return true;
        case 1: *t = ncclBfloat16; return true;
        case 2: *t = ncclFloat32; return true;
        case 3: *t = ncclInt32; return true;
        case 4: *t = ncclInt8; return true;
        default: return false;
    }
}

namespace {

constexpr int kMaxChunks = 64;
constexpr unsigned kMaxPolls = 1u << 22;     // ~ seconds: a peer that never arrives ends in an error word, not a hang

// what a rank's SHARED buffer holds: [2 data slots of max_bytes][flags[rank p][chunk c], written by peer p]
struct ArState {                               // device-resident, private to the rank
    uint64_t buf[ZL_AR_MAX_RANKS];             // base address of every rank's shared buffer as mapped HERE
    int64_t max_bytes;
    int world, rank;
    unsigned err;
    unsigned epoch[kMaxChunks];                // the message sequence number, one copy per workgroup index (device-side:
                                               // survives graph replay); all copies advance with every launch
};

__device__ __forceinline__ uint16_t* slot_of(const ArState* st, int r, unsigned parity) {
    return reinterpret_cast<uint16_t*>(st->buf[r] + (uint64_t)parity * (uint64_t)st->max_bytes);
}
__device__ __forceinline__ unsigned* flags_of(const ArState* st, int r) {   // table inside rank r's buffer
    return reinterpret_cast<unsigned*>(st->buf[r] + 2ull * (uint64_t)st->max_bytes);
}

template <int DT> __device__ __forceinline__ float to_f32(uint16_t v);
template <> __device__ __forceinline__ float to_f32<0>(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
template <> __device__ __forceinline__ float to_f32<1>(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
template <int DT> __device__ __forceinline__ uint16_t from_f32(float f);
template <> __device__ __forceinline__ uint16_t from_f32<0>(float f) {
    asm volatile("" : "+v"(f));
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}
template <> __device__ __forceinline__ uint16_t from_f32<1>(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// One launch = one all-reduce.  Workgroup c owns element chunk c on EVERY rank, so chunk c only ever waits for the peers'
// chunk c: no grid-wide step.  Two data slots by the parity of the MESSAGE number e (the same for every chunk of a message,
// whatever its size): a rank can be at most one message ahead of a peer (it needs the peer's flags of message e to finish
// e), so slot e & 1 is never overwritten while a peer still reads it -- also when consecutive messages cut the slot into
// different chunks.  A flag holds the number of the last message its chunk index was published for: a waiter accepts any
// value >= e (a peer that is one message ahead has already overwritten e with e + 1; its slot e & 1 is still intact).
template <int DT>
__global__ __launch_bounds__(256) void k_ar(ArState* st, const uint16_t* __restrict__ x, const uint16_t* __restrict__ residual,
                                            uint16_t* __restrict__ out, int64_t n, int64_t per) {
    __shared__ unsigned s_epoch, s_timeout;
    const int c = blockIdx.x, world = st->world, rank = st->rank;
    if (threadIdx.x == 0) s_timeout = 0;
    if (n * 2 > st->max_bytes) {                         // message larger than the slots: refuse (error word), touch nothing
        if (threadIdx.x == 0) atomicAdd(&st->err, 1u);
        return;
    }
    if (threadIdx.x == 0) s_epoch = st->epoch[c] + 1u;
    __syncthreads();
    const unsigned e = s_epoch, parity = e & 1u;
    const int64_t i0 = (int64_t)c * per, i1 = i0 + per < n ? i0 + per : n;
    // ---- publish my rows of this chunk (8 halfs = 16 bytes per step)
    uint16_t* mine = slot_of(st, rank, parity);
    for (int64_t i = i0 + (int64_t)threadIdx.x * 8; i < i1; i += 256 * 8)
        *reinterpret_cast<uint4*>(mine + i) = *reinterpret_cast<const uint4*>(x + i);
    __atomic_thread_fence(__ATOMIC_RELEASE);            // system scope: the rows are in memory before any flag
    // (ROCm 7.2 may drop the s_waitcnt vmcnt(0) behind buffer_wbl2 when it can prove this wave's scoreboard empty; inline asm is
    //  invisible to that pass -- MI355X_MICROARCH.md, "compiler hazard")
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)threadIdx.x < world && (int)threadIdx.x != rank)
        __hip_atomic_store(flags_of(st, (int)threadIdx.x) + rank * kMaxChunks + c, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- wait for the peers' chunk (their flag lands in MY buffer: local polls)
    if ((int)threadIdx.x < world && (int)threadIdx.x != rank) {
        const unsigned* f = flags_of(st, rank) + (int)threadIdx.x * kMaxChunks + c;
        unsigned polls = 0;
        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > kMaxPolls) {
                atomicAdd(&st->err, 1u);
                s_timeout = 1;                          // benign race: every writer stores 1
                break;
            }
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);            // system scope: nothing cached from an earlier message
    // ---- reduce in rank order (fp32), one rounding to T, then the residual add in T arithmetic
    for (int64_t i = i0 + (int64_t)threadIdx.x * 8; i < i1; i += 256 * 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r) {
            uint64_t lo, hi;
            if (r == rank) {
                const uint4 v = *reinterpret_cast<const uint4*>(x + i);
                lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
                hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
            } else {       // a peer's rows: system-scope loads (never served from a stale cache line of message e - 2)
                const uint64_t* src = reinterpret_cast<const uint64_t*>(slot_of(st, r, parity) + i);
                lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] += to_f32<DT>((uint16_t)(lo >> (16 * j)));
                acc[4 + j] += to_f32<DT>((uint16_t)(hi >> (16 * j)));
            }
        }
        uint16_t o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = from_f32<DT>(acc[j]);
        if (s_timeout) {
            // a peer never published this chunk within the bounded wait: the sum would be silently wrong (and the ranks would
            // diverge) -- poison it instead, so that it cannot pass unnoticed (NaN propagates into the logits); zl_ar_status
            // reports the count
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = DT == ZL_F16 ? (uint16_t)0x7e00 : (uint16_t)0x7fc0;
        }
        if (residual) {
            const uint4 rv = *reinterpret_cast<const uint4*>(residual + i);
            const uint32_t ru[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = from_f32<DT>(to_f32<DT>((uint16_t)(ru[j / 2] >> (16 * (j & 1)))) + to_f32<DT>(o[j]));
        }
        *reinterpret_cast<uint4*>(out + i) = make_uint4(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16),
                                                       o[4] | ((uint32_t)o[5] << 16), o[6] | ((uint32_t)o[7] << 16));
    }
    if (threadIdx.x == 0) st->epoch[c] = e;
    // the copies of the workgroup indices this message did not use (nobody reads them in this launch)
    if (c == 0 && (int)threadIdx.x >= (int)gridDim.x && threadIdx.x < kMaxChunks) st->epoch[threadIdx.x] = e;
}

// ---- the same exchange with the rows travelling as group-32 INT8 codes + T scales: ModelContext::reduce_tp_int8
// (src/model/model_context.cpp:244-326, src/nn/quant/int8/quant_reduce_kernel.cu:13-330) as ONE launch.  The reference moves
// 1.06 bytes per value and hop instead of 2 in five steps (quantise all slices; all-to-all of the codes; dequantise + sum +
// re-quantise the rank's own slice; all-gather of the re-quantised slices; dequantise); round 3 composed those from three
// kernels and two RCCL send / recv rounds.  Here the steps are phases of k_ar's flag protocol:
//   phase 1  every rank quantises its whole vector (quant_group_32: amax of the 32 T values, codes rint(x 127 / amax), scale
//            T(amax / 127)) into ITS slot of the message's first parity, flags e1;
//   phase 2  rank r reads the peers' codes + scales of SLICE r (n / W values: 1.06 (W - 1) n / W bytes over the links), adds its
//            own UNQUANTISED slice -- fp32 fma chain in rank-distance order, dequant_sum_quant_g32 -- re-quantises (amax of the
//            sums rounded to T) into its slot of the other parity, flags e2;
//   phase 3  every rank reads the W re-quantised slices (another 1.06 (W - 1) n / W bytes) and dequantises: T(q scale), then the
//            layer's residual add in T arithmetic (element_add_scale) if asked.
// Bit for bit the values of the five-step composition (tests/test_gpu_comm.py).  A message = two message numbers (e1, e1 + 1):
// the slot-parity argument of k_ar holds for both halves (a rank enters phase 1 of the next message only after every peer's e2
// flag, i.e. after every peer has finished reading its phase-1 rows).  Lane = 4 consecutive values (one dword of codes), 8 lanes
// = one group; workgroup c = the c-th block of groups of EVERY slice.
__device__ __forceinline__ float group8_max(float v) {          // all-reduce over the 8 lanes of a group
    v = fmaxf(v, __shfl_xor(v, 1, 8));
    v = fmaxf(v, __shfl_xor(v, 2, 8));
    v = fmaxf(v, __shfl_xor(v, 4, 8));
    return v;
}
__device__ __forceinline__ uint32_t pack_codes(const float (&v)[4], float amax) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = v[j] * 127.0f;
        asm volatile("" : "+v"(t));                               // the product is rounded to fp32 before the division
        const int q = amax > 0.f ? (int)nearbyintf(t / amax) : 0;
        w |= ((uint32_t)q & 0xffu) << (8 * j);
    }
    return w;
}

template <int DT>
__global__ __launch_bounds__(256) void k_ar_q8(ArState* st, const uint16_t* __restrict__ x, const uint16_t* __restrict__ residual,
                                               uint16_t* __restrict__ out, int64_t n, int64_t gper, int world_arg) {
    __shared__ unsigned s_epoch, s_timeout;
    const int c = blockIdx.x, world = st->world, rank = st->rank;
    if (threadIdx.x == 0) s_timeout = 0;
    // the host cut the grid for world_arg ranks (the caller's argument); the exchange geometry is the state's: a mismatch would pass
    // every host check and exchange the wrong regions silently -- it is an error like an oversized message
    if (n * 2 > st->max_bytes || world_arg != world) {
        if (threadIdx.x == 0) atomicAdd(&st->err, 1u);
        return;
    }
    if (threadIdx.x == 0) s_epoch = st->epoch[c] + 1u;
    __syncthreads();
    const unsigned e1 = s_epoch, e2 = e1 + 1u, pa = e1 & 1u, pb = e2 & 1u;
    const int64_t m = n / world / 32;                              // groups per slice
    const int64_t g0 = (int64_t)c * gper, g1 = g0 + gper < m ? g0 + gper : m;
    const int l8 = threadIdx.x & 7;
    const int64_t slice_codes = m * 32;                            // bytes of codes per slice
    auto wait_peers = [&](unsigned e) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if ((int)threadIdx.x < world && (int)threadIdx.x != rank) {
            __hip_atomic_store(flags_of(st, (int)threadIdx.x) + rank * kMaxChunks + c, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned* f = flags_of(st, rank) + (int)threadIdx.x * kMaxChunks + c;
            unsigned polls = 0;
            while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > kMaxPolls) {
                    atomicAdd(&st->err, 1u);
                    s_timeout = 1;
                    break;
                }
            }
        }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    };
    // ---- phase 1: quantise block c of every slice into my slot of parity pa: [codes n bytes][scales n / 32 halfs]
    {
        uint8_t* area = reinterpret_cast<uint8_t*>(slot_of(st, rank, pa));
        for (int s = 0; s < world; ++s) {
            for (int64_t g = g0 + (threadIdx.x >> 3); g < g1; g += 32) {
                const int64_t gi = (int64_t)s * m + g;
                const uint2 raw = *reinterpret_cast<const uint2*>(x + gi * 32 + l8 * 4);
                float v[4] = {to_f32<DT>((uint16_t)raw.x), to_f32<DT>((uint16_t)(raw.x >> 16)), to_f32<DT>((uint16_t)raw.y), to_f32<DT>((uint16_t)(raw.y >> 16))};
                const float amax = group8_max(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                *reinterpret_cast<uint32_t*>(area + gi * 32 + l8 * 4) = pack_codes(v, amax);
                if (l8 == 0) reinterpret_cast<uint16_t*>(area + n)[gi] = from_f32<DT>(amax / 127.0f);
            }
        }
    }
    wait_peers(e1);
    // ---- phase 2: my slice: own unquantised rows + the peers' codes (rank-distance order), re-quantised into parity pb:
    //      [codes n / W bytes][scales n / 32 / W halfs]
    {
        uint8_t* area = reinterpret_cast<uint8_t*>(slot_of(st, rank, pb));
        for (int64_t g = g0 + (threadIdx.x >> 3); g < g1; g += 32) {
            const int64_t gi = (int64_t)rank * m + g;
            const uint2 raw = *reinterpret_cast<const uint2*>(x + gi * 32 + l8 * 4);
            float sum[4] = {to_f32<DT>((uint16_t)raw.x), to_f32<DT>((uint16_t)(raw.x >> 16)), to_f32<DT>((uint16_t)raw.y), to_f32<DT>((uint16_t)(raw.y >> 16))};
            for (int i = 0; i < world - 1; ++i) {
                const int src = (rank + i + 1) % world;
                const uint8_t* peer = reinterpret_cast<const uint8_t*>(slot_of(st, src, pa));
                const uint32_t codes = __hip_atomic_load(reinterpret_cast<const uint32_t*>(peer + gi * 32 + l8 * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                // (a 16-bit scale: read the aligned dword that holds it)
                const uint32_t sw = __hip_atomic_load(reinterpret_cast<const uint32_t*>(peer + n) + (gi >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const float sc = to_f32<DT>((uint16_t)(sw >> (16 * (int)(gi & 1))));
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[j] = __builtin_fmaf((float)(int8_t)(codes >> (8 * j)), sc, sum[j]);
            }
            // warpReduceMaxB<T>(fabsf(sum)): the magnitude is rounded to T before the maximum
            float mag = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) mag = fmaxf(mag, to_f32<DT>(from_f32<DT>(fabsf(sum[j]))));
            const float amax = group8_max(mag);
            *reinterpret_cast<uint32_t*>(area + g * 32 + l8 * 4) = pack_codes(sum, amax);
            if (l8 == 0) reinterpret_cast<uint16_t*>(area + slice_codes)[g] = from_f32<DT>(amax / 127.0f);
        }
    }
    wait_peers(e2);
    // ---- phase 3: dequantise block c of every slice (+ residual)
    for (int s = 0; s < world; ++s) {
        const uint8_t* area = reinterpret_cast<const uint8_t*>(slot_of(st, s, pb));
        for (int64_t g = g0 + (threadIdx.x >> 3); g < g1; g += 32) {
            const uint32_t codes = __hip_atomic_load(reinterpret_cast<const uint32_t*>(area + g * 32 + l8 * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t sw = __hip_atomic_load(reinterpret_cast<const uint32_t*>(area + slice_codes) + (g >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const float sc = to_f32<DT>((uint16_t)(sw >> (16 * (int)(g & 1))));
            uint16_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pr = (float)(int8_t)(codes >> (8 * j)) * sc;
                o[j] = from_f32<DT>(pr);
                if (s_timeout) o[j] = DT == ZL_F16 ? (uint16_t)0x7e00 : (uint16_t)0x7fc0;       // an expired wait: poison, never a silent wrong sum
            }
            const int64_t at = ((int64_t)s * m + g) * 32 + l8 * 4;
            if (residual) {
                const uint2 rv = *reinterpret_cast<const uint2*>(residual + at);
                const uint16_t ru[4] = {(uint16_t)rv.x, (uint16_t)(rv.x >> 16), (uint16_t)rv.y, (uint16_t)(rv.y >> 16)};
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = from_f32<DT>(to_f32<DT>(ru[j]) + to_f32<DT>(o[j]));
            }
            *reinterpret_cast<uint2*>(out + at) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
        }
    }
    if (threadIdx.x == 0) st->epoch[c] = e2;
    if (c == 0 && (int)threadIdx.x >= (int)gridDim.x && threadIdx.x < kMaxChunks) st->epoch[threadIdx.x] = e2;
}

}  // namespace

extern "C" {

int zl_comm_unique_id(void* id) {
    ZL_CHECK_ARG(id, ZL_EINVAL);
    static_assert(sizeof(ncclUniqueId) <= ZL_COMM_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    ZL_NCCL(ncclGetUniqueId(&u));
    memset(id, 0, ZL_COMM_UNIQUE_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return ZL_OK;
}
int zl_comm_create(zl_comm_t** out, int world_size, int rank, const void* id) {
    ZL_CHECK_ARG(out && id && world_size > 0 && rank >= 0 && rank < world_size, ZL_EINVAL);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    zl_comm* c = new zl_comm;
    c->rank = rank;
    c->size = world_size;
    ncclResult_t r = ncclCommInitRank(&c->comm, world_size, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return 1000 + (int)r;
    }
    *out = c;
    return ZL_OK;
}
int zl_comm_destroy(zl_comm_t* comm) {
    if (!comm) return ZL_OK;
    ncclResult_t r = ncclCommDestroy(comm->comm);
    delete comm;
    return r == ncclSuccess ? ZL_OK : 1000 + (int)r;
}
int zl_comm_rank(const zl_comm_t* comm) { return comm ? comm->rank : ZL_EINVAL; }
int zl_comm_size(const zl_comm_t* comm) { return comm ? comm->size : ZL_EINVAL; }

int zl_comm_all_reduce_sum(zl_comm_t* comm, const void* send, void* recv, int64_t count, int dtype, zl_comm_stream_t s) {
    ncclDataType_t t;
    ZL_CHECK_ARG(comm && send && recv && count > 0, ZL_EINVAL);
    ZL_CHECK_ARG(nccl_type(dtype, &t), ZL_EDTYPE);
    ZL_NCCL(ncclAllReduce(send, recv, (size_t)count, t, ncclSum, comm->comm, (hipStream_t)s));
    return ZL_OK;
}
int zl_comm_all_gather(zl_comm_t* comm, const void* send, void* recv, int64_t count, int dtype, zl_comm_stream_t s) {
    ncclDataType_t t;
    ZL_CHECK_ARG(comm && send && recv && count > 0, ZL_EINVAL);
    ZL_CHECK_ARG(nccl_type(dtype, &t), ZL_EDTYPE);
    ZL_NCCL(ncclAllGather(send, recv, (size_t)count, t, comm->comm, (hipStream_t)s));
    return ZL_OK;
}
int zl_comm_reduce_scatter_sum(zl_comm_t* comm, const void* send, void* recv, int64_t count, int dtype, zl_comm_stream_t s) {
    ncclDataType_t t;
    ZL_CHECK_ARG(comm && send && recv && count > 0, ZL_EINVAL);
    ZL_CHECK_ARG(nccl_type(dtype, &t), ZL_EDTYPE);
    ZL_NCCL(ncclReduceScatter(send, recv, (size_t)count, t, ncclSum, comm->comm, (hipStream_t)s));
    return ZL_OK;
}
int zl_comm_broadcast(zl_comm_t* comm, void* buf, int64_t count, int dtype, int root, zl_comm_stream_t s) {
    ncclDataType_t t;
    ZL_CHECK_ARG(comm && buf && count > 0 && root >= 0 && root < comm->size, ZL_EINVAL);
    ZL_CHECK_ARG(nccl_type(dtype, &t), ZL_EDTYPE);
    ZL_NCCL(ncclBroadcast(buf, buf, (size_t)count, t, root, comm->comm, (hipStream_t)s));
    return ZL_OK;
}
int zl_comm_send(zl_comm_t* comm, const void* buf, int64_t count, int dtype, int peer, zl_comm_stream_t s) {
    ncclDataType_t t;
    ZL_CHECK_ARG(comm && buf && count > 0 && peer >= 0 && peer < comm->size, ZL_EINVAL);
    ZL_CHECK_ARG(nccl_type(dtype, &t), ZL_EDTYPE);
    ZL_NCCL(ncclSend(buf, (size_t)count, t, peer, comm->comm, (hipStream_t)s));
    return ZL_OK;
}
int zl_comm_recv(zl_comm_t* comm, void* buf, int64_t count, int dtype, int peer, zl_comm_stream_t s) {
    ncclDataType_t t;
    ZL_CHECK_ARG(comm && buf && count > 0 && peer >= 0 && peer < comm->size, ZL_EINVAL);
    ZL_CHECK_ARG(nccl_type(dtype, &t), ZL_EDTYPE);
    ZL_NCCL(ncclRecv(buf, (size_t)count, t, peer, comm->comm, (hipStream_t)s));
    return ZL_OK;
}
int zl_comm_group_start(void) {
    ZL_NCCL(ncclGroupStart());
    return ZL_OK;
}
int zl_comm_group_end(void) {
    ZL_NCCL(ncclGroupEnd());
    return ZL_OK;
}

// ---- one-shot all-reduce -----------------------------------------------------------------------------------------------
int64_t zl_ar_buffer_bytes(int64_t max_message_bytes) {
    if (max_message_bytes <= 0 || max_message_bytes % 16) return ZL_EINVAL;
    return 2 * max_message_bytes + (int64_t)ZL_AR_MAX_RANKS * kMaxChunks * (int64_t)sizeof(unsigned);
}
int64_t zl_ar_state_bytes(void) { return (int64_t)sizeof(ArState); }
// setup-time allocation of a buffer peers can map and read coherently (fine-grained device memory), zeroed
int zl_ar_alloc(int64_t bytes, void** out) {
    ZL_CHECK_ARG(out && bytes > 0, ZL_EINVAL);
    void* p = nullptr;
    // fine-grained (peer-coherent) memory or nothing: flags and rows in a coarse-grained allocation are not guaranteed to become
    // visible to the peers inside a launch
    ZL_HIP(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained));
    ZL_HIP(hipMemset(p, 0, (size_t)bytes));
    ZL_HIP(hipDeviceSynchronize());
    *out = p;
    return ZL_OK;
}
int zl_ar_free(void* p) {
    if (p) ZL_HIP(hipFree(p));
    return ZL_OK;
}
int zl_ar_export(void* buffer, void* handle) {
    ZL_CHECK_ARG(buffer && handle, ZL_EINVAL);
    static_assert(sizeof(hipIpcMemHandle_t) <= ZL_AR_IPC_HANDLE_BYTES, "handle size");
    hipIpcMemHandle_t h;
    ZL_HIP(hipIpcGetMemHandle(&h, buffer));
    memset(handle, 0, ZL_AR_IPC_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    return ZL_OK;
}
int zl_ar_open(const void* handle, void** mapped) {
    ZL_CHECK_ARG(handle && mapped, ZL_EINVAL);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    ZL_HIP(hipIpcOpenMemHandle(mapped, h, hipIpcMemLazyEnablePeerAccess));
    return ZL_OK;
}
int zl_ar_close(void* mapped) {
    ZL_CHECK_ARG(mapped, ZL_EINVAL);
    ZL_HIP(hipIpcCloseMemHandle(mapped));
    return ZL_OK;
}
int zl_ar_init(void* state, int world_size, int rank, void* const* buffers, int64_t max_message_bytes, zl_comm_stream_t s) {
    ZL_CHECK_ARG(state && buffers && world_size >= 1 && world_size <= ZL_AR_MAX_RANKS && rank >= 0 && rank < world_size, ZL_EINVAL);
    ZL_CHECK_ARG(max_message_bytes > 0 && max_message_bytes % 16 == 0, ZL_ESHAPE);
    ArState h;
    memset(&h, 0, sizeof(h));
    for (int r = 0; r < world_size; ++r) {
        ZL_CHECK_ARG(buffers[r], ZL_EINVAL);
        h.buf[r] = (uint64_t)(uintptr_t)buffers[r];
    }
    h.max_bytes = max_message_bytes;
    h.world = world_size;
    h.rank = rank;
    ZL_HIP(hipMemcpyAsync(state, &h, sizeof(h), hipMemcpyHostToDevice, (hipStream_t)s));
    ZL_HIP(hipStreamSynchronize((hipStream_t)s));    // h lives on this stack frame (setup path, not a launcher)
    return ZL_OK;
}
int zl_ar_all_reduce(void* state, const uint16_t* x, const uint16_t* residual, uint16_t* out, int64_t n, int dtype,
                     zl_comm_stream_t s) {
    ZL_CHECK_ARG(state && x && out && n > 0, ZL_EINVAL);
    ZL_CHECK_ARG(n % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && (!residual || ((uintptr_t)residual & 15) == 0),
                 ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    // chunking is a function of n only: every rank derives the same grid (>= 4 KB per workgroup, at most kMaxChunks)
    int64_t chunks = (n * 2 + 4095) / 4096;
    if (chunks > kMaxChunks) chunks = kMaxChunks;
    int64_t per = (n + chunks - 1) / chunks;
    per = (per + 7) / 8 * 8;
    chunks = (n + per - 1) / per;
    ArState* st = reinterpret_cast<ArState*>(state);
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_ar<0>, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)s, st, x, residual, out, n, per);
    else hipLaunchKernelGGL(k_ar<1>, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)s, st, x, residual, out, n, per);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ZL_OK : (int)e;
}
int zl_ar_all_reduce_int8(void* state, const uint16_t* x, const uint16_t* residual, uint16_t* out, int64_t n, int world_size, int dtype,
                          zl_comm_stream_t s) {
    ZL_CHECK_ARG(state && x && out && n > 0 && world_size >= 2 && world_size <= ZL_AR_MAX_RANKS, ZL_EINVAL);
    ZL_CHECK_ARG(n % (32 * (int64_t)world_size) == 0 && n % 64 == 0, ZL_ESHAPE);           // whole groups per slice; aligned scale rows
    ZL_CHECK_ARG((n / world_size / 32) % 2 == 0, ZL_ESHAPE);                                 // (scales are read as aligned dwords)
    ZL_CHECK_ARG(((uintptr_t)x & 7) == 0 && ((uintptr_t)out & 7) == 0 && (!residual || ((uintptr_t)residual & 7) == 0), ZL_ESHAPE);
    ZL_CHECK_ARG(dtype == ZL_F16 || dtype == ZL_BF16, ZL_EDTYPE);
    // chunking is a function of (n, world) only: >= 32 groups of every slice per workgroup, at most kMaxChunks
    const int64_t m = n / world_size / 32;
    int64_t chunks = (m + 31) / 32;
    if (chunks > kMaxChunks) chunks = kMaxChunks;
    int64_t gper = (m + chunks - 1) / chunks;
    gper = (gper + 1) / 2 * 2;
    chunks = (m + gper - 1) / gper;
    ArState* st = reinterpret_cast<ArState*>(state);
    if (dtype == ZL_F16) hipLaunchKernelGGL(k_ar_q8<0>, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)s, st, x, residual, out, n, gper, world_size);
    else hipLaunchKernelGGL(k_ar_q8<1>, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)s, st, x, residual, out, n, gper, world_size);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ZL_OK : (int)e;
}
int zl_ar_status(void* state, zl_comm_stream_t s) {
    ZL_CHECK_ARG(state, ZL_EINVAL);
    unsigned err = 0;
    ZL_HIP(hipMemcpyAsync(&err, reinterpret_cast<char*>(state) + offsetof(ArState, err), sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)s));
    ZL_HIP(hipStreamSynchronize((hipStream_t)s));
    return (int)err;
}

}  // extern "C"

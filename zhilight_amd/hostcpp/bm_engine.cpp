// bm_engine.cpp -- bmengine::core::Engine on MI355X (bm_engine.h): one thread, one exchange buffer, one one-shot all-reduce
// state and -- between distinct devices -- one RCCL communicator per tensor-parallel rank, all inside one process
// (3rd/bmengine/bmengine/core/engine.cpp:56-59, 140-157, 307-340 is what it stands in for).  Host code only; the transports
// are libzhilight_amd_comm.so's (include/zhilight_amd_comm.h).
#include "bm_engine.h"

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <set>

#include "bm_c10d.h"
#include "zhilight_amd.h"
#include "zhilight_amd_comm.h"

namespace bmengine {
namespace core {

namespace {

std::string comm_status(int st) {
    if (st >= 1000) return "ncclResult_t " + std::to_string(st - 1000);
    return zl_status_string(st);
}
#define EN_CK(call, what)                                                                                              \
    do {                                                                                                               \
        const int st_ = (call);                                                                                        \
        if (st_ != 0) throw BMEngineException(std::string(what) + ": " + comm_status(st_), __FILE__, __LINE__, __func__); \
    } while (0)

// dtype codes of zhilight_amd_comm.h
int comm_dtype(DataType dt) {
    switch (dt) {
    case DataType::kHalf: return 0;
    case DataType::kBFloat16: return 1;
    case DataType::kFloat: return 2;
    case DataType::kInt32: return 3;
    case DataType::kInt8: return 4;
    default: return -1;
    }
}
bool is_16bit_float(DataType dt) { return dt == DataType::kHalf || dt == DataType::kBFloat16; }
bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
int64_t env_i64(const char* name, int64_t dflt) {
    const char* v = std::getenv(name);
    return v && *v ? std::atoll(v) : dflt;
}

}  // namespace

class EngineImpl {
public:
    struct Rank {
        int device = 0;
        zl_comm_t* comm = nullptr;
        void* ar_buffer = nullptr;
        void* ar_state = nullptr;
        void* staging = nullptr;      // ranks that share a device: scratch of the byte-wise collectives (oneshot_bytes; no allocation per call)
        hipStream_t setup_stream = nullptr;
        hipStream_t secondary_stream = nullptr;   // ranks that share a device: the rank's second stream (bm_hip.h set_shared_device_rank)
        MemoryAllocator allocator;
        std::unique_ptr<TaskThreadPool> thread;
        int share_index = 0;          // how many lower ranks sit on the same device
    };
    std::vector<Rank> ranks;
    bool rccl = false;
    int64_t oneshot_bytes = 8 << 20;
    int world() const { return (int)ranks.size(); }

    // every rank's thread runs fn(rank) at the same time (the communicator handshake needs them concurrent); the first failure
    // is rethrown after all of them finished
    void foreach_rank(const std::function<void(int)>& fn) {
        for (int r = 0; r < world(); ++r) {
            const int dev = ranks[r].device;
            ranks[r].thread->run([fn, r, dev] {
                BM_HIPRT_ASSERT(hipSetDevice(dev));
                fn(r);
            });
        }
        std::exception_ptr first;
        for (int r = 0; r < world(); ++r) {
            try {
                ranks[r].thread->wait();
            } catch (...) {
                if (!first) first = std::current_exception();
            }
        }
        if (first) std::rethrow_exception(first);
    }

    // ---- the collectives of one rank (stream-ordered, no host synchronisation) ------------------------------------------------
    // fp16 / bf16 sums: the one-shot exchange (in pieces when the message outgrows the exchange buffers and there is no RCCL)
    bool oneshot_fits(const void* send, const void* recv, size_t count, DataType dt) const {
        if (!is_16bit_float(dt) || count % 8 || !aligned16(send) || !aligned16(recv)) return false;
        return !rccl || (int64_t)(count * 2) <= oneshot_bytes;
    }
    void all_reduce_sum(int r, const void* send, void* recv, size_t count, DataType dt, hipStream_t st) const {
        const Rank& R = ranks[r];
        if (oneshot_fits(send, recv, count, dt)) {
            const size_t per = (size_t)oneshot_bytes / 2;
            for (size_t off = 0; off < count; off += per) {
                const size_t n = std::min(per, count - off);
                EN_CK(zl_ar_all_reduce(R.ar_state, (const uint16_t*)send + off, nullptr, (uint16_t*)recv + off, (int64_t)n, comm_dtype(dt), st),
                      "one-shot all-reduce");
            }
            return;
        }
        BM_ASSERT(rccl, "all-reduce of this dtype / shape / alignment needs the RCCL communicator, and the ranks of this engine share a device "
                        "(fp16 / bf16 messages of a multiple of 8 elements at 16-byte aligned addresses run on the one-shot exchange)");
        BM_ASSERT(comm_dtype(dt) >= 0, "all-reduce: dtype without an RCCL type");
        EN_CK(zl_comm_all_reduce_sum(R.comm, send, recv, (int64_t)count, comm_dtype(dt), st), "ncclAllReduce");
    }
    // Ranks that share a device have no RCCL: gathers and broadcasts are sums of zero-padded slices on the one-shot exchange (adding
    // zeros is exact; a -0.0 comes back as +0.0).  fp16 / bf16 payloads travel as they are; ANY other payload travels byte by byte as
    // the fp16 value of each byte's int8 reading (-128 .. 127 are exact in fp16, x + 0 is exact, the way back is exact): bit-exact
    // for int32 routing tables and fp32 routing weights (FeedForward::route broadcasts them, feedforward.cpp:472-478).
    void sum_bytes(int r, void* buf, size_t nbytes, hipStream_t st) const {
        // through the rank's persistent staging block, in pieces of oneshot_bytes / 2 payload bytes (one fp16 per byte).  (A
        // stream-ordered allocation per call handed blocks back and forth between the two ranks' streams through the device's
        // shared pool; the second broadcast of a step then came back as zeros on one rank -- gpurun_out/r21_ds.log.)
        const Rank& R = ranks[r];
        BM_ASSERT(R.staging, "byte-wise collective without a staging block");
        const size_t piece = (size_t)oneshot_bytes / 2;
        for (size_t off = 0; off < nbytes; off += piece) {
            const size_t n = std::min(piece, nbytes - off), padded = (n + 7) / 8 * 8;
            if (padded != n) BM_HIPRT_ASSERT(hipMemsetAsync(R.staging, 0, padded * 2, st));
            EN_CK(zl_cast((const char*)buf + off, ZL_T_I8, R.staging, ZL_T_F16, (int64_t)n, st), "bytes -> fp16");
            all_reduce_sum(r, R.staging, R.staging, padded, DataType::kHalf, st);
            EN_CK(zl_cast(R.staging, ZL_T_F16, (char*)buf + off, ZL_T_I8, (int64_t)n, st), "fp16 -> bytes");
        }
    }
    // recv holds this rank's contribution and zeros elsewhere -> the sum over the ranks, in place
    void sum_padded(int r, void* recv, size_t count, DataType dt, hipStream_t st) const {
        if (is_16bit_float(dt) && count % 8 == 0 && aligned16(recv)) all_reduce_sum(r, recv, recv, count, dt, st);
        else sum_bytes(r, recv, count * get_elem_size(dt), st);
    }
    void gather_by_sum(int r, const void* send, void* recv, size_t count, DataType dt, hipStream_t st) const {
        const size_t esz = get_elem_size(dt);
        BM_HIPRT_ASSERT(hipMemsetAsync(recv, 0, count * esz * world(), st));
        BM_HIPRT_ASSERT(hipMemcpyAsync((char*)recv + (size_t)r * count * esz, send, count * esz, hipMemcpyDeviceToDevice, st));
        sum_padded(r, recv, count * world(), dt, st);
    }
};

Engine::Engine(const std::vector<DeviceConfiguration>& dev_cfg, const DistConfiguration& dist_cfg) : pimpl(new EngineImpl) {
    const int world = (int)dev_cfg.size();
    BM_ASSERT(world >= 1 && world <= ZL_AR_MAX_RANKS, "Engine: 1 .. 8 ranks on one node");
    BM_ASSERT(dist_cfg.tp <= 0 || dist_cfg.tp == world, "Engine: tp must equal the number of devices (no pipeline parallelism on this path)");
    BM_ASSERT(dist_cfg.nnodes == 1, "Engine: one node");
    std::set<int> distinct;
    pimpl->ranks.resize(world);
    for (int r = 0; r < world; ++r) {
        pimpl->ranks[r].device = dev_cfg[r].device_id;
        pimpl->ranks[r].thread.reset(new TaskThreadPool(1, r));
        distinct.insert(dev_cfg[r].device_id);
    }
    const bool all_distinct = (int)distinct.size() == world;
    for (int r = 0; r < world; ++r)
        for (int p = 0; p < r; ++p) pimpl->ranks[r].share_index += pimpl->ranks[p].device == pimpl->ranks[r].device;
    if (world > 1 && !all_distinct) {
        // The ranks' SECONDARY streams (what dual_stream_encode asks cudaStreamCreateWithPriority for, block.cpp:221-224): created
        // here, one after the other and before anything else touches the runtime's lowest-priority queue pool -- the runtime deals
        // a pool's hardware queues in creation order, so no two of them share a queue, and none shares one with a rank's main stream
        // (those sit in the higher-priority pools, create_context_rank).  They live as long as the engine; a rank thread's request
        // hands its own out, and "destroying" one is a no-op (zl_shim_stream_destroy).
        int least = 0, greatest = 0;
        BM_HIPRT_ASSERT(hipDeviceGetStreamPriorityRange(&least, &greatest));
        for (int r = 0; r < world; ++r) {
            BM_HIPRT_ASSERT(hipSetDevice(pimpl->ranks[r].device));
            BM_HIPRT_ASSERT(hipStreamCreateWithPriority(&pimpl->ranks[r].secondary_stream, hipStreamNonBlocking, least));
        }
    }
    pimpl->rccl = world > 1 && all_distinct && env_i64("ZL_ENGINE_RCCL", 1) != 0;
    pimpl->oneshot_bytes = std::max<int64_t>(4096, env_i64("ZL_ENGINE_ONESHOT_BYTES", 8 << 20)) / 16 * 16;
    EngineImpl* impl = pimpl.get();
    // phase 1: streams, exchange buffers, peer access
    impl->foreach_rank([impl, world, all_distinct](int r) {
        EngineImpl::Rank& R = impl->ranks[r];
        BM_HIPRT_ASSERT(hipStreamCreateWithFlags(&R.setup_stream, hipStreamNonBlocking));
        if (world == 1) return;
        EN_CK(zl_ar_alloc(zl_ar_buffer_bytes(impl->oneshot_bytes), &R.ar_buffer), "exchange buffer");
        BM_HIPRT_ASSERT(hipMalloc(&R.ar_state, (size_t)zl_ar_state_bytes()));
        // zeroed on the stream zl_ar_init writes the state on (a null-stream memset would be unordered against that non-blocking stream)
        BM_HIPRT_ASSERT(hipMemsetAsync(R.ar_state, 0, (size_t)zl_ar_state_bytes(), R.setup_stream));
        if (!all_distinct) BM_HIPRT_ASSERT(hipMalloc(&R.staging, (size_t)impl->oneshot_bytes));
        BM_HIPRT_ASSERT(hipStreamSynchronize(R.setup_stream));
        if (all_distinct)
            for (int p = 0; p < world; ++p) {
                if (p == r) continue;
                const hipError_t e = hipDeviceEnablePeerAccess(impl->ranks[p].device, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                else BM_HIPRT_ASSERT(e);
            }
    });
    if (world == 1) return;
    // phase 2: the communicator handshake (every rank at once) and the address tables of the one-shot exchange
    char uid[ZL_COMM_UNIQUE_ID_BYTES];
    std::memset(uid, 0, sizeof(uid));
    if (impl->rccl) EN_CK(zl_comm_unique_id(uid), "ncclGetUniqueId");
    std::vector<void*> buffers(world);
    for (int r = 0; r < world; ++r) buffers[r] = impl->ranks[r].ar_buffer;
    impl->foreach_rank([impl, world, &uid, &buffers](int r) {
        EngineImpl::Rank& R = impl->ranks[r];
        if (impl->rccl) EN_CK(zl_comm_create(&R.comm, world, r, uid), "ncclCommInitRank");
        EN_CK(zl_ar_init(R.ar_state, world, r, buffers.data(), impl->oneshot_bytes, R.setup_stream), "one-shot exchange setup");
    });
}

Engine::~Engine() {
    EngineImpl* impl = pimpl.get();
    try {
        impl->foreach_rank([impl](int r) {
            EngineImpl::Rank& R = impl->ranks[r];
            (void)hipDeviceSynchronize();
            if (R.comm) (void)zl_comm_destroy(R.comm);
            if (R.ar_buffer) (void)zl_ar_free(R.ar_buffer);
            if (R.ar_state) (void)hipFree(R.ar_state);
            if (R.staging) (void)hipFree(R.staging);
            if (R.setup_stream) (void)hipStreamDestroy(R.setup_stream);
            if (R.secondary_stream) {
                forget_secondary_stream(R.secondary_stream);
                (void)hipStreamDestroy(R.secondary_stream);
            }
        });
    } catch (...) {
    }
}

int Engine::num_gpus() const { return pimpl->world(); }
int Engine::world_size() const { return pimpl->world(); }
int Engine::local_ranks() const { return pimpl->world(); }
bool Engine::has_rccl() const { return pimpl->rccl; }
MemoryAllocator* Engine::get_allocator(int dev_id) {
    BM_ASSERT(dev_id >= 0 && dev_id < pimpl->world(), "Engine::get_allocator: rank out of range");
    return &pimpl->ranks[dev_id].allocator;
}
GPUInfo Engine::get_gpu_info(int device_idx) const {
    BM_ASSERT(device_idx >= 0 && device_idx < pimpl->world(), "Engine::get_gpu_info: rank out of range");
    GPUInfo info{};
    info.real_device_idx = pimpl->ranks[device_idx].device;
    info.compute_capability = 90;      // (Context::get_compute_capability: what the reference's > 80 gates see)
    int prev = 0;
    BM_HIPRT_ASSERT(hipGetDevice(&prev));
    BM_HIPRT_ASSERT(hipSetDevice(info.real_device_idx));
    size_t free_b = 0, total_b = 0;
    BM_HIPRT_ASSERT(hipMemGetInfo(&free_b, &total_b));
    BM_HIPRT_ASSERT(hipSetDevice(prev));
    info.total_memory = total_b;
    info.free_memory = free_b;
    info.alloc_memory = total_b - free_b;
    return info;
}
void Engine::print_memory_summary() {
    for (int r = 0; r < pimpl->world(); ++r) {
        const GPUInfo g = get_gpu_info(r);
        std::cerr << "rank " << r << " device " << g.real_device_idx << ": " << (g.alloc_memory >> 20) << " MB in use of " << (g.total_memory >> 20) << " MB\n";
    }
}
void Engine::device_foreach(std::function<void(int)> fn) { pimpl->foreach_rank(fn); }
void Engine::run(int rank, std::function<void()> fn) {
    BM_ASSERT(rank >= 0 && rank < pimpl->world(), "Engine::run: rank out of range");
    const int dev = pimpl->ranks[rank].device;
    pimpl->ranks[rank].thread->runSync([fn, dev] {
        BM_HIPRT_ASSERT(hipSetDevice(dev));
        fn();
    });
}
int Engine::exchange_errors(int rank) const {
    BM_ASSERT(rank >= 0 && rank < pimpl->world(), "Engine::exchange_errors: rank out of range");
    const EngineImpl::Rank& R = pimpl->ranks[rank];
    return R.ar_state ? zl_ar_status(R.ar_state, R.setup_stream) : 0;
}

Context Engine::create_context() const { return create_context_rank(0); }
Context Engine::create_context(const std::vector<int>& devices) const {
    BM_ASSERT(devices.size() == 1, "Engine::create_context: one device per context (tensor parallelism = one context per rank)");
    return create_context_rank(devices[0]);
}

Context Engine::create_context_rank(int rank) const {
    EngineImpl* impl = pimpl.get();
    const int world = impl->world();
    BM_ASSERT(rank >= 0 && rank < world, "Engine::create_context_rank: rank out of range");
    Context ctx(impl->ranks[rank].device, rank, world);
    if (world == 1) return ctx;
    bool shared_device = false;
    for (int p = 0; p < world; ++p) shared_device |= p != rank && impl->ranks[p].device == impl->ranks[rank].device;
    if (shared_device) {
        // Ranks that share a device exchange through kernels that WAIT for each other inside a launch: their streams must sit on
        // different hardware queues, or one rank's waiting kernel blocks the peer's kernel queued behind it until the bounded wait
        // expires (measured: whole steps timing out alternately on either rank whenever the runtime's round-robin put both streams on
        // one of its 4 queues, gpurun_out/r16_diag_tp.log).  The runtime keeps one queue pool per stream PRIORITY, so ranks of one
        // device get streams of different priorities -- as many ranks per device as there are priority levels.
        int least = 0, greatest = 0;
        BM_HIPRT_ASSERT(hipDeviceGetStreamPriorityRange(&least, &greatest));
        // (the lowest level is kept for the ranks' SECONDARY streams: dual_stream_encode's reduce stream, bm_hip.h set_shared_device_rank)
        const int levels = least - greatest;
        set_shared_device_rank(impl->ranks[rank].share_index, impl->ranks[rank].secondary_stream);
        BM_ASSERT(impl->ranks[rank].share_index < levels,
                  "Engine: more ranks share one device than the runtime has stream priority levels (use one process per rank there)");
        hipStream_t s = nullptr;
        BM_HIPRT_ASSERT(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest + impl->ranks[rank].share_index));
        ctx.set_current_stream(std::make_shared<Stream_>(s, [](hipStream_t p) { (void)hipStreamDestroy(p); }));
    }
    const int r = rank;
    c10d::Collectives c;
    c.comm_count = world;
    c.user_rank = rank;
    c.all_reduce = [impl, r](const Tensor& send, Tensor& recv, ncclRedOp_t op, hipStream_t st) {
        BM_ASSERT(op == ncclSum, "NCCLAllReduce: the path sums (ModelContext::reduce_sum2, model_context.cpp:236)");
        BM_ASSERT_EQ(send.nbytes(), recv.nbytes(), "NCCLAllReduce: send / recv sizes differ");
        impl->all_reduce_sum(r, send.data(), recv.data(), send.numel(), send.dtype(), st);
    };
    c.all_gather = [impl, r, world](const Tensor& send, Tensor& recv, hipStream_t st) {
        BM_ASSERT_EQ(send.nbytes() * (size_t)world, recv.nbytes(), "NCCLAllGather: recv is world x send");
        if (impl->rccl) {
            BM_ASSERT(comm_dtype(send.dtype()) >= 0, "NCCLAllGather: dtype without an RCCL type");
            EN_CK(zl_comm_all_gather(impl->ranks[r].comm, send.data(), recv.data(), (int64_t)send.numel(), comm_dtype(send.dtype()), st), "ncclAllGather");
        } else {
            impl->gather_by_sum(r, send.data(), recv.data(), send.numel(), send.dtype(), st);
        }
    };
    c.broadcast = [impl, r](const Tensor& send, Tensor& recv, int root, hipStream_t st) {
        BM_ASSERT(root >= 0 && root < impl->world(), "NCCLBroadcast: root out of range");
        if (r == root) {
            BM_ASSERT_EQ(send.nbytes(), recv.nbytes(), "NCCLBroadcast: send / recv sizes differ");
            if (send.data() != recv.data()) BM_HIPRT_ASSERT(hipMemcpyAsync(recv.data(), send.data(), send.nbytes(), hipMemcpyDeviceToDevice, st));
        }
        if (impl->rccl) {
            BM_ASSERT(comm_dtype(recv.dtype()) >= 0, "NCCLBroadcast: dtype without an RCCL type");
            EN_CK(zl_comm_broadcast(impl->ranks[r].comm, recv.data(), (int64_t)recv.numel(), comm_dtype(recv.dtype()), root, st), "ncclBroadcast");
        } else {
            if (r != root) BM_HIPRT_ASSERT(hipMemsetAsync(recv.data(), 0, recv.nbytes(), st));
            impl->sum_padded(r, recv.data(), recv.numel(), recv.dtype(), st);
        }
    };
    c.reduce_scatter = [impl, r, world](const Tensor& send, Tensor& recv, ncclRedOp_t op, hipStream_t st) {
        BM_ASSERT(op == ncclSum, "NCCLReduceScatter: sum only");
        BM_ASSERT_EQ(recv.nbytes() * (size_t)world, send.nbytes(), "NCCLReduceScatter: send is world x recv");
        if (impl->rccl) {
            BM_ASSERT(comm_dtype(send.dtype()) >= 0, "NCCLReduceScatter: dtype without an RCCL type");
            EN_CK(zl_comm_reduce_scatter_sum(impl->ranks[r].comm, send.data(), recv.data(), (int64_t)recv.numel(), comm_dtype(send.dtype()), st),
                  "ncclReduceScatter");
        } else {
            // ranks sharing a device: the whole sum in the staging block, then this rank's slice
            BM_ASSERT((int64_t)send.nbytes() <= impl->oneshot_bytes && impl->ranks[r].staging, "reduce-scatter between ranks that share a device: within the staging block");
            void* tmp = impl->ranks[r].staging;
            impl->all_reduce_sum(r, send.data(), tmp, send.numel(), send.dtype(), st);
            BM_HIPRT_ASSERT(hipMemcpyAsync(recv.data(), (char*)tmp + (size_t)r * recv.nbytes(), recv.nbytes(), hipMemcpyDeviceToDevice, st));
        }
    };
    if (impl->rccl) {
        c.send = [impl, r](const Tensor& buf, int peer, hipStream_t st) {
            EN_CK(zl_comm_send(impl->ranks[r].comm, buf.data(), (int64_t)buf.numel(), comm_dtype(buf.dtype()), peer, st), "ncclSend");
        };
        c.recv = [impl, r](Tensor& buf, int peer, hipStream_t st) {
            EN_CK(zl_comm_recv(impl->ranks[r].comm, buf.data(), (int64_t)buf.numel(), comm_dtype(buf.dtype()), peer, st), "ncclRecv");
        };
        // (group calls carry no context and reach every rank's entry, bm_c10d.cpp: ncclGroupStart / End are per calling thread)
        c.group_start = [] { EN_CK(zl_comm_group_start(), "ncclGroupStart"); };
        c.group_end = [] { EN_CK(zl_comm_group_end(), "ncclGroupEnd"); };
    }
    c10d::set_collectives(ctx, c);
    // Context::reduce_sum of a plain (non-model) context: in place
    ctx.set_reduce_hook([impl, r](Tensor& data, hipStream_t st) { impl->all_reduce_sum(r, data.data(), data.data(), data.numel(), data.dtype(), st); });
    return ctx;
}

}  // namespace core
}  // namespace bmengine
